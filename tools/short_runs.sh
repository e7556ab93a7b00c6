#!/bin/bash
# Spread of short bench runs (the timed region holds an auto-reset): env-steps/s, ms per step, kernel ms per step
for a in "--shards 3" "--shards 3" "--shards 3" "--shards 3" "--shards 1" "--shards 1" "--shards 1" "--shards 1"; do
  echo "== --steps 20 --warmup 5 $a"; timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 $a --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])" || exit 1
done
