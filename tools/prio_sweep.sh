#!/bin/bash
# One launch sequence with and without the solves' issue priority, over batch sizes (SWEEP_ENVS)
for E in ${SWEEP_ENVS:-7168 8192 9216 10240}; do for f in "" "--no-solve-priority"; do
  echo "== E=$E $f"; timeout -k 10 120 python bench.py --config 3 --steps 3000 --warmup 200 --no-cpu-baseline --shards 1 --envs-per-gpu $E $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" || exit 1
done; done
