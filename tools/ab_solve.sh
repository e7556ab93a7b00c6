#!/bin/bash
# A/B of library builds / step schedules on config 3: env-steps/s, ms per step (and the one-stream figure)
# usage: tools/ab_solve.sh "<lib>:<schedule>:<shards>[:<envs>]" ...
for spec in "$@"; do
  IFS=: read lib sched sh envs <<< "$spec"
  echo "== $spec"
  MARL_ENV_LIB=$PWD/$lib timeout -k 10 120 python bench.py --config 3 --steps 3000 --warmup 200 --no-cpu-baseline --schedule $sched --shards $sh ${envs:+--envs-per-gpu $envs} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('one_stream_ms_per_step'))" || exit 1
done
