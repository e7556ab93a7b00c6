#!/bin/bash
# One launch sequence over batch sizes (SWEEP_ENVS), per library build (SWEEP_LIBS: paths, "-" = the built library) and with
# the solves' issue priority allowed and forbidden: ms per step.
for E in ${SWEEP_ENVS:-4096 6144 7168 8192 9216 10240}; do for lib in ${SWEEP_LIBS:--}; do for f in "" "--no-solve-priority"; do
  [ "$lib" = "-" ] && unset MARL_ENV_LIB || export MARL_ENV_LIB=$lib
  echo -n "E=$E lib=$lib $f: "
  timeout -k 10 120 python bench.py --config 3 --steps 2000 --warmup 200 --no-cpu-baseline --shards 1 --solver direct --envs-per-gpu $E $f 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms/step  %.3e env-steps/s' % (d['ms_per_step'], d['value']))" || exit 1
done; done; done
