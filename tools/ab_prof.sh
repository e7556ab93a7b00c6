set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  MARL_ENV_LIB=$PWD/build_ab/lib$v.so timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/ab_$v -o r -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/ab_$v.log 2>&1
done
