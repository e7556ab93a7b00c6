set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in NA NP; do
  MARL_ENV_LIB=$PWD/build_ab/lib$v.so timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/fprof_$v -o r -- python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/fprof_$v.log 2>&1
done
