set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/qprof -o r -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/qprof.log 2>&1
tail -1 gpurun_out/qprof.log | cut -c1-200
