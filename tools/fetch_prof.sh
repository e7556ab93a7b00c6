set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline"
timeout -k 10 200 python3 bench.py --no-cpu-baseline | cut -c1-200
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc5_fetch -- $B > gpurun_out/pmc5_fetch.log 2>&1
echo done
