set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline"
timeout -k 10 400 python3 bench.py > gpurun_out/bench4.json 2> gpurun_out/bench4.err
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof4 -o r -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/prof4.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_WAIT_INST_ANY -d gpurun_out/pmc4_sq -- $B > gpurun_out/pmc4_sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc4_fetch -- $B > gpurun_out/pmc4_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc4_write -- $B > gpurun_out/pmc4_write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc GRBM_COUNT GRBM_GUI_ACTIVE -d gpurun_out/pmc4_grbm -- $B > gpurun_out/pmc4_grbm.log 2>&1
cat gpurun_out/bench4.json
