#!/bin/bash
# kernel trace of tools/ppo_rows_bench.py: average duration of the PPO kernels
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ptr
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ptr -o r -- python3 tools/ppo_rows_bench.py > gpurun_out/ptr.log 2>&1 || { tail -5 gpurun_out/ptr.log; exit 1; }
f=$(find gpurun_out/ptr -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/ptr_kernel_stats.csv
grep -E "ppo_|Name" "$f" | cut -c1-200
rm -rf gpurun_out/ptr
