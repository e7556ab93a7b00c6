#!/bin/bash
# kernel trace of bench.py --config $1 (default 3): per-kernel launch statistics to stdout
set -o pipefail
c=${1:-3}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/qt
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/qt -o r -- python3 bench.py --config $c --no-cpu-baseline "$@" > gpurun_out/qt_bench.json 2> gpurun_out/qt.err || { tail -5 gpurun_out/qt.err; exit 1; }
python3 tools/db_summary.py stats gpurun_out/qt_stats.csv $(ls -t gpurun_out/qt/*/*.db gpurun_out/qt/*.db 2>/dev/null | head -1) | cut -c1-140
rm -rf gpurun_out/qt
