#!/bin/bash
# kernel trace of bench.py --config 5: the 40 kernels with the largest total time (torch's included)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tk
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tk -o r -- python3 bench.py --config 5 --steps 64 --warmup 64 --no-cpu-baseline > gpurun_out/tk_bench.json 2> gpurun_out/tk.err || { tail -5 gpurun_out/tk.err; exit 1; }
f=$(find gpurun_out/tk -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total ms", tot / 1e6, "kernels", sum(int(r["Calls"]) for r in rows))
for r in rows[:40]:
    print("%8.2f ms %7d calls %9.1f us  %s" % (float(r["TotalDurationNs"]) / 1e6, int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:110]))
PY
cp "$f" gpurun_out/tk_kernel_stats.csv
rm -rf gpurun_out/tk
