#!/bin/bash
# one counter pass over tools/ppo_rows_bench.py: per-kernel sums for the two PPO kernels
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ppmc
timeout -s KILL 120 rocprofv3 --pmc ${PPO_PMC:-SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS} --output-format csv -d gpurun_out/ppmc -o r -- python3 tools/ppo_rows_bench.py > gpurun_out/ppmc.log 2>&1 || { tail -5 gpurun_out/ppmc.log; exit 1; }
f=$(find gpurun_out/ppmc -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    if "ppo_" not in name: continue
    k = name[name.index("ppo_"):].split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
for k in acc:
    print(k, "launches", n[k])
    for c, v in sorted(acc[k].items()):
        print("   %-28s %14.0f per launch" % (c, v / max(n[k], 1)))
PY
rm -rf gpurun_out/ppmc
