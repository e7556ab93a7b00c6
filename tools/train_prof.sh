set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/trainprof
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/trainprof -o r -- python3 tools/train_bench.py --envs 2048 --rollouts 1 > gpurun_out/trainprof.log 2>&1
python3 - > gpurun_out/trainprof_top.txt <<'PY'
import sqlite3, collections, glob
db = sqlite3.connect(glob.glob('/tmp/trainprof/**/*_results.db', recursive=True)[0])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
namec = "name" if "name" in cols else "kernel_name"
per = collections.defaultdict(lambda: [0, 0])
for n, d in db.execute(f"select {namec}, duration from kernels"):
    per[n][0] += 1; per[n][1] += d
tot = sum(v[1] for v in per.values())
print("total ms", tot / 1e6)
for n, v in sorted(per.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{v[1]/1e6:9.2f} ms {v[0]:7d} calls {v[1]/v[0]/1e3:8.1f} us  {n[:150]}")
PY
grep "envs=" gpurun_out/trainprof.log
