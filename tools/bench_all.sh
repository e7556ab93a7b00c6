#!/bin/bash
# bench.py for every BASELINE config on one GPU; one JSON line each under gpurun_out/.
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
tag=${1:-r02}
for c in 3 2 4 5; do
  timeout -k 10 420 python3 bench.py --config $c ${BENCH_ARGS:-} > gpurun_out/${tag}_bench_c$c.json 2> gpurun_out/${tag}_bench_c$c.err || { echo "config $c failed"; tail -5 gpurun_out/${tag}_bench_c$c.err; exit 1; }
  python3 - <<PY
import json
d = json.load(open("gpurun_out/${tag}_bench_c$c.json"))
print("config $c: %.4g env-steps/s  %.4f ms/step  kernel %.4f ms  frac %.3f  l2 stage %.4f us/env  cpu %.4g" % (
    d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["lambda2_stage_op_us_per_env"],
    d.get("cpu_baseline", {}).get("value", 0)), {k: d[k] for k in ("collection_env_steps_per_s", "update_s_per_rollout") if k in d})
PY
done
