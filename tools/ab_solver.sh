for f in "--solver lanczos" "--solver direct"; do
  for a in "--steps 20 --warmup 5" "--steps 2000" "--steps 400 --shards 1 --envs-per-gpu 16384" "--steps 400 --shards 1 --envs-per-gpu 4096" "--steps 400 --shards 1 --envs-per-gpu 12288" "--steps 400 --shards 1 --envs-per-gpu 65536"; do
    timeout -k 10 120 python bench.py --no-cpu-baseline $a $f > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; exit 1; }
    python - "$a $f" <<'PY'
import json, sys
d = json.load(open('gpurun_out/ab.json'))
print(f"{sys.argv[1]:60s} value {d['value']:.4g}  ms/step {d['ms_per_step']:.4f}  kernel_ms {d['roofline']['kernel_ms']:.4f} streams {d.get('streams')}")
PY
  done
done
