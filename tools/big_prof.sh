set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/bigstep.py <<'PY'
import sys, torch
sys.path.insert(0, ".")
from marl_amd import BatchedJammerEnv, EnvConfig
E, N, M = 8192, 200, 16
env = BatchedJammerEnv(E, EnvConfig.fast_preset(N=N, M=M), seed=1, auto_reset=True)
env.reset()
pool = torch.empty(8, E, M, 3, device="cuda").uniform_(-5, 5)
pool[..., 2] = torch.randint(0, 4, (8, E, M), device="cuda").float()
for i in range(12):
    env.step(pool[i % 8])
torch.cuda.synchronize()
print("np quartiles", torch.quantile(env.views["workspace_pending"].float(), torch.tensor([0.0, .25, .5, .75, 1.0], device="cuda")).tolist() if "workspace_pending" in env.views else "")
PY
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/bigprof -o r -- python3 /tmp/bigstep.py > gpurun_out/bigprof.log 2>&1
tail -2 gpurun_out/bigprof.log
