import torch
for n in (4096, 65536, 262144, 2097152):
    x = torch.zeros(n, device="cuda")
    y = torch.zeros(4, dtype=torch.float64, device="cuda")
    x.normal_()
    def f():
        m = x.mean(); e = (x * x).mean(); c = (x.abs() > 0.2).float().mean(); k = ((x - 1) - x).mean()
        y.copy_(torch.stack([m, e, c, k]).double())
    f(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        f()
    bad = 0
    for r in range(5):
        x.normal_(mean=float(r))
        g.replay(); torch.cuda.synchronize()
        got = y.clone()
        f(); torch.cuda.synchronize()
        d = float((got - y).abs().max())
        bad = max(bad, d)
    def f2():
        v = lambda t: t.view(-1, 1024).sum(1).sum() / t.numel()
        y.copy_(torch.stack([v(x), v(x * x), v((x.abs() > 0.2).float()), v((x - 1) - x)]).double())
    g2 = torch.cuda.CUDAGraph()
    f2(); torch.cuda.synchronize()
    with torch.cuda.graph(g2):
        f2()
    bad2 = 0
    for r in range(5):
        x.normal_(mean=float(r))
        g2.replay(); torch.cuda.synchronize()
        got = y.clone()
        f(); torch.cuda.synchronize()
        bad2 = max(bad2, float((got - y).abs().max()))
    print(f"n={n}: replay-vs-eager worst diff: plain mean {bad:.3e}, two-stage {bad2:.3e}", flush=True)
