#!/usr/bin/env python3
"""What a plain streaming read / copy reaches on this GPU (calibration for the row-matrix passes of the PPO step)."""
import torch
n = 1 << 28                                    # 1 GiB of float32
x = torch.randn(n, device="cuda")
y = torch.empty_like(x)


def timed(f, reps=20):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


t = timed(lambda: x.sum())
print("sum   of 1 GiB: %.3f ms  %.2f TB/s read" % (t, n * 4 / t / 1e9))
t = timed(lambda: y.copy_(x))
print("copy  of 1 GiB: %.3f ms  %.2f TB/s read+write" % (t, 2 * n * 4 / t / 1e9))
t = timed(lambda: y.fill_(1.0))
print("fill  of 1 GiB: %.3f ms  %.2f TB/s write" % (t, n * 4 / t / 1e9))
