#!/usr/bin/env python3
"""Per-SIMD view of gpurun_out/solve_timeline.npy (written by tools/solve_timeline.py): span of every SIMD's
share of the solve launch, and the blocks (item, n', start, end) of the slowest, a median and the fastest SIMDs."""
import numpy as np
d = np.load('gpurun_out/solve_timeline.npy')
blk, xcc, hw, npv, t0, t1 = d
simd = (xcc.astype(np.int64) << 16) | (hw & 0xFF30)
rows = []
for k in np.unique(simd):
    m = simd == k
    b = t0[m].min()
    s = (t0[m] - b) & 0xFFFFFFFF
    e = (t1[m] - b) & 0xFFFFFFFF
    o = np.argsort(s)
    rows.append((int(e.max()), [(int(a), int(n), int(c), int(dd)) for a, n, c, dd in zip(blk[m][o], npv[m][o], s[o], e[o])]))
spans = np.array([r[0] for r in rows])
print("per-SIMD span: min %d  p10 %d median %d p90 %d max %d" % (spans.min(), np.percentile(spans, 10), np.median(spans),
                                                               np.percentile(spans, 90), spans.max()))
rows.sort(key=lambda r: -r[0])
for r in rows[:3] + rows[len(rows) // 2:len(rows) // 2 + 3] + rows[-3:]:
    print(r[0], r[1])
