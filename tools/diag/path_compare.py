"""Diagnostic (GPU box): vertex-by-vertex comparison of device and oracle for the samples that differ.
Usage: path_compare.py <scene name of sample_divergence.mk> [spp] [max samples shown] [all]
"all": walk every camera sample of the frame instead of the ones whose one-sample film differs (needed when the filter is
not the box filter or the sampler is not Halton: then a sample's film is not one pixel) and print only paths that differ."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import pbrt_v3_spectral_amd as pt, oracle_binding as ob
from sample_divergence import mk, per_sample

name = sys.argv[1]
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nshow = int(sys.argv[3]) if len(sys.argv) > 3 else 10
s = mk(name, -1, spp)
walk_all = len(sys.argv) > 4 and sys.argv[4] == "all"
if walk_all:
    w, h = s.film_size
    bad = [(x, y, k) for y in range(h) for x in range(w) for k in range(spp)]
else:
    bad = per_sample(s, list(range(1, spp)))
    print(name, "differing samples:", len(bad))
integ = pt.CreatePathIntegrator(s)
FIELDS = [("bounces", 0, 1), ("prim", 1, 2), ("dim_before", 2, 3), ("ended", 3, 4), ("ray_o", 4, 7), ("t_hit", 7, 8), ("ray_d", 8, 11),
          ("etaScale_in", 11, 12), ("next_o", 12, 15), ("dim_after", 15, 16), ("next_d", 16, 19), ("etaScale_out", 19, 20),
          ("beta", 20, 51), ("L", 51, 82)]
shown = 0
for (x, y, k) in list(bad):
    if shown >= nshow: break
    d = integ.debug_path(x, y, k)
    o = ob.path_log(s, x, y, k)
    dg, og = d[:, :20].copy(), o[:, :20].copy()
    if len(d) == len(o):   # a path's last vertex spawns no ray that anything reads: what the two sides leave there differs
        dg[-1, 12:15] = og[-1, 12:15] = 0; dg[-1, 16:20] = og[-1, 16:20] = 0
    if walk_all and len(d) == len(o) and np.array_equal(dg.view(np.uint32), og.view(np.uint32)) \
            and np.allclose(d[:, 51:82], o[:, 51:82], rtol=1e-4, atol=1e-7, equal_nan=True) and not (np.isinf(d[:, 51:82]).any() or np.isnan(d[:, 51:82]).any() or (d[:, 51:82] < 0).any()): continue
    shown += 1
    print("pixel (%d,%d) k=%d: device %d vertices, oracle %d" % (x, y, k, len(d), len(o)))
    for v in range(min(len(d), len(o))):
        diffs = []
        for nm, a, b in FIELDS:
            dv, ov = d[v, a:b], o[v, a:b]
            if not np.array_equal(dv.view(np.uint32), ov.view(np.uint32)) and not (np.isnan(dv).all() and np.isnan(ov).all()):
                if b - a > 3:
                    i = int(np.argmax(np.abs(dv - ov)))
                    diffs.append("%s[%d] %.9g vs %.9g" % (nm, i, dv[i], ov[i]))
                else:
                    diffs.append("%s %s vs %s" % (nm, np.array2string(dv, precision=9), np.array2string(ov, precision=9)))
        print("   vertex %d (bounces %d, prim %d, dims %d->%d): %s" % (v, d[v, 0], d[v, 1], d[v, 2], d[v, 15], "; ".join(diffs) if diffs else "identical"))
        if diffs and any(not t.startswith(("L[", "beta[")) for t in diffs):
            break
