"""Diagnostic (GPU box): for the samples that differ, the contribution each path depth adds on the device and in the
oracle: L(maxdepth = d) - L(maxdepth = d - 1) per sample. One differing term with identical later terms = a direct-lighting
value computed differently at that vertex; all later terms different = the path took another direction there."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pbrt_v3_spectral_amd as pt, oracle_binding as ob, scenes_text as st
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from sample_divergence import mk

name = sys.argv[1]
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
full = mk(name, -1, spp)
D = int(full.desc.integrator.max_depth)
w, h = full.film_size
ys, xs = np.mgrid[0:h, 0:w]
dev, orc = {}, {}
for depth in range(0, D + 1):
    s = mk(name, depth, spp)
    integ = pt.CreatePathIntegrator(s)
    for k in range(1, spp):
        f1, _ = integ.Render(spp=1, sample_begin=k)
        samples = np.stack([xs.ravel(), ys.ravel(), np.full(xs.size, k)], axis=1)
        li, _ = ob.li(s, samples)
        dev[(depth, k)] = f1.astype(np.float64).sum(axis=2)
        orc[(depth, k)] = li.reshape(h, w, -1).astype(np.float64).sum(axis=2)
shown = 0
for k in range(1, spp):
    d = np.abs(dev[(D, k)] - orc[(D, k)])
    scale = np.maximum(np.abs(orc[(D, k)]), np.abs(dev[(D, k)]))
    for y, x in np.argwhere(d > 1e-4 * np.maximum(scale, 1e-6)):
        dt = [dev[(dd, k)][y, x] - (dev[(dd - 1, k)][y, x] if dd else 0) for dd in range(D + 1)]
        ot = [orc[(dd, k)][y, x] - (orc[(dd - 1, k)][y, x] if dd else 0) for dd in range(D + 1)]
        print("pixel (%d,%d) k=%d" % (x, y, k))
        print("   device terms:", " ".join("%.6g" % v for v in dt))
        print("   oracle terms:", " ".join("%.6g" % v for v in ot))
        shown += 1
        if shown >= 40:
            sys.exit(0)
