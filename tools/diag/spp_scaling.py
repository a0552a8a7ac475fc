#!/usr/bin/env python3
"""Where a film's distance to the oracle comes from as the sample count grows (procedural 10M-triangle scene, shard 0 of 64):
one 2048-spp pass against eight accumulated 256-spp passes of the same sample numbers, both against the fixture, per pixel."""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import pbrt_v3_spectral_amd as pt
import make_golden as mg

z = np.load(os.path.join(ROOT, "tests", "golden", "procedural_10M_2048spp_tiles.npz"))
s = pt.Scene(mg.procedural_scene(tempfile.mkdtemp()), spp=2048)
integ = pt.CreatePathIntegrator(s)
ys, xs = z["ys"].astype(int), z["xs"].astype(int)
gx = z["film_exact"].astype(np.float64)

def stats(name, f):
    d = f[ys, xs].astype(np.float64) - gx
    rel = np.sqrt((d ** 2).sum() / (gx ** 2).sum())
    pp = np.sqrt((d ** 2).sum(axis=1)) / np.maximum(np.sqrt((gx ** 2).sum(axis=1)), 1e-30)
    order = np.argsort(-pp)
    print("%-22s rel L2 %.3e; per-pixel rel: median %.2e, 99%% %.2e, max %.2e; pixels > 1e-5: %d of %d; the worst 20 carry %.0f%% of the squared error"
          % (name, rel, np.median(pp), np.quantile(pp, .99), pp.max(), (pp > 1e-5).sum(), len(pp),
             100 * (d[order[:20]] ** 2).sum() / (d ** 2).sum()))
    return d

f1, w1 = integ.Render(shard_index=0, shard_count=64)
print("counters", integ.counters.as_dict()["regular_rays"], integ.counters.as_dict()["shadow_rays"])
d1 = stats("one 2048-spp pass", f1)
acc = None
for k in range(8):
    f, w = integ.Render(shard_index=0, shard_count=64, spp=256, sample_begin=256 * k, accumulate=(k > 0))
stats("8 x 256 accumulated", f)
da = f1.astype(np.float64) - f
print("one pass vs 8 passes: rel L2 %.3e" % np.sqrt((da[ys, xs] ** 2).sum() / (gx ** 2).sum()))
for pool in (1 << 22, 1 << 24):
    f2, _ = integ.Render(shard_index=0, shard_count=64, path_pool=pool)
    stats("pool %d" % pool, f2)
os.environ["MIPT_WORK_RUN"] = "1"
f3, _ = integ.Render(shard_index=0, shard_count=64)
stats("runs of 1 sample", f3)
