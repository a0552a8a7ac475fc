"""BASELINE configs[2] against the glibc-libm oracle tiles: the measured figures behind the bars of tests/test_golden.py."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pbrt_v3_spectral_amd as pt
z = np.load(os.path.join(ROOT, "tests", "golden", "cornell_4096spp_tiles.npz"))
s = pt.Scene(os.path.join(ROOT, "scenes", "cornell-glass.pbrt"), spp=4096)
integ = pt.CreatePathIntegrator(s)
f0, w0 = integ.Render(shard_index=0, shard_count=int(z["shard_count"]))
ys, xs = z["ys"].astype(int), z["xs"].astype(int)
film = f0[ys, xs]; spp = 4096
gx = z["film_exact"]; mean = gx.mean() / spp
gold = z["film"]
pg = np.sqrt(((film.astype(np.float64) - gold) ** 2).mean(axis=-1)) / spp
rel = np.sqrt(((film.astype(np.float64) - gold) ** 2).sum() / (gold.astype(np.float64) ** 2).sum())
print("cornell vs glibc oracle: rel L2 %.3e, frac over 1e-3 mean %.4f, max over mean %.3f" % (rel, (pg > 1e-3 * mean).mean(), pg.max() / mean))
