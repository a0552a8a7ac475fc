"""Quick GPU-vs-oracle check used during development (not a test): renders a scene
with the HIP path and with the CPU oracle and prints counters + image differences."""
import argparse, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pbrt_v3_spectral_amd as pt
import oracle_binding as ob

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default=os.path.join(ROOT, "scenes/killeroo-simple.pbrt"))
ap.add_argument("--spp", type=int, default=4)
ap.add_argument("--res", type=int, default=-1)
ap.add_argument("--pool", type=int, default=0)
ap.add_argument("--no-oracle", action="store_true")
a = ap.parse_args()
s = pt.Scene(a.scene, spp=a.spp, xres=a.res, yres=a.res)
print("scene", s.stats, "film", s.film_size, "spp", s.spp, "errors", s.errors)
t = time.time(); integ = pt.CreatePathIntegrator(s); print("create %.2fs" % (time.time() - t))
t = time.time(); film, weight = integ.Render(path_pool=a.pool); dt = time.time() - t
c = integ.counters.as_dict()
rays = c["regular_rays"] + c["shadow_rays"]
print("GPU render %.3fs (kernel %.3fs) %s" % (dt, integ.timings()[0], c))
print("timings [total,gen,extend,shade,shadow,mis]", ["%.4f" % t for t in integ.timings()[:6]])
print("GPU Mray/s %.1f Msamples/s %.2f iterations %d" % (rays / integ.timings()[0] / 1e6, c["camera_rays"] / integ.timings()[0] / 1e6, integ.counters.iterations))
print("GPU film mean/sample %.6f weight mean %.6f" % (film.mean() / s.spp, weight.mean()))
if not a.no_oracle:
    ofilm, oweight, oc, secs = ob.render(s)
    print("oracle %.2fs %s" % (secs, oc.as_dict()))
    d = film.astype(np.float64) - ofilm
    l2 = np.sqrt((d ** 2).mean(axis=2)) / s.spp
    rel = np.sqrt((d ** 2).sum() / (ofilm.astype(np.float64) ** 2).sum())
    print("per-pixel L2 (per sample units): max %.3e mean %.3e ; image rel L2 %.3e ; weight maxdiff %.3e" % (l2.max(), l2.mean(), rel, np.abs(weight - oweight).max()))
    nbad = (l2 > 1e-3 * max(1e-9, ofilm.mean() / s.spp)).sum()
    print("pixels with L2 > 1e-3*mean:", int(nbad), "of", l2.size)
    if os.environ.get("DUMP"):
        idx = np.argsort(-l2.ravel())[:12]
        for i in idx:
            y, x = divmod(int(i), l2.shape[1])
            print("px", x, y, "l2", l2[y, x], "gpu", film[y, x, :3], "orc", ofilm[y, x, :3], "w", weight[y, x], oweight[y, x])
        rel_px = np.abs(d).max(axis=2) / np.maximum(np.abs(ofilm).max(axis=2), 1e-6)
        for thr in (1e-6, 1e-5, 1e-4, 1e-3, 1e-2):
            print("pixels with max-bin rel diff >", thr, int((rel_px > thr).sum()))
