"""Fuzz the HIP path against the oracle on seeded random scenes (tests/scenes_text.py: random_scene), in the exact mode:
the oracle with correctly rounded libm calls, i.e. the device's arithmetic -- every counter must be
equal (2 counts of slack) and the film within float accumulation order. Usage: python tools/fuzz_parity.py <first seed> <count> [res] [spp]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import scenes_text as st
import pbrt_v3_spectral_amd as pt
import oracle_binding as ob

first, count = int(sys.argv[1]), int(sys.argv[2])
res = int(sys.argv[3]) if len(sys.argv) > 3 else 32
spp = int(sys.argv[4]) if len(sys.argv) > 4 else 8
d = tempfile.mkdtemp()
st.write_texture_files(d); st.write_alpha_png(d)
bad = []
for seed in range(first, first + count):
    s = pt.Scene(text=st.random_scene(seed, res=res, spp=spp), base_dir=d)
    if s.errors:
        print(seed, "front-end errors", s.errors[:2]); bad.append(seed); continue
    integ = pt.CreatePathIntegrator(s)
    film, weight = integ.Render()
    with ob.exact_libm():
        ofilm, oweight, oc, _ = ob.render(s)
    c, o = integ.counters.as_dict(), oc.as_dict()
    rel = float(np.sqrt(((film.astype(np.float64) - ofilm) ** 2).sum() / max((ofilm.astype(np.float64) ** 2).sum(), 1e-30)))
    dc = max(abs(c[k] - o[k]) for k in ("camera_rays", "regular_rays", "shadow_rays", "total_paths", "zero_radiance_paths", "path_length_sum"))
    flag = "" if (rel < 1e-6 and dc <= 2 and not np.isnan(film).any() and np.allclose(weight, oweight, rtol=1e-5, atol=1e-6)) else "   <-- CHECK"
    if flag or seed % 20 == 0:
        print("seed %4d rel %.2e max counter difference %d bad %d/%d%s" % (seed, rel, dc, c["bad_samples"], o["bad_samples"], flag), flush=True)
    if flag: bad.append(seed)
    del integ
print("checked %d scenes, %d flagged: %s" % (count, len(bad), bad))
