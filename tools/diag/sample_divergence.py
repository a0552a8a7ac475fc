"""Diagnostic (GPU box): which samples differ between device and oracle, and from which path depth on.
Device: a pass with spp = 1, sample_begin = k gives the film of sample k alone (box filter: one pixel per sample for
k >= 1; the k = 0 samples of some pixels sit exactly on a pixel border and reach two pixels, so k = 0 is skipped);
oracle: oracle_li of the same (pixel, k). The depth at which a sample first differs comes from repeating both with
maxdepth = 0, 1, 2, ..."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pbrt_v3_spectral_amd as pt, oracle_binding as ob, scenes_text as st

ob.set_libm(int(os.environ.get("ORACLE_LIBM", "1")))   # 1: correctly rounded libm in the oracle, like the device
tmp = tempfile.mkdtemp()
st.write_env_pfm(os.path.join(tmp, "env.pfm"))


def mk(name, depth, spp):
    if name.startswith("random:"):   # random:<seed>:<res>  (tests/scenes_text.py random_scene, as tools/fuzz_parity.py draws them)
        _, seed, res = name.split(":")
        st.write_texture_files(tmp); st.write_alpha_png(tmp)
        return pt.Scene(text=st.random_scene(int(seed), res=int(res), spp=spp), base_dir=tmp, max_depth=depth)
    if name in ("rough", "rough_lens", "inst", "inst_lens"):   # tests/scenes_text.py roughness_scene / instanced_scene
        st.write_texture_files(tmp); st.write_alpha_png(tmp)
        gen = st.roughness_scene if name.startswith("rough") else st.instanced_scene
        return pt.Scene(text=gen(spp=spp, lens=name.endswith("_lens")), base_dir=tmp, max_depth=depth)
    if name.startswith("zoo_"):
        return pt.Scene(text=st.material_zoo(res=96, spp=spp, depth=6, strategy=name[4:]), max_depth=depth)
    if name == "cornell128":
        return pt.Scene(os.path.join(ROOT, "scenes", "cornell-glass.pbrt"), spp=spp, xres=128, yres=128, max_depth=depth)
    if name.startswith("env_map_"):
        return pt.Scene(text=st.zoo_with_infinite_light("map", strategy=name[8:]), base_dir=tmp, max_depth=depth)


def per_sample(s, ks):
    integ = pt.CreatePathIntegrator(s)
    w, h = s.film_size
    ys, xs = np.mgrid[0:h, 0:w]
    out = {}
    for k in ks:
        f1, _ = integ.Render(spp=1, sample_begin=k)
        samples = np.stack([xs.ravel(), ys.ravel(), np.full(xs.size, k)], axis=1)
        li, _ = ob.li(s, samples)
        li = li.reshape(h, w, -1)
        d = np.abs(f1.astype(np.float64) - li).max(axis=2)
        scale = np.maximum(np.abs(li).max(axis=2), np.abs(f1).max(axis=2))
        for y, x in np.argwhere(d > 1e-4 * np.maximum(scale, 1e-6)):
            out[(int(x), int(y), k)] = (float(f1[y, x].sum()), float(li[y, x].sum()))
    return out


if __name__ != "__main__":
    names = []
else:
    names = sys.argv[1:] or ["zoo_spatial", "zoo_power", "zoo_uniform", "cornell128", "env_map_spatial", "env_map_power"]
for name in names:
    spp = 16
    full_depth = int(mk(name, -1, spp).desc.integrator.max_depth)
    ks = list(range(1, spp))
    bad = per_sample(mk(name, -1, spp), ks)
    n = len(ks) * mk(name, -1, spp).film_size[0] * mk(name, -1, spp).film_size[1]
    print("%-18s maxdepth %d: %d of %d samples differ (%.2e)" % (name, full_depth, len(bad), n, len(bad) / n))
    first = {}
    for depth in range(0, full_depth + 1):
        b = per_sample(mk(name, depth, spp), ks)
        for key in b:
            first.setdefault(key, depth)
    hist = {}
    for key in bad:
        hist[first.get(key, -1)] = hist.get(first.get(key, -1), 0) + 1
    print("     first maxdepth at which the sample differs -> count:", dict(sorted(hist.items())))
    for key in list(bad)[:12]:
        print("     pixel (%d,%d) k=%d: device sum %.6g oracle sum %.6g, differs from maxdepth %s" % (key + bad[key] + (first.get(key),)))
