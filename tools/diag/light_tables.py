"""Diagnostic (GPU box): spatial light-selection tables of the device against the oracle, voxel by voxel."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pbrt_v3_spectral_amd as pt, oracle_binding as ob, scenes_text as st

tmp = tempfile.mkdtemp()
st.write_env_pfm(os.path.join(tmp, "env.pfm"))
scenes = {
    "zoo": pt.Scene(text=st.material_zoo(res=96, spp=32, depth=6, strategy="spatial")),
    "cornell": pt.Scene(os.path.join(ROOT, "scenes", "cornell-glass.pbrt"), spp=16, xres=128, yres=128),
    "env_map": pt.Scene(text=st.zoo_with_infinite_light("map", strategy="spatial"), base_dir=tmp),
}
for name, s in scenes.items():
    integ = pt.CreatePathIntegrator(s)
    df, dfi = integ.light_distribution()
    of, ofi = ob.light_table(s)
    same = (df.view(np.uint32) == of.view(np.uint32)) | (np.isnan(df) & np.isnan(of))
    print(name, df.shape, "func entries differing: %d of %d" % ((~same).sum(), same.size),
          "voxels differing: %d of %d" % ((~same.all(axis=-1)).sum(), same[..., 0].size))
    bad = np.argwhere(~same)
    per_light = [(int((~same[..., j]).sum())) for j in range(same.shape[-1])]
    print("   per light:", per_light, "light types", [s.desc.lights[j].type for j in range(s.desc.n_lights)])
    for b in bad[:6]:
        d, o = df[tuple(b)], of[tuple(b)]
        print("   ", b, d, o, "ulps", int(d.view(np.uint32)) - int(o.view(np.uint32)) if np.isfinite(d) and np.isfinite(o) else "nan/inf")
    fsame = (dfi.view(np.uint32) == ofi.view(np.uint32)) | (np.isnan(dfi) & np.isnan(ofi))
    print("   funcInt differing: %d" % (~fsame).sum())
