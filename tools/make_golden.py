#!/usr/bin/env python3
"""Write the committed golden fixtures under tests/golden/ with the CPU oracle (oracle/, the restatement
of the reference algorithm pinned by the reference's own statistics, see tests/test_oracle_pins.py).
They let the GPU box check the HIP path at BASELINE config sizes without spending CPU minutes there, and
they pin the oracle itself against regressions.

  killeroo_1024spp_crop.npz   film sum [40,40,31] + weights of the 1024-spp frame (BASELINE configs[1]) on a
                              40x40 crop window of the 700x700 film that straddles the killeroo's silhouette,
                              and the oracle's counters for it
  cornell_256spp_crop.npz     same for scenes/cornell-glass.pbrt (maxdepth 8) at 256 spp on a 32x32 crop over
                              the glass sphere's caustic
  killeroo_rays.npz           4096 camera + 4096 random rays with closest-hit (prim, t, b0, b1) and any-hit results
  textured_zoo_64spp.npz      the image-textured material zoo of tests/scenes_text.py (64x64, 64 spp): image textures of
                              three file formats, checkerboards, bump maps, textured spheres

Usage: python tools/make_golden.py   (a few minutes on 8 cores)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pbrt_v3_spectral_amd as pt          # noqa: E402
import oracle_binding as ob                # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
KILLEROO = os.path.join(ROOT, "scenes", "killeroo-simple.pbrt")
CORNELL = os.path.join(ROOT, "scenes", "cornell-glass.pbrt")
KILLEROO_CROP = (0.30, 0.30 + 40 / 700, 0.44, 0.44 + 40 / 700)
CORNELL_CROP = (0.25, 0.25 + 32 / 512, 0.70, 0.70 + 32 / 512)


def film_fixture(path, scene_file, spp, crop):
    s = pt.Scene(scene_file, spp=spp, crop=crop)
    film, weight, c, secs = ob.render(s)
    d = c.as_dict()
    np.savez_compressed(path, film=film, weight=weight, crop=np.array(crop), spp=spp,
                        counters=np.array([d[k] for k in sorted(d)], np.int64), counter_names=np.array(sorted(d)))
    print("%s: film %s mean/spp %.6f, %d camera rays, %.1f s" % (os.path.basename(path), film.shape, film.mean() / spp,
                                                                 d["camera_rays"], secs))


def ray_fixture(path):
    s = pt.Scene(KILLEROO, spp=1)
    rng = np.random.default_rng(11)
    n = 4096
    samples = np.stack([rng.integers(0, 700, n), rng.integers(0, 700, n), np.zeros(n, int)], axis=1)
    cam = ob.camera_rays(s, samples)
    o = rng.uniform(-300, 300, (n, 3)).astype(np.float32) + np.array([0, 60, -100], np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    tmax = np.where(rng.random(n) < 0.5, np.inf, rng.uniform(10, 400, n)).astype(np.float32)
    rays = np.concatenate([cam, np.concatenate([o, d, tmax[:, None]], axis=1)]).astype(np.float32)
    closest, _ = ob.trace(s, rays, any_hit=False)
    anyhit, _ = ob.trace(s, rays, any_hit=True)
    np.savez_compressed(path, rays=rays, closest=closest.view(np.int32), anyhit=anyhit.view(np.int32)[:, 0])
    print("%s: %d rays, %d closest hits, %d occluded" % (os.path.basename(path), len(rays),
                                                        (closest.view(np.int32)[:, 0] >= 0).sum(), (anyhit.view(np.int32)[:, 0] >= 0).sum()))


def text_scene_fixture(path, which, res, spp):
    """A fixture of one of the authored test scenes that need generated image files (tests/scenes_text.py)."""
    import tempfile
    import scenes_text as st
    d = tempfile.mkdtemp()
    st.write_texture_files(d)
    st.write_alpha_png(d)
    text = {"textured_zoo": st.textured_zoo, "bump_scene": st.bump_scene, "alpha_scene": st.alpha_scene}[which](res=res, spp=spp)
    s = pt.Scene(text=text, base_dir=d)
    assert s.errors == [], s.errors
    film, weight, c, secs = ob.render(s)
    dd = c.as_dict()
    np.savez_compressed(path, film=film, weight=weight, spp=spp, res=res, scene=which,
                        counters=np.array([dd[k] for k in sorted(dd)], np.int64), counter_names=np.array(sorted(dd)))
    print("%s: film %s mean/spp %.6f, %.1f s" % (os.path.basename(path), film.shape, film.mean() / spp, secs))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    text_scene_fixture(os.path.join(OUT, "textured_zoo_64spp.npz"), "textured_zoo", 64, 64)
    if "--textured-only" in sys.argv:
        sys.exit(0)
    ray_fixture(os.path.join(OUT, "killeroo_rays.npz"))
    film_fixture(os.path.join(OUT, "cornell_256spp_crop.npz"), CORNELL, 256, CORNELL_CROP)
    film_fixture(os.path.join(OUT, "killeroo_1024spp_crop.npz"), KILLEROO, 1024, KILLEROO_CROP)
