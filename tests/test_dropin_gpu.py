"""The drop-in surface itself on the GPU: the `pbrt_amd` command line (the reference's `pbrt scene.pbrt`,
src/main/pbrt.cpp:83-139) and the Integrator-shaped C entry `mi_integrator_render` (what `integrator->Render(*scene)`
does at src/core/api.cpp:1707), both ending in the spectral `.dat` file Film::WriteImage writes
(src/core/film.cpp:226-308; integrator.cpp:341). The other GPU tests drive `mi_pt_*` from Python; these two go the
way a user of the reference goes: scene file in, film file + statistics out.
"""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import KILLEROO, ROOT

pytestmark = pytest.mark.gpu

PKG = os.path.join(ROOT, "pbrt-v3-spectral_amd")


def _read_dat_raw(path, w, h):
    """The file layout of film.cpp:226-308 read by hand: header "<w> <h> 31\\nv3 \\n", then 31 planes of h*w float64."""
    raw = open(path, "rb").read()
    header = ("%d %d 31\nv3 \n" % (w, h)).encode()
    assert raw.startswith(header), raw[:24]
    assert len(raw) == len(header) + w * h * 31 * 8, len(raw)
    planes = np.frombuffer(raw, "<f8", offset=len(header)).reshape(31, h, w)
    return np.ascontiguousarray(planes.transpose(1, 2, 0))


def _oracle(pt, ob, spp):
    s = pt.Scene(KILLEROO, spp=spp)
    with ob.exact_libm():
        ofilm, oweight, oc, _ = ob.render(s)
    return s, ofilm, oc.as_dict()


def _check_film(film64, ofilm, spp):
    # the file holds the un-normalised sums (mean 2.06 per sample for this scene), as float64 of the
    # float32 accumulators: equal to the oracle's film within float accumulation order
    assert film64.dtype == np.float64 and np.array_equal(film64, film64.astype(np.float32))
    d = film64 - ofilm
    rel = float(np.sqrt((d ** 2).sum() / (ofilm.astype(np.float64) ** 2).sum()))
    assert rel < 1e-6, rel
    assert abs(film64.mean() / spp - 2.06) < 0.02


def test_pbrt_amd_command_line_writes_the_reference_dat_file(pt, ob, tmp_path):
    """`pbrt_amd scenes/killeroo-simple.pbrt --spp 4 --outfile out.dat` in a fresh process: exit code 0, the 700 x 700 x 31
    plane-major float64 file, and the printed statistics equal to the oracle's counters."""
    out = tmp_path / "killeroo.dat"
    exe = os.path.join(PKG, "pbrt_amd")
    r = subprocess.run([exe, KILLEROO, "--spp", "4", "--outfile", str(out)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    s, ofilm, o = _oracle(pt, ob, 4)
    assert s.film_size == (700, 700)
    raw = open(out, "rb").read()
    assert raw[:15] == b"700 700 31\nv3 \n" and len(raw) == 15 + 700 * 700 * 31 * 8
    film = _read_dat_raw(str(out), 700, 700)
    _check_film(film, ofilm, 4)
    # the front end's own reader agrees with the hand-read layout
    assert np.array_equal(pt.read_dat(str(out)), film.astype(np.float32))
    # Statistics with the reference's STAT names
    stats = {m.group(1).strip(): m.group(2) for m in re.finditer(r"^\s+([A-Za-z/\- ]+?)\s+(\d[\d /]*)$", r.stdout, re.M)}
    assert int(stats["Integrator/Camera rays traced"]) == o["camera_rays"] == 700 * 700 * 4
    assert abs(int(stats["Intersections/Regular ray intersection tests"]) - o["regular_rays"]) <= 2
    assert abs(int(stats["Intersections/Shadow ray intersection tests"]) - o["shadow_rays"]) <= 2
    zero, total = [int(v) for v in stats["Integrator/Zero-radiance paths"].split("/")]
    assert abs(zero - o["zero_radiance_paths"]) <= 2 and abs(total - o["total_paths"]) <= 2
    # ... which in the reference's own libm are the reference's numbers (8 435 510 + 3 077 259)
    assert abs(int(stats["Intersections/Regular ray intersection tests"]) - 8435510) <= 1e-4 * 8435510
    assert abs(int(stats["Intersections/Shadow ray intersection tests"]) - 3077259) <= 1e-4 * 3077259


def test_mi_integrator_render_writes_the_film_and_returns_the_counters(pt, ob, tmp_path):
    """The C entry a host `Integrator::Render` calls: scene handle in, `.dat` file written by the library, counters back."""
    s, ofilm, o = _oracle(pt, ob, 2)
    out = tmp_path / "from_capi.dat"
    c = pt.Counters()
    rc = pt.host_lib().mi_integrator_render(s._h, 0, os.fsencode(str(out)), C.byref(c))
    assert rc == 0, pt.host_lib().mi_scene_last_error()
    d = c.as_dict()
    assert d["camera_rays"] == o["camera_rays"] == 700 * 700 * 2 and d["bad_samples"] == 0
    for k in ("regular_rays", "shadow_rays", "total_paths", "zero_radiance_paths", "path_length_sum"):
        assert abs(d[k] - o[k]) <= 2, (k, d[k], o[k])
    _check_film(_read_dat_raw(str(out), 700, 700), ofilm, 2)
    # without an explicit name the file goes where the scene's Film says (killeroo-simple.pbrt: "killeroo-simple.exr" ->
    # ".dat" beside it, film.cpp:236-240), relative to the working directory like the reference
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        assert pt.host_lib().mi_integrator_render(s._h, 0, None, C.byref(c)) == 0
        assert os.path.exists(os.path.splitext(s.film_filename)[0] + ".dat")
    finally:
        os.chdir(cwd)
    # a device that does not exist is an error code, not an abort
    assert pt.host_lib().mi_integrator_render(s._h, 12345, os.fsencode(str(out)), C.byref(c)) != 0
