"""Accelerator "bvh" "string splitmethod" "hlbvh" (BVHAccel::HLBVHBuild, src/accelerators/bvh.cpp:404-638).

CPU: the host restatement builds a valid tree (every primitive in exactly one leaf, every node's box
containing its children's, leaf sizes below maxnodeprims) whose closest hits on recorded rays are those of the SAH tree.
GPU: the device build (mi_bvh_build_hlbvh: Morton codes, stable split sort, one lane per treelet for emitLBVH, flatten)
returns the same node array and primitive order as the host restatement, bit for bit; the render under it matches the oracle."""
import os
import sys

import numpy as np
import pytest

from conftest import KILLEROO, ROOT
import scenes_text as st


def _nodes(s):
    d = s.desc
    raw = np.ctypeslib.as_array(d.nodes, (d.n_nodes,))   # structured: bmin, bmax, offset, n_prims, axis, pad
    return raw


def _check_tree(s, max_prims):
    nodes = _nodes(s)
    n = len(nodes)
    bmin, bmax = np.array(nodes["bmin"]), np.array(nodes["bmax"])
    offset, nprims = np.array(nodes["offset"]), np.array(nodes["n_prims"])
    seen = np.zeros(s.desc.n_prims, int)
    next_leaf = 0
    stack = sorted({0} | {int(s.desc.instances[k].root) for k in range(s.desc.n_instances)})   # the world's tree and the objects'
    while stack:   # depth first: first child = i + 1, second = offset
        i = stack.pop()
        if nprims[i] > 0:
            next_leaf += int(nprims[i])           # (treelets keep Morton order inside; the SAH tree above reorders the treelets)
            seen[offset[i]:offset[i] + nprims[i]] += 1
            continue
        a, b = i + 1, int(offset[i])
        assert i < a < n and a < b < n
        for c in (a, b):
            assert (bmin[c] >= bmin[i]).all() and (bmax[c] <= bmax[i]).all()
        stack.append(b)
        stack.append(a)
    assert (seen == 1).all() and next_leaf == s.desc.n_prims
    leaves = nprims[nprims > 0]
    return int((nprims == 0).sum()), len(leaves), int(leaves.max())


def _with_hlbvh(text, maxprims=None):
    extra = ' "integer maxnodeprims" [%d]' % maxprims if maxprims else ""
    return text.replace("WorldBegin", 'Accelerator "bvh" "string splitmethod" "hlbvh"%s\nWorldBegin' % extra, 1)


def _rays(rng, s, n):
    d = s.desc
    lo = np.array([d.nodes[0].bmin[i] for i in range(3)], np.float32)
    hi = np.array([d.nodes[0].bmax[i] for i in range(3)], np.float32)
    o = (lo + (hi - lo) * rng.random((n, 3))).astype(np.float32)
    dr = rng.normal(size=(n, 3)).astype(np.float32)
    return np.concatenate([o, dr, np.full((n, 1), np.inf, np.float32)], axis=1).astype(np.float32)


def test_host_hlbvh_is_a_valid_tree_with_the_sah_trees_hits(pt, ob, monkeypatch):
    monkeypatch.setenv("MIPT_HLBVH", "host")
    text = open(KILLEROO).read()
    base = os.path.dirname(KILLEROO)
    sah = pt.Scene(text=text, base_dir=base, spp=1)
    hl = pt.Scene(text=_with_hlbvh(text), base_dir=base, spp=1)
    assert hl.errors == [] and not any("hlbvh" in w for w in hl.warnings) and hl.stats["accel_on_device"] == 0
    interior, leaves, biggest = _check_tree(hl, 4)
    assert interior == hl.stats["interior_nodes"] and leaves == hl.stats["leaf_nodes"] and interior == leaves - 1
    assert biggest < 4 or biggest <= 255      # emitLBVH: leaves of fewer than maxPrimsInNode primitives (more only when the code bits run out)
    # the same geometry: closest hits (t, barycentrics) on recorded rays equal the SAH tree's, primitive for primitive
    rng = np.random.default_rng(3)
    rays = _rays(rng, sah, 20000)
    a, _ = ob.trace(sah, rays)
    b, _ = ob.trace(hl, rays)
    assert np.array_equal(a[:, 1:].view(np.int32), b[:, 1:].view(np.int32))
    hit = a.view(np.int32)[:, 0] >= 0
    assert np.array_equal(hit, b.view(np.int32)[:, 0] >= 0) and hit.mean() > 0.3
    # (primitive numbers differ -- another leaf order -- but they name the same shapes)
    pa = np.array([sah.desc.prims[int(i)].shape for i in a.view(np.int32)[hit, 0]])
    pb = np.array([hl.desc.prims[int(i)].shape for i in b.view(np.int32)[hit, 0]])
    assert np.array_equal(pa, pb)
    # and the render under it is the SAH render up to the order the leaves are met in
    small_sah = pt.Scene(text=text, base_dir=base, spp=2, xres=64, yres=64)
    small_hl = pt.Scene(text=_with_hlbvh(text), base_dir=base, spp=2, xres=64, yres=64)
    fa, wa, ca, _ = ob.render(small_sah, n_threads=4)
    fb, wb, cb, _ = ob.render(small_hl, n_threads=4)
    assert ca.camera_rays == cb.camera_rays and np.array_equal(wa, wb)
    assert np.sqrt(((fa.astype(np.float64) - fb) ** 2).sum() / (fa.astype(np.float64) ** 2).sum()) < 1e-3


def test_host_hlbvh_on_random_scenes_and_maxnodeprims(pt, ob, tmp_path, monkeypatch):
    monkeypatch.setenv("MIPT_HLBVH", "host")
    st.write_texture_files(str(tmp_path))
    st.write_alpha_png(str(tmp_path))
    for seed in range(6):
        s = pt.Scene(text=_with_hlbvh(st.random_scene(seed), maxprims=2 + seed % 3), base_dir=str(tmp_path))
        assert s.errors == []
        _check_tree(s, 2 + seed % 3)
        f, w, c, _ = ob.render(s, n_threads=4)
        assert np.isfinite(f).all()


@pytest.mark.gpu
def test_device_hlbvh_equals_the_host_tree_node_for_node(pt, ob, tmp_path, monkeypatch):
    """mi_bvh_build_hlbvh against the host restatement: killeroo (66 533 primitives), fuzz scenes, and the 10 000 002-triangle
    scene (build time reported; target < 0.5 s)."""
    st.write_texture_files(str(tmp_path))
    st.write_alpha_png(str(tmp_path))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_procedural_scene as mps
    big = tmp_path / "proc.pbrt"
    with open(big, "w") as fh:
        mps.write_scene(fh, 10_000_000, 64, 1, 7, 5)
    cases = [("killeroo", dict(text=_with_hlbvh(open(KILLEROO).read()), base_dir=os.path.dirname(KILLEROO), spp=1))]
    cases += [("random %d" % k, dict(text=_with_hlbvh(st.random_scene(k), maxprims=2 + k % 3), base_dir=str(tmp_path))) for k in range(4)]
    cases.append(("procedural 10M", dict(text=_with_hlbvh(open(big).read()), base_dir=str(tmp_path))))
    for name, kw in cases:
        monkeypatch.setenv("MIPT_HLBVH", "host")
        host = pt.Scene(**kw)
        monkeypatch.delenv("MIPT_HLBVH")
        dev = pt.Scene(**kw)
        assert host.stats["accel_on_device"] == 0 and dev.stats["accel_on_device"] == 1, name
        assert dev.desc.n_nodes == host.desc.n_nodes and dev.desc.n_prims == host.desc.n_prims, name
        assert np.array_equal(_nodes(dev).view(np.uint8), _nodes(host).view(np.uint8)), name
        pd = np.ctypeslib.as_array(dev.desc.prims, (dev.desc.n_prims,))
        ph = np.ctypeslib.as_array(host.desc.prims, (host.desc.n_prims,))
        assert np.array_equal(pd.view(np.uint8), ph.view(np.uint8)), name


@pytest.mark.gpu
def test_render_under_the_device_built_hlbvh_matches_the_oracle(pt, ob):
    """The HIP path traversing a device-built HLBVH against the oracle traversing the same tree: exact-mode parity."""
    text = _with_hlbvh(open(KILLEROO).read())
    s = pt.Scene(text=text, base_dir=os.path.dirname(KILLEROO), spp=4, xres=200, yres=200)
    assert s.stats["accel_on_device"] == 1
    integ = pt.CreatePathIntegrator(s)
    film, weight = integ.Render()
    with ob.exact_libm():
        ofilm, oweight, oc, _ = ob.render(s)
    c, o = integ.counters.as_dict(), oc.as_dict()
    for k in ("camera_rays", "regular_rays", "shadow_rays", "total_paths", "zero_radiance_paths", "path_length_sum"):
        assert abs(c[k] - o[k]) <= 2, (k, c[k], o[k])
    assert np.array_equal(weight, oweight)
    d = film.astype(np.float64) - ofilm
    assert np.sqrt((d ** 2).sum() / (ofilm.astype(np.float64) ** 2).sum()) < 1e-6
