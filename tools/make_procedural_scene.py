#!/usr/bin/env python3
"""Seeded procedural stand-in for BASELINE.json configs 4/5 (San Miguel is not available
offline): many small, randomly oriented and sized triangle meshes
(bumpy icospheres, 320 triangles each) scattered through a slab above a ground quad, a
matte / plastic / uber material mix, one distant light plus a few spherical area lights.
At the default 10 M triangles the BVH + triangle records are ~1.1 GB on the device, far
beyond L2 and the 256 MB Infinity Cache, so traversal has to go to HBM.

  tools/make_procedural_scene.py --tris 10000000 --out /tmp/proc10m.pbrt [--res 700] [--spp 256]
"""
import argparse
import io
import sys

import numpy as np


def icosphere(level):
    t = (1 + 5 ** 0.5) / 2
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], np.float64)
    v /= np.linalg.norm(v, axis=1)[:, None]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6),
         (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7),
         (9, 8, 1)]
    verts = [tuple(x) for x in v]
    for _ in range(level):
        cache, nf = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = np.array(verts[a]) + np.array(verts[b])
                m /= np.linalg.norm(m)
                verts.append(tuple(m))
                cache[key] = len(verts) - 1
            return cache[key]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return np.array(verts, np.float64), np.array(f, np.int32)


def random_rotations(rng, n):
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1)[:, None]
    w, x, y, z = q.T
    return np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], 1),
                     np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], 1),
                     np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1)], 1)


MATERIALS = [
    'MakeNamedMaterial "m0" "string type" "matte" "rgb Kd" [.7 .7 .7]',
    'MakeNamedMaterial "m1" "string type" "matte" "rgb Kd" [.7 .3 .2] "float sigma" [20]',
    'MakeNamedMaterial "m2" "string type" "matte" "rgb Kd" [.2 .5 .7]',
    'MakeNamedMaterial "m3" "string type" "plastic" "rgb Kd" [.3 .6 .3] "rgb Ks" [.4 .4 .4] "float roughness" [.05]',
    'MakeNamedMaterial "m4" "string type" "plastic" "rgb Kd" [.6 .5 .2] "rgb Ks" [.3 .3 .3] "float roughness" [.2]',
    'MakeNamedMaterial "m5" "string type" "uber" "rgb Kd" [.4 .3 .5] "rgb Ks" [.3 .3 .3] "rgb Kr" [.1 .1 .1] "float roughness" [.1]',
    'MakeNamedMaterial "m6" "string type" "matte" "rgb Kd" [.8 .75 .6]',
    'MakeNamedMaterial "m7" "string type" "plastic" "rgb Kd" [.15 .15 .5] "rgb Ks" [.5 .5 .5] "float roughness" [.02]',
]


def write_scene(out, n_tris, res, spp, seed, depth):
    rng = np.random.default_rng(seed)
    verts, faces = icosphere(2)                 # 162 vertices, 320 triangles
    n_blobs = max(1, n_tris // len(faces))
    extent = 60.0 * (n_blobs / 31250.0) ** (1.0 / 3.0) + 4.0   # keep the blob density constant
    out.write('# procedural scene: %d blobs x %d triangles = %d triangles, seed %d\n'
              % (n_blobs, len(faces), n_blobs * len(faces), seed))
    out.write('LookAt %g %g %g  0 %g 0  0 1 0\n' % (1.2 * extent, 1.3 * extent, 1.4 * extent, 0.1 * extent))
    out.write('Camera "perspective" "float fov" [40]\n')
    out.write('Film "image" "integer xresolution" [%d] "integer yresolution" [%d] "string filename" "procedural.exr"\n'
              % (res, res))
    out.write('Sampler "halton" "integer pixelsamples" [%d]\n' % spp)
    out.write('Integrator "path" "integer maxdepth" [%d]\n' % depth)
    out.write('WorldBegin\n')
    out.write('LightSource "distant" "point from" [0.3 1 0.2] "point to" [0 0 0] "rgb L" [2.5 2.4 2.2]\n')
    for k in range(4):
        ang = k * np.pi / 2 + 0.3
        out.write('AttributeBegin\n  AreaLightSource "diffuse" "rgb L" [60 55 45]\n  Translate %g %g %g\n'
                  '  Shape "sphere" "float radius" [%g]\nAttributeEnd\n'
                  % (0.9 * extent * np.cos(ang), 0.75 * extent, 0.9 * extent * np.sin(ang), 0.04 * extent))
    for m in MATERIALS:
        out.write(m + '\n')
    g = 12 * extent
    out.write('NamedMaterial "m6"\nShape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" '
              '[%g 0 %g  %g 0 %g  %g 0 %g  %g 0 %g]\n' % (-g, -g, -g, g, g, g, g, -g))
    idx_text = ' '.join(map(str, faces.reshape(-1)))
    rot = random_rotations(rng, n_blobs)
    centre = np.stack([rng.uniform(-extent, extent, n_blobs), rng.uniform(0.02, 0.7, n_blobs) * extent,
                       rng.uniform(-extent, extent, n_blobs)], 1)
    radius = 0.35 * np.exp(rng.normal(0.0, 0.6, n_blobs))          # log-normal sizes
    mat = rng.integers(0, len(MATERIALS), n_blobs)
    chunk = 512
    for c0 in range(0, n_blobs, chunk):
        c1 = min(n_blobs, c0 + chunk)
        n = c1 - c0
        bump = 1.0 + 0.25 * rng.uniform(-1, 1, (n, len(verts), 1))
        stretch = np.exp(rng.normal(0, 0.35, (n, 1, 3)))
        p = verts[None] * bump * stretch * radius[c0:c1, None, None]
        p = np.einsum('nij,nvj->nvi', rot[c0:c1], p) + centre[c0:c1, None, :]
        buf = io.StringIO()
        for i in range(n):
            buf.write('NamedMaterial "m%d"\nShape "trianglemesh" "integer indices" [%s] "point P" [' % (mat[c0 + i], idx_text))
            buf.write(' '.join('%.5g' % x for x in p[i].reshape(-1)))
            buf.write(']\n')
        out.write(buf.getvalue())
    out.write('WorldEnd\n')
    return n_blobs * len(faces) + 2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tris", type=int, default=10_000_000)
    ap.add_argument("--res", type=int, default=700)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--depth", type=int, default=5)
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--out", default="-")
    a = ap.parse_args()
    out = sys.stdout if a.out == "-" else open(a.out, "w")
    n = write_scene(out, a.tris, a.res, a.spp, a.seed, a.depth)
    if out is not sys.stdout:
        out.close()
        print("wrote %s: %d triangles" % (a.out, n), file=sys.stderr)


if __name__ == "__main__":
    main()
