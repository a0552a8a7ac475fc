"""Small scene descriptions used by the tests (authored here, .pbrt syntax)."""

_HEAD = """
LookAt 0 0 0  0 0 1  0 1 0
Camera "perspective" "float fov" [45]
Film "image" "integer xresolution" [%(res)d] "integer yresolution" [%(res)d]
Sampler "halton" "integer pixelsamples" [%(spp)d]
Integrator "path" "integer maxdepth" [%(depth)d] %(extra)s
WorldBegin
"""


def _c(v):
    return "[300 %g 800 %g]" % (v, v)


def furnace_point(res=10, spp=256, depth=8, n_lights=1, extra=""):
    """tests/analytic_scenes.cpp:71-133: unit inward-facing sphere, Kd=0.5, point light(s)
    of total intensity pi at the centre -> radiance 1."""
    s = _HEAD % dict(res=res, spp=spp, depth=depth, extra=extra)
    for _ in range(n_lights):
        s += 'LightSource "point" "spectrum I" %s\n' % _c(3.14159265358979 / n_lights)
    s += 'Material "matte" "spectrum Kd" %s\nReverseOrientation\nShape "sphere" "float radius" [1]\nWorldEnd\n' % _c(0.5)
    return s


def furnace_area(res=10, spp=256, depth=8):
    """tests/analytic_scenes.cpp:135-165: Kd=0.5 emissive sphere Le=0.5 -> radiance 1."""
    s = _HEAD % dict(res=res, spp=spp, depth=depth, extra="")
    s += 'Material "matte" "spectrum Kd" %s\nReverseOrientation\n' % _c(0.5)
    s += 'AreaLightSource "diffuse" "spectrum L" %s\nShape "sphere" "float radius" [1]\nWorldEnd\n' % _c(0.5)
    return s


def furnace_uber(res=10, spp=256, depth=8):
    """tests/analytic_scenes.cpp:167-203: UberMaterial Kd=.25 Kr=.5 (eta 1), I=3pi -> ~1."""
    s = _HEAD % dict(res=res, spp=spp, depth=depth, extra="")
    s += 'LightSource "point" "spectrum I" %s\n' % _c(3 * 3.14159265358979)
    s += ('Material "uber" "spectrum Kd" %s "spectrum Ks" %s "spectrum Kr" %s "spectrum Kt" %s '
          '"float roughness" [0] "float index" [1] "bool remaproughness" ["false"]\n' % (_c(.25), _c(0), _c(.5), _c(0)))
    s += 'ReverseOrientation\nShape "sphere" "float radius" [1]\nWorldEnd\n'
    return s


MATERIAL_ZOO = """
LookAt 0 3 -8  0 0.6 0  0 1 0
Camera "perspective" "float fov" [40]
Film "image" "integer xresolution" [%(res)d] "integer yresolution" [%(res)d]
Sampler "halton" "integer pixelsamples" [%(spp)d]
Integrator "path" "integer maxdepth" [%(depth)d] "string lightsamplestrategy" "%(strategy)s"
WorldBegin
AttributeBegin
  AreaLightSource "diffuse" "rgb L" [18 17 15]
  Translate 0 5 0
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-1.2 0 -1.2  1.2 0 -1.2  1.2 0 1.2  -1.2 0 1.2]
AttributeEnd
LightSource "point" "rgb I" [6 6 8] "point from" [-4 3 -3]
LightSource "distant" "rgb L" [.4 .4 .5] "point from" [0 10 -4] "point to" [0 0 0]
LightSource "spot" "rgb I" [40 36 30] "point from" [3 4 -4] "point to" [0.5 0 -2] "float coneangle" [25] "float conedeltaangle" [8]
AttributeBegin
  Material "matte" "rgb Kd" [.6 .6 .6] "float sigma" [20]
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-8 0 -8  -8 0 8  8 0 8  8 0 -8]
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-8 0 6  -8 8 6  8 8 6  8 0 6]
AttributeEnd
AttributeBegin
  Material "glass" "float index" [1.5]
  Translate -2.4 0.8 0
  Shape "sphere" "float radius" [.8]
AttributeEnd
AttributeBegin
  Material "uber" "rgb Kd" [.3 .1 .1] "rgb Ks" [.4 .4 .4] "rgb Kr" [.2 .2 .2] "float roughness" [.05] "rgb opacity" [.8 .8 .8]
  Translate -0.8 0.8 0
  Shape "sphere" "float radius" [.8]
AttributeEnd
AttributeBegin
  Material "disney" "rgb color" [.2 .5 .3] "float metallic" [.4] "float roughness" [.3] "float sheen" [.5] "float clearcoat" [.6] "float anisotropic" [.3]
  Translate 0.8 0.8 0
  Shape "sphere" "float radius" [.8]
AttributeEnd
AttributeBegin
  Material "disney" "rgb color" [.7 .6 .2] "float spectrans" [.6] "float roughness" [.2] "bool thin" ["true"] "float flatness" [.3] "float difftrans" [.8]
  Translate 2.4 0.8 0
  Shape "sphere" "float radius" [.8]
AttributeEnd
AttributeBegin
  Material "glass" "float index" [1.4] "float uroughness" [.1] "float vroughness" [.2] "rgb Kt" [.9 .9 1]
  Translate 0 0.5 -2.2
  Shape "sphere" "float radius" [.5]
AttributeEnd
AttributeBegin
  Material "mirror" "rgb Kr" [.8 .8 .8]
  Translate 1.6 0.5 -2.2
  Shape "sphere" "float radius" [.5]
AttributeEnd
AttributeBegin
  Material "metal" "float roughness" [.2]
  Translate -1.4 0.35 -3.6
  Shape "sphere" "float radius" [.35]
AttributeEnd
AttributeBegin
  Material "metal" "spectrum eta" [400 1.5 550 0.9 700 0.3] "spectrum k" [400 2.0 550 2.6 700 4.1] "float uroughness" [.05] "float vroughness" [.3]
  Translate -0.7 0.35 -3.6
  Shape "sphere" "float radius" [.35]
AttributeEnd
AttributeBegin
  Material "substrate" "rgb Kd" [.5 .2 .1] "rgb Ks" [.3 .3 .3] "float uroughness" [.1] "float vroughness" [.25]
  Translate 0 0.35 -3.6
  Shape "sphere" "float radius" [.35]
AttributeEnd
AttributeBegin
  Material "translucent" "rgb Kd" [.5 .5 .3] "rgb Ks" [.3 .3 .3] "rgb reflect" [.6 .6 .6] "rgb transmit" [.4 .4 .4] "float roughness" [.15]
  Translate 0.7 0.35 -3.6
  Shape "sphere" "float radius" [.35]
AttributeEnd
MakeNamedMaterial "mixA" "string type" "plastic" "rgb Kd" [.7 .1 .1] "rgb Ks" [.3 .3 .3] "float roughness" [.1]
MakeNamedMaterial "mixB" "string type" "mirror" "rgb Kr" [.9 .9 .9]
AttributeBegin
  Material "mix" "string namedmaterial1" "mixA" "string namedmaterial2" "mixB" "rgb amount" [.7 .5 .3]
  Translate 1.4 0.35 -3.6
  Shape "sphere" "float radius" [.35]
AttributeEnd
AttributeBegin
  Material "plastic" "rgb Kd" [.1 .2 .6] "rgb Ks" [.5 .5 .5] "float roughness" [.08]
  Translate -1.6 0 -2.2
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3 0 3 1 1 3 2]
     "point P" [0 0 0  .9 0 0  .45 0 .8  .45 .9 .3] "normal N" [-.5 -.3 -.4  .6 -.3 -.4  0 -.2 .8  0 1 0]
AttributeEnd
WorldEnd
"""


def material_zoo(res=96, spp=16, depth=6, strategy="spatial"):
    return MATERIAL_ZOO % dict(res=res, spp=spp, depth=depth, strategy=strategy)


def write_env_pfm(path, w=40, h=24, seed=5):
    """A small procedural environment map (non-power-of-two on purpose: the MIPMap resamples it) with a 'sun'."""
    import numpy as np
    rng = np.random.default_rng(seed)
    img = (rng.random((h, w, 3)) ** 3 * 4).astype(np.float32)
    img[3:6, 10:14] = [30, 25, 18]
    with open(path, "wb") as f:
        f.write(b"PF\n%d %d\n-1.0\n" % (w, h))
        f.write(img[::-1].tobytes())
    return img


def zoo_with_infinite_light(kind, res=64, spp=16, depth=6, strategy="power"):
    """The material zoo lit by (also) a LightSource "infinite": kind = "const" (constant L beside the other lights),
    "map" (rotated PFM map beside the other lights) or "only_env" (the map alone). The map file is env.pfm in base_dir."""
    zoo = material_zoo(res=res, spp=spp, depth=depth, strategy=strategy)
    if kind == "const":
        return zoo.replace('LightSource "point"', 'LightSource "infinite" "rgb L" [.4 .5 .7]\nLightSource "point"')
    if kind == "map":
        return zoo.replace('LightSource "point"', 'AttributeBegin\nRotate -90 1 0 0\nRotate 30 0 0 1\nLightSource "infinite" "rgb L" [.6 .6 .6] '
                           '"string mapname" "env.pfm"\nAttributeEnd\nLightSource "point"')
    t = zoo.replace('LightSource "point" "rgb I" [6 6 8] "point from" [-4 3 -3]', '')
    t = t.replace('LightSource "distant" "rgb L" [.4 .4 .5] "point from" [0 10 -4] "point to" [0 0 0]', '')
    t = t.replace('AreaLightSource "diffuse" "rgb L" [18 17 15]', '')
    return t.replace('WorldBegin', 'WorldBegin\nAttributeBegin\nRotate -90 1 0 0\nLightSource "infinite" "string mapname" "env.pfm"\nAttributeEnd')


# ------------------------------------------------------------------ image textures
def _texture_image(w, h, seed):
    """A procedural 8-bit RGB image with structure at several scales, saturated patches and a pure black region
    (where a textured lobe drops out of the BSDF)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.zeros((h, w, 3), np.float64)
    img[..., 0] = 0.5 + 0.5 * np.sin(x * 0.9) * np.cos(y * 0.35)
    img[..., 1] = ((x // 4 + y // 3) % 2) * 0.8 + 0.1
    img[..., 2] = rng.random((h, w))
    img[h // 3:h // 2, w // 4:w // 2] = [1.0, 0.05, 0.02]
    img[: h // 5, : w // 5] = 0.0
    return (np.clip(img, 0, 1) * 255 + 0.5).astype(np.uint8)


def write_png(path, rgb8, with_alpha=False, filters=True):
    """Minimal PNG writer (colour type 2 or 6, 8 bit); cycles through the five scanline filters."""
    import struct, zlib
    import numpy as np
    h, w, _ = rgb8.shape
    px = rgb8
    if with_alpha:
        px = np.concatenate([rgb8, np.full((h, w, 1), 200, np.uint8)], axis=2)
    bpp = px.shape[2]
    raw = bytearray()
    prev = np.zeros(w * bpp, np.int32)
    for yy in range(h):
        cur = px[yy].reshape(-1).astype(np.int32)
        ft = (yy % 5) if filters else 0
        left = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        upleft = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
        if ft == 0:
            out = cur
        elif ft == 1:
            out = cur - left
        elif ft == 2:
            out = cur - prev
        elif ft == 3:
            out = cur - (left + prev) // 2
        else:
            p = left + prev - upleft
            pa, pb, pc = abs(p - left), abs(p - prev), abs(p - upleft)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, upleft))
            out = cur - pred
        raw.append(ft)
        raw += (out & 255).astype(np.uint8).tobytes()
        prev = cur

    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6 if with_alpha else 2, 0, 0, 0)))
        comp = zlib.compress(bytes(raw), 6)
        half = len(comp) // 2
        f.write(chunk(b"IDAT", comp[:half]))
        f.write(chunk(b"IDAT", comp[half:]))
        f.write(chunk(b"IEND", b""))


def write_tga(path, rgb8, rle=False):
    """Uncompressed or run-length-encoded 24-bit TGA, bottom-up like most writers."""
    import struct
    h, w, _ = rgb8.shape
    with open(path, "wb") as f:
        f.write(struct.pack("<BBBHHBHHHHBB", 0, 0, 10 if rle else 2, 0, 0, 0, 0, 0, w, h, 24, 0))
        rows = rgb8[::-1, :, ::-1]   # bottom-up, BGR
        if not rle:
            f.write(rows.tobytes())
        else:
            flat = rows.reshape(-1, 3)
            i = 0
            while i < len(flat):
                run = 1
                while i + run < len(flat) and run < 128 and (flat[i + run] == flat[i]).all():
                    run += 1
                if run > 1:
                    f.write(bytes([0x80 | (run - 1)]) + flat[i].tobytes())
                    i += run
                else:
                    n = 1
                    while i + n < len(flat) and n < 128 and not (flat[i + n] == flat[i + n - 1]).all():
                        n += 1
                    f.write(bytes([n - 1]) + flat[i:i + n].tobytes())
                    i += n


def write_texture_files(base_dir):
    """The image files of textured_zoo(): a PNG (RGBA, all scanline filters), an RLE TGA, a non-power-of-two PFM."""
    import os
    import numpy as np
    a = _texture_image(32, 16, 1)
    write_png(os.path.join(base_dir, "tex_a.png"), a, with_alpha=True)
    b = _texture_image(24, 20, 2)          # not a power of two: the MIPMap resamples it
    write_tga(os.path.join(base_dir, "tex_b.tga"), b, rle=True)
    c = (_texture_image(12, 10, 3).astype(np.float32) / 255.0) ** 2 * 1.5
    with open(os.path.join(base_dir, "tex_c.pfm"), "wb") as f:
        f.write(b"PF\n%d %d\n-1.0\n" % (c.shape[1], c.shape[0]))
        f.write(c[::-1].astype(np.float32).tobytes())
    # a roughness map: strictly positive everywhere (a zero, used unremapped as an alpha, is a NaN in the reference too)
    r = 0.04 + 0.5 * (_texture_image(20, 12, 4).astype(np.float32) / 255.0)
    with open(os.path.join(base_dir, "rough.pfm"), "wb") as f:
        f.write(b"PF\n%d %d\n-1.0\n" % (r.shape[1], r.shape[0]))
        f.write(r[::-1].astype(np.float32).tobytes())
    return a, b, c


TEXTURED_ZOO = """
LookAt 0 2.2 -7  0 0.3 0  0 1 0
Camera "perspective" "float fov" [42] %(lens)s
Film "image" "integer xresolution" [%(res)d] "integer yresolution" [%(res)d]
Sampler "halton" "integer pixelsamples" [%(spp)d]
Integrator "path" "integer maxdepth" [%(depth)d]
WorldBegin
AttributeBegin
  AreaLightSource "diffuse" "rgb L" [16 15 13]
  Translate 0 5 0
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-1.5 0 -1.5  1.5 0 -1.5  1.5 0 1.5  -1.5 0 1.5]
AttributeEnd
LightSource "point" "rgb I" [10 10 12] "point from" [-3 3 -4]
Texture "ewa_png" "spectrum" "imagemap" "string filename" "tex_a.png" "float uscale" [6] "float vscale" [6]
Texture "tri_tga" "spectrum" "imagemap" "string filename" "tex_b.tga" "bool trilinear" ["true"] "float udelta" [.25]
Texture "pfm_clamp" "spectrum" "imagemap" "string filename" "tex_c.pfm" "string wrap" "clamp" "float uscale" [2] "float vscale" [2] "float scale" [.8]
Texture "png_black" "spectrum" "imagemap" "string filename" "tex_a.png" "string wrap" "black" "float uscale" [1.5] "float udelta" [-.2] "float maxanisotropy" [4]
Texture "tga_nofilt" "spectrum" "imagemap" "string filename" "tex_b.tga" "bool noFiltering" ["true"] "bool gamma" ["false"]
Texture "tinted" "spectrum" "scale" "texture tex1" "tri_tga" "rgb tex2" [.9 .6 .4]
Texture "sphere_bump_raw" "float" "imagemap" "string filename" "tex_b.tga" "float uscale" [3] "float vscale" [2]
Texture "sphere_bump" "float" "scale" "texture tex1" "sphere_bump_raw" "float tex2" [.05]
Texture "checks" "spectrum" "checkerboard" "float uscale" [7] "float vscale" [5] "rgb tex1" [.8 .75 .1] "rgb tex2" [0 0 0]
Texture "checks_pt" "spectrum" "checkerboard" "float uscale" [3] "float vscale" [3] "string aamode" "none" "rgb tex1" [.1 .2 .8] "rgb tex2" [.9 .9 .9]
# ground: matte, EWA-filtered at a grazing angle
AttributeBegin
  Material "matte" "texture Kd" "ewa_png"
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-6 0 -6  6 0 -6  6 0 6  -6 0 6] "float uv" [0 0 1 0 1 1 0 1]
AttributeEnd
# back wall: plastic with a textured Kd and a constant Ks
AttributeBegin
  Material "plastic" "texture Kd" "tinted" "rgb Ks" [.3 .3 .3] "float roughness" [.15]
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-4 0 4  4 0 4  4 4 4  -4 4 4] "float uv" [0 0 2 0 2 1 0 1]
AttributeEnd
# uber with textured Kd and Ks (three lobes) on a tilted panel without uv (default parametrisation)
AttributeBegin
  Material "uber" "texture Kd" "pfm_clamp" "texture Ks" "png_black" "rgb Kr" [.1 .1 .1] "float roughness" [.2]
  Translate -2.2 .8 0
  Rotate 35 0 1 0
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-1 -.8 0  1 -.8 0  1 .8 0  -1 .8 0]
AttributeEnd
# substrate, both spectra textured
AttributeBegin
  Material "substrate" "texture Kd" "tga_nofilt" "texture Ks" "pfm_clamp" "float uroughness" [.2] "float vroughness" [.1]
  Translate 0 .8 1
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-1 -.8 0  1 -.8 0  1 .8 0  -1 .8 0] "float uv" [0 0 1 0 1 1 0 1]
AttributeEnd
# mirror and translucent panels
AttributeBegin
  Material "uber" "texture Kd" "checks_pt" "texture Kr" "png_black" "rgb Ks" [0 0 0]
  Translate 2.2 .8 0
  Rotate -35 0 1 0
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-1 -.8 0  1 -.8 0  1 .8 0  -1 .8 0] "float uv" [0 0 1 0 1 1 0 1]
AttributeEnd
AttributeBegin
  Material "translucent" "texture Kd" "checks" "rgb Ks" [.2 .2 .2] "rgb reflect" [.4 .5 .4] "rgb transmit" [.5 .4 .5]
  Translate 0 2.6 2
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-1.5 -.5 0  1.5 -.5 0  1.5 .5 0  -1.5 .5 0] "float uv" [0 0 1 0 1 1 0 1]
AttributeEnd
# textured, bump-mapped spheres (uv from the sphere's parametrisation; one partial, one mirrored)
AttributeBegin
  Material "plastic" "texture Kd" "ewa_png" "rgb Ks" [.2 .2 .2] "texture bumpmap" "sphere_bump"
  Translate -3.2 .7 -1.5
  Rotate 40 0 1 0
  Shape "sphere" "float radius" [.7] "float zmin" [-.5] "float phimax" [300]
AttributeEnd
AttributeBegin
  Material "matte" "texture Kd" "checks_pt"
  Translate 3.3 .6 -1.8
  Scale -1 1 1
  Shape "sphere" "float radius" [.6]
AttributeEnd
# specular glass pane with a textured transmittance
AttributeBegin
  Material "glass" "texture Kt" "tri_tga" "rgb Kr" [.9 .9 .9]
  Translate 0 .6 -2.5
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-.7 -.6 0  .7 -.6 0  .7 .6 0  -.7 .6 0] "float uv" [0 0 1 0 1 1 0 1]
AttributeEnd
WorldEnd
"""


def textured_zoo(res=64, spp=16, depth=5, lens=False):
    """Quads with image-textured matte / plastic / uber / substrate / mirror / translucent / glass materials: PNG, TGA
    and PFM files (write_texture_files), EWA / trilinear / unfiltered lookups, the three wrap modes, uv scale / offset,
    black texels (lobes leave the BSDF), meshes with and without uv; lens=True adds a thin lens (differentials with
    lens samples)."""
    return TEXTURED_ZOO % dict(res=res, spp=spp, depth=depth,
                               lens='"float lensradius" [.05] "float focaldistance" [7]' if lens else "")


ALPHA_SCENE = """
LookAt 0 2.5 -7  0 0.8 0  0 1 0
Camera "perspective" "float fov" [40]
Film "image" "integer xresolution" [%(res)d] "integer yresolution" [%(res)d]
Sampler "halton" "integer pixelsamples" [%(spp)d]
Integrator "path" "integer maxdepth" [%(depth)d]
WorldBegin
AttributeBegin
  AreaLightSource "diffuse" "rgb L" [20 19 17]
  Translate 0 5.5 -1
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-1 0 -1  1 0 -1  1 0 1  -1 0 1]
AttributeEnd
LightSource "point" "rgb I" [12 12 14] "point from" [-3 4 -5]
Texture "leaf" "float" "imagemap" "string filename" "alpha.png" "bool gamma" ["false"] "float uscale" [3] "float vscale" [2]
Texture "holes" "float" "imagemap" "string filename" "alpha.png" "bool gamma" ["false"] "bool noFiltering" ["true"]
Texture "zero" "float" "constant" "float value" [0]
Material "matte" "rgb Kd" [.6 .6 .6]
Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-6 0 -6  6 0 -6  6 0 6  -6 0 6]
Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-5 0 4  5 0 4  5 5 4  -5 5 4]
# a cut-out panel: visible and shadow-casting only where the mask is not 0
AttributeBegin
  Material "plastic" "rgb Kd" [.7 .2 .1] "rgb Ks" [.2 .2 .2]
  Translate -1.6 1.4 0
  Rotate 20 0 1 0
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-1.2 -1 0  1.2 -1 0  1.2 1 0  -1.2 1 0] "float uv" [0 0 1 0 1 1 0 1]
        "texture alpha" "leaf"
AttributeEnd
# a panel that is fully visible but lets shadow rays through its holes ("shadowalpha")
AttributeBegin
  Material "matte" "rgb Kd" [.1 .3 .7]
  Translate 1.7 1.4 .5
  Rotate -25 0 1 0
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-1.2 -1 0  1.2 -1 0  1.2 1 0  -1.2 1 0] "float uv" [0 0 1 0 1 1 0 1]
        "texture shadowalpha" "holes"
AttributeEnd
# an invisible mesh (alpha = constant 0 by texture, and by value)
AttributeBegin
  Material "matte" "rgb Kd" [.9 .9 .1]
  Translate 0 1 -2
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-.8 -.8 0  .8 -.8 0  .8 .8 0  -.8 .8 0] "texture alpha" "zero"
  Translate 0 0 -1
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-.8 -.8 0  .8 -.8 0  .8 .8 0  -.8 .8 0] "float alpha" [0]
AttributeEnd
WorldEnd
"""


def write_alpha_png(base_dir, w=32, h=32):
    """A mask with exact zeros (holes), ones and intermediate values."""
    import os
    import numpy as np
    y, x = np.mgrid[0:h, 0:w]
    m = np.where(((x // 4 + y // 4) % 2) == 0, 255, 0).astype(np.uint8)
    m[(x - w // 2) ** 2 + (y - h // 2) ** 2 < (w // 5) ** 2] = 128
    write_png(os.path.join(base_dir, "alpha.png"), np.stack([m, m, m], -1))
    return m


def alpha_scene(res=64, spp=16, depth=4):
    """Triangle meshes with "alpha" / "shadowalpha" float image textures and constant-zero alpha (triangle.cpp:331-338,
    531-570, 716-740); the mask file is alpha.png in base_dir (write_alpha_png)."""
    return ALPHA_SCENE % dict(res=res, spp=spp, depth=depth)


def _curved_patch(nx=6, ny=4):
    """A gently curved grid mesh with per-vertex normals and uv (so that dndu / dndv are not zero)."""
    import math
    P, N, UV, idx = [], [], [], []
    for j in range(ny + 1):
        for i in range(nx + 1):
            u, v = i / nx, j / ny
            ang = (u - 0.5) * 1.2
            P += [2.0 * math.sin(ang), (v - 0.5) * 1.6, -2.0 * math.cos(ang) + 2.0]
            N += [-math.sin(ang), 0.0, -math.cos(ang)]
            UV += [u, v]
    for j in range(ny):
        for i in range(nx):
            a = j * (nx + 1) + i
            idx += [a, a + 1, a + nx + 2, a, a + nx + 2, a + nx + 1]
    f = lambda xs: " ".join("%.9g" % x for x in xs)
    return ('Shape "trianglemesh" "integer indices" [%s] "point P" [%s] "normal N" [%s] "float uv" [%s]'
            % (" ".join(map(str, idx)), f(P), f(N), f(UV)))


BUMP_SCENE = """
LookAt 0 1.6 -6  0 0.9 0  0 1 0
Camera "perspective" "float fov" [40]
Film "image" "integer xresolution" [%(res)d] "integer yresolution" [%(res)d]
Sampler "halton" "integer pixelsamples" [%(spp)d]
Integrator "path" "integer maxdepth" [%(depth)d]
WorldBegin
AttributeBegin
  AreaLightSource "diffuse" "rgb L" [14 14 13]
  Translate 2 4.5 -2
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-1 0 -1  1 0 -1  1 0 1  -1 0 1]
AttributeEnd
LightSource "point" "rgb I" [14 13 12] "point from" [-3 2.5 -4]
Texture "bumps" "float" "imagemap" "string filename" "tex_a.png" "float uscale" [3] "float vscale" [3] "float scale" [.04]
Texture "bumps_raw" "float" "imagemap" "string filename" "tex_b.tga" "bool trilinear" ["true"]
Texture "bumps_tri" "float" "scale" "texture tex1" "bumps_raw" "float tex2" [.08]
Texture "colour" "spectrum" "imagemap" "string filename" "tex_c.pfm"
# ground: flat quad without normals, bump-mapped matte
AttributeBegin
  Material "matte" "rgb Kd" [.5 .5 .5] "texture bumpmap" "bumps"
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-5 0 -5  5 0 -5  5 0 5  -5 0 5] "float uv" [0 0 1 0 1 1 0 1]
AttributeEnd
# curved patch with normals: bump-mapped plastic (dndu, dndv enter)
AttributeBegin
  Material "plastic" "rgb Kd" [.2 .4 .7] "rgb Ks" [.4 .4 .4] "float roughness" [.1] "texture bumpmap" "bumps_tri"
  Translate -1.3 1 0
  %(patch)s
AttributeEnd
# curved patch, mirrored (flipped handedness), bump + textured Kd on uber
AttributeBegin
  Material "uber" "texture Kd" "colour" "rgb Ks" [.3 .3 .3] "rgb Kr" [.15 .15 .15] "texture bumpmap" "bumps"
  Translate 1.5 1 .3
  Scale -1 1 1
  %(patch)s
AttributeEnd
WorldEnd
"""


def bump_scene(res=64, spp=16, depth=4):
    """Material::Bump (material.cpp:47-84) with float image textures: a flat quad without normals, a curved patch with
    vertex normals, a mirrored patch with a textured Kd as well. Uses the files of write_texture_files()."""
    return BUMP_SCENE % dict(res=res, spp=spp, depth=depth, patch=_curved_patch())


SPHERE_ROW_SCENE = """
LookAt 0 1 -9  0 1 0  0 1 0
Camera "perspective" "float fov" [30]
Film "image" "integer xresolution" [%(res)d] "integer yresolution" [%(res)d]
Sampler "halton" "integer pixelsamples" [%(spp)d]
Integrator "path" "integer maxdepth" [%(depth)d]
WorldBegin
AttributeBegin
  AreaLightSource "diffuse" "rgb L" [30 28 25]
  Translate 0 6 2
  Shape "sphere" "float radius" [1]
AttributeEnd
LightSource "point" "rgb I" [20 20 25] "point from" [3 3 -8]
Material "matte" "rgb Kd" [.5 .5 .5]
Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-8 0 -8  8 0 -8  8 0 20  -8 0 20]
Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-8 0 20  8 0 20  8 9 20  -8 9 20]
# nine glass and matte spheres in a row along the view axis: a camera ray meets up to nine quadrics, the shadow rays of the
# wall behind them as many (more than the four a ray's list of postponed quadrics holds)
%(row)s
WorldEnd
"""


def sphere_row_scene(res=48, spp=16, depth=12):
    """Rays that meet more quadrics than the traversal kernel's per-ray list holds (MAX_PEND = 4): a row of spheres
    along the view axis, glass ones in front (the path goes on through them), partial ones (phimax) among them."""
    row = []
    for i in range(9):
        mat = 'Material "glass" "float index" [1.3]' if i < 6 else 'Material "matte" "rgb Kd" [.2 .6 .3]'
        extra = ' "float phimax" [300]' if i in (2, 5) else ""
        row.append('AttributeBegin\n  %s\n  Translate %.2f 1 %.1f\n  Shape "sphere" "float radius" [.7]%s\nAttributeEnd' % (mat, 0.05 * i, -4 + 2.0 * i, extra))
    return SPHERE_ROW_SCENE % dict(res=res, spp=spp, depth=depth, row="\n".join(row))


ROUGHNESS_SCENE = """
LookAt 0 2.4 -7  0 0.8 0  0 1 0
Camera "perspective" "float fov" [40] %(lens)s
Film "image" "integer xresolution" [%(res)d] "integer yresolution" [%(res)d]
Sampler "halton" "integer pixelsamples" [%(spp)d]
Integrator "path" "integer maxdepth" [%(depth)d]
WorldBegin
AttributeBegin
  AreaLightSource "diffuse" "rgb L" [18 17 15]
  Translate 0 5 -1
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-1.5 0 -1.5  1.5 0 -1.5  1.5 0 1.5  -1.5 0 1.5]
AttributeEnd
LightSource "point" "rgb I" [10 10 12] "point from" [-3 3 -5]
# roughness maps: values in (0, 1), as files (gamma off: they are data), scaled into a useful range
Texture "r_raw" "float" "imagemap" "string filename" "tex_a.png" "bool gamma" ["false"] "float uscale" [2] "float vscale" [2]
Texture "r_soft" "float" "scale" "texture tex1" "r_raw" "float tex2" [.5]
Texture "r_tga" "float" "imagemap" "string filename" "rough.pfm" "bool trilinear" ["true"] "float scale" [.6]
Texture "r_pfm" "float" "imagemap" "string filename" "tex_c.pfm" "float scale" [.3] "float uscale" [3] "float vscale" [3]
Texture "colour" "spectrum" "imagemap" "string filename" "tex_c.pfm"
# plastic: "texture roughness", remapped (the default)
AttributeBegin
  Material "plastic" "rgb Kd" [.3 .1 .1] "rgb Ks" [.6 .6 .6] "texture roughness" "r_soft"
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-6 0 -6  6 0 -6  6 0 6  -6 0 6] "float uv" [0 0 1 0 1 1 0 1]
AttributeEnd
# uber: u from a texture, v constant, not remapped; textured Kd as well
AttributeBegin
  Material "uber" "texture Kd" "colour" "rgb Ks" [.5 .5 .5] "rgb Kr" [.05 .05 .05] "texture uroughness" "r_tga" "float vroughness" [.05] "bool remaproughness" ["false"]
  Translate -2 1 0
  %(patch)s
AttributeEnd
# substrate: both axes textured (different maps)
AttributeBegin
  Material "substrate" "rgb Kd" [.1 .3 .5] "rgb Ks" [.3 .3 .3] "texture uroughness" "r_tga" "texture vroughness" "r_pfm"
  Translate 0 1 .5
  %(patch)s
AttributeEnd
# metal: "texture roughness" on a sphere
AttributeBegin
  Material "metal" "texture roughness" "r_pfm"
  Translate 2.1 .8 0
  Shape "sphere" "float radius" [.8]
AttributeEnd
# a "mix" of image-textured materials (mixmat.cpp:46-64): each lobe keeps its texture and its presence rule under ScaledBxDF
MakeNamedMaterial "mixTexA" "string type" "matte" "texture Kd" "colour"
MakeNamedMaterial "mixTexB" "string type" "plastic" "rgb Kd" [.1 .1 .1] "texture Ks" "colour" "float roughness" [.1]
AttributeBegin
  Material "mix" "string namedmaterial1" "mixTexA" "string namedmaterial2" "mixTexB" "rgb amount" [.6 .4 .3]
  Translate 1 .5 -2.2
  Shape "sphere" "float radius" [.5]
AttributeEnd
# translucent: both microfacet lobes take the map
AttributeBegin
  Material "translucent" "rgb Kd" [.3 .3 .3] "rgb Ks" [.5 .5 .5] "texture roughness" "r_soft"
  Translate -.5 1.2 -2
  Rotate 30 0 1 0
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-.8 -.8 0  .8 -.8 0  .8 .8 0  -.8 .8 0] "float uv" [0 0 1 0 1 1 0 1]
AttributeEnd
WorldEnd
"""


def roughness_scene(res=64, spp=16, depth=4, lens=False):
    """Float image textures on "roughness" / "uroughness" / "vroughness" of plastic, uber, substrate, metal and translucent
    (plastic.cpp:57-62, uber.cpp:88-96, substrate.cpp:55-60, metal.cpp:66-73, translucent.cpp:70-72): remapped and not,
    one axis or both, "scale"d, trilinear and EWA. Needs write_texture_files()."""
    return ROUGHNESS_SCENE % dict(res=res, spp=spp, depth=depth, patch=_curved_patch(),
                                  lens='"float lensradius" [.05] "float focaldistance" [7]' if lens else "")


DISNEY_TEXTURED_SCENE = """
LookAt 0 2.2 -7  0 0.6 0  0 1 0
Camera "perspective" "float fov" [42] %(lens)s
Film "image" "integer xresolution" [%(res)d] "integer yresolution" [%(res)d]
Sampler "halton" "integer pixelsamples" [%(spp)d]
Integrator "path" "integer maxdepth" [%(depth)d]
WorldBegin
AttributeBegin
  AreaLightSource "diffuse" "rgb L" [16 15 13]
  Translate 0 5 0
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-1.5 0 -1.5  1.5 0 -1.5  1.5 0 1.5  -1.5 0 1.5]
AttributeEnd
LightSource "point" "rgb I" [10 10 12] "point from" [-3 3 -4]
Texture "ewa_png" "spectrum" "imagemap" "string filename" "tex_a.png" "float uscale" [4] "float vscale" [4]
Texture "tri_tga" "spectrum" "imagemap" "string filename" "tex_b.tga" "bool trilinear" ["true"] "float udelta" [.25]
Texture "pfm_clamp" "spectrum" "imagemap" "string filename" "tex_c.pfm" "string wrap" "clamp" "float uscale" [2] "float vscale" [2]
Texture "png_black" "spectrum" "imagemap" "string filename" "tex_a.png" "string wrap" "black" "float uscale" [1.5] "float udelta" [-.2]
Texture "dz_rough_raw" "float" "imagemap" "string filename" "tex_b.tga" "float uscale" [2] "float vscale" [2]
Texture "dz_rough" "float" "scale" "texture tex1" "dz_rough_raw" "float tex2" [.8]
# ground: every diffuse-side lobe (diffuse, retro, sheen) and the specular lobe take the colour from the map
AttributeBegin
  Material "disney" "texture color" "ewa_png" "float roughness" [.4] "float sheen" [.7] "float sheentint" [.3] "float speculartint" [.6]
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-6 0 -6  6 0 -6  6 0 6  -6 0 6] "float uv" [0 0 1 0 1 1 0 1]
AttributeEnd
# back wall: metallic with clearcoat and anisotropy
AttributeBegin
  Material "disney" "texture color" "tri_tga" "float metallic" [.6] "texture roughness" "dz_rough" "float clearcoat" [.5] "float anisotropic" [.4] "float speculartint" [.2]
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-4 0 4  4 0 4  4 4 4  -4 4 4] "float uv" [0 0 2 0 2 1 0 1]
AttributeEnd
# a thin sheet: fake subsurface, specular and diffuse transmission (strans * Sqrt(c), dt * c)
AttributeBegin
  Material "disney" "texture color" "pfm_clamp" "bool thin" ["true"] "float spectrans" [.5] "float flatness" [.4] "float difftrans" [.8] "texture roughness" "dz_rough_raw" "float eta" [1.4]
  Translate -2.2 1 0
  Rotate 35 0 1 0
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-1 -.9 0  1 -.9 0  1 .9 0  -1 .9 0] "float uv" [0 0 1 0 1 1 0 1]
AttributeEnd
# a solid: specular transmission through the Disney microfacet distribution; black texels outside the map (wrap "black")
AttributeBegin
  Material "disney" "texture color" "png_black" "float spectrans" [.7] "float roughness" [.2] "float sheen" [1] "float metallic" [.2]
  Translate 2.2 1 0
  Rotate -30 0 1 0
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-1 -.9 0  1 -.9 0  1 .9 0  -1 .9 0] "float uv" [0 0 1 0 1 1 0 1]
AttributeEnd
# a curved, bumpy-normal patch
AttributeBegin
  Material "disney" "rgb color" [.7 .4 .2] "texture roughness" "dz_rough" "float sheen" [.4] "float clearcoat" [1] "float clearcoatgloss" [.3] "float spectrans" [.3]
  Translate 0 .2 -1.5
%(patch)s
AttributeEnd
WorldEnd
"""


METAL_TEXTURED_SCENE = DISNEY_TEXTURED_SCENE.split("# ground:")[0] + """# ground: both spectra of the conductor from maps
AttributeBegin
  Material "metal" "texture eta" "ewa_png" "texture k" "tri_tga" "float roughness" [.08]
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-6 0 -6  6 0 -6  6 0 6  -6 0 6] "float uv" [0 0 1 0 1 1 0 1]
AttributeEnd
# back wall: eta from a map, k the copper default; anisotropic roughness
AttributeBegin
  Material "metal" "texture eta" "pfm_clamp" "float uroughness" [.05] "float vroughness" [.2]
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-4 0 4  4 0 4  4 4 4  -4 4 4] "float uv" [0 0 2 0 2 1 0 1]
AttributeEnd
# a panel: k from a map with black texels, eta a constant spectrum
AttributeBegin
  Material "metal" "rgb eta" [.2 .9 1.1] "texture k" "png_black" "float roughness" [.02] "bool remaproughness" ["false"]
  Translate -2.2 1 0
  Rotate 35 0 1 0
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-1 -.9 0  1 -.9 0  1 .9 0  -1 .9 0] "float uv" [0 0 1 0 1 1 0 1]
AttributeEnd
AttributeBegin
  Material "metal" "texture eta" "tri_tga" "texture k" "ewa_png" "float roughness" [.15]
  Translate 0 .2 -1.5
%(patch)s
AttributeEnd
WorldEnd
"""


def metal_textured_scene(res=64, spp=16, depth=5, lens=False):
    """"metal" with image-textured eta / k (metal.cpp:119-122): both, one of them, with black texels. Needs write_texture_files()."""
    return METAL_TEXTURED_SCENE % dict(res=res, spp=spp, depth=depth, patch=_curved_patch(),
                                       lens='"float lensradius" [.05] "float focaldistance" [7]' if lens else "")


SIGMA_TEXTURED_SCENE = DISNEY_TEXTURED_SCENE.split("# ground:")[0] + """Texture "sig_png" "float" "imagemap" "string filename" "tex_a.png" "float uscale" [3] "float vscale" [3]
Texture "sig_deg" "float" "scale" "texture tex1" "sig_png" "float tex2" [60]
Texture "sig_tga" "float" "imagemap" "string filename" "tex_b.tga" "bool trilinear" ["true"]
Texture "sig_big" "float" "scale" "texture tex1" "sig_tga" "float tex2" [200]
Texture "sig_black" "float" "imagemap" "string filename" "tex_a.png" "string wrap" "black" "float uscale" [1.5] "float udelta" [-.2]
Texture "sig_b40" "float" "scale" "texture tex1" "sig_black" "float tex2" [40]
# ground: sigma 0..60 degrees from a map (black texels: Lambertian there)
AttributeBegin
  Material "matte" "rgb Kd" [.7 .6 .5] "texture sigma" "sig_deg"
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-6 0 -6  6 0 -6  6 0 6  -6 0 6] "float uv" [0 0 1 0 1 1 0 1]
AttributeEnd
# back wall: values beyond 90 (clamped), and a textured Kd as well
AttributeBegin
  Material "matte" "texture Kd" "ewa_png" "texture sigma" "sig_big"
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-4 0 4  4 0 4  4 4 4  -4 4 4] "float uv" [0 0 2 0 2 1 0 1]
AttributeEnd
# a panel whose map is black outside [0,1]^2 (wrap "black": sigma = 0, the Lambertian lobe)
AttributeBegin
  Material "matte" "rgb Kd" [.3 .5 .8] "texture sigma" "sig_b40"
  Translate -2.2 1 0
  Rotate 35 0 1 0
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-1 -.9 0  1 -.9 0  1 .9 0  -1 .9 0] "float uv" [0 0 1 0 1 1 0 1]
AttributeEnd
AttributeBegin
  Material "matte" "rgb Kd" [.8 .8 .8] "texture sigma" "sig_deg"
  Translate 0 .2 -1.5
%(patch)s
AttributeEnd
WorldEnd
"""


def sigma_textured_scene(res=64, spp=16, depth=5, lens=False):
    """"matte" with `sigma` a float image texture (matte.cpp:55-62): Lambertian where the map is 0, Oren-Nayar with the A, B
    of the clamped value elsewhere. Needs write_texture_files()."""
    return SIGMA_TEXTURED_SCENE % dict(res=res, spp=spp, depth=depth, patch=_curved_patch(),
                                       lens='"float lensradius" [.05] "float focaldistance" [7]' if lens else "")


GLASS_ROUGH_SCENE = DISNEY_TEXTURED_SCENE.split("# ground:")[0] + """Texture "r_black" "float" "imagemap" "string filename" "tex_a.png" "string wrap" "black" "float uscale" [1.5] "float udelta" [-.2]
Texture "r_b3" "float" "scale" "texture tex1" "r_black" "float tex2" [.3]
Texture "r_tga" "float" "imagemap" "string filename" "tex_b.tga" "bool trilinear" ["true"]
Texture "r_t2" "float" "scale" "texture tex1" "r_tga" "float tex2" [.2]
Material "matte" "rgb Kd" [.6 .6 .6]
Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-6 0 -6  6 0 -6  6 0 6  -6 0 6]
Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-4 0 4  4 0 4  4 4 4  -4 4 4]
# a pane whose roughness map is black outside [0,1]^2: specular glass there, rough glass inside
AttributeBegin
  Material "glass" "rgb Kr" [.9 .9 .9] "rgb Kt" [.9 .8 .7] "texture uroughness" "r_b3" "texture vroughness" "r_b3"
  Translate -2 1.1 0
  Rotate 25 0 1 0
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-1.2 -1 0  1.2 -1 0  1.2 1 0  -1.2 1 0] "float uv" [0 0 1.6 0 1.6 1 0 1]
AttributeEnd
# one axis from a map, the other constant 0; no reflection lobe (Kr black). (Not remapped, a map value of 0 on one axis alone is
# an alpha of 0 and NaNs in the reference as well: the maps here go through RoughnessToAlpha, which stops at 1e-3)
AttributeBegin
  Material "glass" "rgb Kr" [0 0 0] "rgb Kt" [.8 .9 1] "texture uroughness" "r_t2" "float index" [1.3]
  Translate 2 1.1 0
  Rotate -25 0 1 0
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-1.2 -1 0  1.2 -1 0  1.2 1 0  -1.2 1 0] "float uv" [0 0 1 0 1 1 0 1]
AttributeEnd
# a solid ball of map-roughened glass, and a curved patch with different maps on the two axes
AttributeBegin
  Material "glass" "texture uroughness" "r_t2" "texture vroughness" "r_b3"
  Translate 0 .9 1.2
  Shape "sphere" "float radius" [.8]
AttributeEnd
AttributeBegin
  Material "glass" "rgb Kt" [0 0 0] "texture vroughness" "r_b3" "float uroughness" [.05]
  Translate 0 .2 -1.5
%(patch)s
AttributeEnd
WorldEnd
"""


def glass_rough_scene(res=64, spp=16, depth=6, lens=False):
    """"glass" with `uroughness` / `vroughness` float image textures (glass.cpp:60-92): FresnelSpecular where both are 0 at the hit,
    the microfacet reflection / transmission lobes elsewhere. Needs write_texture_files()."""
    return GLASS_ROUGH_SCENE % dict(res=res, spp=spp, depth=depth, patch=_curved_patch(),
                                    lens='"float lensradius" [.05] "float focaldistance" [7]' if lens else "")


NESTED_MIX_SCENE = DISNEY_TEXTURED_SCENE.split("# ground:")[0] + """MakeNamedMaterial "nmA" "string type" "plastic" "rgb Kd" [.7 .1 .1] "rgb Ks" [.3 .3 .3] "float roughness" [.1]
MakeNamedMaterial "nmB" "string type" "mirror" "rgb Kr" [.9 .9 .9]
MakeNamedMaterial "nmC" "string type" "mix" "string namedmaterial1" "nmA" "string namedmaterial2" "nmB" "rgb amount" [.7 .5 .3]
MakeNamedMaterial "nmD" "string type" "matte" "texture Kd" "ewa_png"
MakeNamedMaterial "nmE" "string type" "mix" "string namedmaterial1" "nmC" "string namedmaterial2" "nmD" "rgb amount" [.4 .6 .5]
MakeNamedMaterial "nmF" "string type" "mix" "string namedmaterial1" "nmD" "string namedmaterial2" "nmC" "rgb amount" [.2 .3 .8]
Texture "nm_bump_raw" "float" "imagemap" "string filename" "tex_b.tga" "float uscale" [3] "float vscale" [2]
Texture "nm_bump" "float" "scale" "texture tex1" "nm_bump_raw" "float tex2" [.05]
MakeNamedMaterial "nmH" "string type" "plastic" "rgb Kd" [.2 .5 .7] "rgb Ks" [.4 .4 .4] "float roughness" [.05] "texture bumpmap" "nm_bump"
MakeNamedMaterial "nmI" "string type" "mix" "string namedmaterial1" "nmH" "string namedmaterial2" "nmB" "rgb amount" [.6 .6 .6]
MakeNamedMaterial "nmJ" "string type" "mix" "string namedmaterial1" "nmB" "string namedmaterial2" "nmH" "rgb amount" [.5 .5 .5]
# m1's bump map shapes the frame of the whole mix (mixmat.cpp:52-56); as m2 it has no effect
AttributeBegin
  NamedMaterial "nmI"
  Translate 2 .9 .3
  Shape "sphere" "float radius" [.8]
AttributeEnd
AttributeBegin
  NamedMaterial "nmJ"
  Translate 2.4 .5 -1.6
  Shape "sphere" "float radius" [.5]
AttributeEnd
AttributeBegin
  NamedMaterial "nmE"
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-6 0 -6  6 0 -6  6 0 6  -6 0 6] "float uv" [0 0 1 0 1 1 0 1]
AttributeEnd
AttributeBegin
  NamedMaterial "nmF"
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-4 0 4  4 0 4  4 4 4  -4 4 4] "float uv" [0 0 2 0 2 1 0 1]
AttributeEnd
AttributeBegin
  NamedMaterial "nmC"
  Translate -2 .9 0
  Shape "sphere" "float radius" [.8]
AttributeEnd
AttributeBegin
  NamedMaterial "nmE"
  Translate 0 .2 -1.5
%(patch)s
AttributeEnd
WorldEnd
"""


def nested_mix_scene(res=64, spp=16, depth=5, lens=False):
    """A "mix" of a "mix" (mixmat.cpp:46-64 applied twice: ScaledBxDF(ScaledBxDF(lobe, inner), outer)), either way round, with an
    image-textured sub-material. Needs write_texture_files()."""
    return NESTED_MIX_SCENE % dict(res=res, spp=spp, depth=depth, patch=_curved_patch(),
                                   lens='"float lensradius" [.05] "float focaldistance" [7]' if lens else "")


def disney_textured_scene(res=64, spp=16, depth=5, lens=False):
    """"disney" with an image-textured "color" (disney.cpp:485-587): thick and thin, metallic, sheen, clearcoat, specular and
    diffuse transmission, a map with black texels. Needs write_texture_files()."""
    return DISNEY_TEXTURED_SCENE % dict(res=res, spp=spp, depth=depth, patch=_curved_patch(),
                                        lens='"float lensradius" [.05] "float focaldistance" [7]' if lens else "")


INSTANCED_SCENE = """
LookAt 0 2.2 -8  0 0.9 0  0 1 0
Camera "perspective" "float fov" [42] %(lens)s
Film "image" "integer xresolution" [%(res)d] "integer yresolution" [%(res)d]
Sampler "halton" "integer pixelsamples" [%(spp)d]
Integrator "path" "integer maxdepth" [%(depth)d]
WorldBegin
AttributeBegin
  AreaLightSource "diffuse" "rgb L" [16 15 14]
  Translate 1 5.5 -2
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-1 0 -1  1 0 -1  1 0 1  -1 0 1]
AttributeEnd
LightSource "point" "rgb I" [12 12 14] "point from" [-3 3 -5]
Texture "bumps" "float" "imagemap" "string filename" "tex_a.png" "float uscale" [3] "float vscale" [3] "float scale" [.04]
Texture "colour" "spectrum" "imagemap" "string filename" "tex_c.pfm"
Texture "leaf" "float" "imagemap" "string filename" "alpha.png" "bool gamma" ["false"] "float uscale" [3] "float vscale" [2]
# the object: declared once under a transform of its own (a sphere, a bump-mapped textured patch with normals, an
# alpha-masked panel, a glass ball), each shape with the material bound at its declaration
AttributeBegin
  Rotate 15 0 0 1
  ObjectBegin "thing"
    Material "plastic" "rgb Kd" [.7 .2 .1] "rgb Ks" [.3 .3 .3] "float roughness" [.05]
    Shape "sphere" "float radius" [.45]
    AttributeBegin
      Material "uber" "texture Kd" "colour" "rgb Ks" [.3 .3 .3] "rgb Kr" [.1 .1 .1] "texture bumpmap" "bumps"
      Translate 0 .9 0
      Scale .5 .5 .5
      %(patch)s
    AttributeEnd
    AttributeBegin
      Material "matte" "rgb Kd" [.1 .5 .2]
      Translate .9 .3 0
      Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-.5 -.5 0  .5 -.5 0  .5 .5 0  -.5 .5 0] "float uv" [0 0 1 0 1 1 0 1]
            "texture alpha" "leaf"
    AttributeEnd
    AttributeBegin
      Material "glass" "float index" [1.5]
      Translate -.8 .2 -.3
      Shape "sphere" "float radius" [.3]
    AttributeEnd
  ObjectEnd
AttributeEnd
# a single-shape object (no tree above the shape in the reference; a one-leaf tree here)
ObjectBegin "ball"
  Material "mirror"
  Shape "sphere" "float radius" [.35]
ObjectEnd
Material "matte" "texture Kd" "colour"
Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-7 0 -7  7 0 -7  7 0 7  -7 0 7] "float uv" [0 0 1 0 1 1 0 1]
Material "matte" "rgb Kd" [.6 .6 .6]
Shape "sphere" "float radius" [.5] "float phimax" [270]
AttributeBegin
  Translate -2.2 .6 .5
  Rotate 40 0 1 0
  ObjectInstance "thing"
AttributeEnd
AttributeBegin
  Translate 2 .8 1
  Rotate -70 0 1 0
  Scale 1.4 .8 1.2
  ObjectInstance "thing"
AttributeEnd
AttributeBegin
  Translate 0 1.2 2.5
  Scale -1 1 1
  Rotate 25 1 0 0
  ObjectInstance "thing"
AttributeEnd
AttributeBegin
  Translate .3 .4 -1.5
  ObjectInstance "ball"
AttributeEnd
AttributeBegin
  Translate -1 2.2 1
  Scale 1.5 .7 1
  ObjectInstance "ball"
AttributeEnd
WorldEnd
"""


def instanced_scene(res=64, spp=16, depth=5, lens=False):
    """ObjectBegin / ObjectInstance (api.cpp:1544-1615) as TransformedPrimitives over objects with trees of their own:
    rotated, non-uniformly scaled and mirrored uses of an object with a quadric, a bump-mapped textured mesh with normals, an
    alpha-masked panel and a glass ball; a single-shape object; world shapes between them. Needs write_texture_files() and
    write_alpha_png()."""
    return INSTANCED_SCENE % dict(res=res, spp=spp, depth=depth, patch=_curved_patch(),
                                  lens='"float lensradius" [.05] "float focaldistance" [8]' if lens else "")


def random_scene(seed, res=32, spp=8):
    """A seeded random scene over the supported feature set (fuzzing the HIP path against the oracle): random meshes
    with / without normals and uv, spheres, every material family with random parameters (image-textured, bump-mapped
    and alpha-masked ones included), random lights, optional lens. Needs write_texture_files() + write_alpha_png()."""
    import numpy as np
    rng = np.random.default_rng(seed)
    r = lambda lo, hi: float(rng.uniform(lo, hi))
    rgb = lambda lo=0.05, hi=0.95: "[%.3f %.3f %.3f]" % (r(lo, hi), r(lo, hi), r(lo, hi))
    out = []
    out.append('LookAt %.3f %.3f %.3f  %.3f %.3f 0  0 1 0' % (r(-2, 2), r(1, 4), r(-9, -6), r(-.5, .5), r(.5, 1.5)))
    lens = ('"float lensradius" [%.3f] "float focaldistance" [%.2f]' % (r(.01, .1), r(5, 9))) if rng.random() < .4 else ""
    out.append('Camera "perspective" "float fov" [%.1f] %s' % (r(30, 55), lens))
    filt = rng.choice(["box", "triangle", "gaussian", "mitchell"])
    out.append('PixelFilter "%s"' % filt)
    out.append('Film "image" "integer xresolution" [%d] "integer yresolution" [%d]' % (res, res))
    sampler = rng.choice(["halton", "halton", "sobol", "random"])
    rng2 = np.random.default_rng(seed + 7_000_003)   # (a generator of its own: the scenes of earlier rounds keep their other draws)
    if rng2.random() < .3:
        sampler = rng2.choice(["02sequence", "stratified"])
    if sampler == "stratified":
        xs = max(1, int(round(spp ** .5)))
        while spp % xs: xs -= 1
        out.append('Sampler "stratified" "integer xsamples" [%d] "integer ysamples" [%d] "bool jitter" ["%s"] "integer dimensions" [%d]'
                   % (spp // xs, xs, "true" if rng2.random() < .8 else "false", int(rng2.integers(1, 6))))
    elif sampler == "02sequence":
        out.append('Sampler "02sequence" "integer pixelsamples" [%d] "integer dimensions" [%d]' % (spp, int(rng2.integers(1, 6))))
    else:
        out.append('Sampler "%s" "integer pixelsamples" [%d]' % (sampler, spp))
    integ = "spectralpath" if rng.random() < .25 else "path"
    extra = ' "integer numCABands" [%d]' % int(rng.integers(2, 5)) if integ == "spectralpath" else ""
    out.append('Integrator "%s" "integer maxdepth" [%d] "string lightsamplestrategy" "%s"%s'
               % (integ, int(rng.integers(1, 7)), rng.choice(["uniform", "power", "spatial"]), extra))
    out.append("WorldBegin")
    # lights
    out.append('AttributeBegin\n  AreaLightSource "diffuse" "rgb L" %s %s\n  Translate %.2f %.2f %.2f\n'
               '  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-1 0 -1  1 0 -1  1 0 1  -1 0 1]\nAttributeEnd'
               % (rgb(5, 20), '"bool twosided" ["true"]' if rng.random() < .3 else "", r(-2, 2), r(4, 6), r(-2, 2)))
    if rng.random() < .35:   # a quadric area light, possibly stretched and partial (every direction has a pdf: sphere.cpp:294-310)
        out.append('AttributeBegin\n  AreaLightSource "diffuse" "rgb L" %s\n  Translate %.2f %.2f %.2f\n  Scale %.2f %.2f %.2f\n'
                   '  Shape "sphere" "float radius" [%.2f]%s\nAttributeEnd'
                   % (rgb(4, 15), r(-3, 3), r(2.5, 5), r(-3, 2), r(.6, 1.5), r(.6, 1.5), r(.6, 1.5), r(.2, .6),
                      ' "float zmax" [%.2f]' % r(.05, .15) if rng.random() < .3 else ""))
    if rng.random() < .6:
        out.append('LightSource "point" "rgb I" %s "point from" [%.2f %.2f %.2f]' % (rgb(3, 15), r(-4, 4), r(2, 5), r(-5, 0)))
    if rng.random() < .4:
        out.append('LightSource "spot" "rgb I" %s "point from" [%.2f %.2f %.2f] "point to" [0 0 0] "float coneangle" [%.1f] "float conedeltaangle" [%.1f]'
                   % (rgb(10, 40), r(-4, 4), r(3, 5), r(-5, -2), r(15, 40), r(2, 10)))
    if rng.random() < .4:
        out.append('LightSource "distant" "rgb L" %s "point from" [%.2f 8 %.2f] "point to" [0 0 0]' % (rgb(.2, 1.2), r(-3, 3), r(-6, 0)))
    if rng.random() < .4:
        out.append('LightSource "infinite" "rgb L" %s' % rgb(.1, .6))
    if rng.random() < .3:
        out.append('AttributeBegin\n  AreaLightSource "diffuse" "rgb L" %s\n  Translate %.2f %.2f %.2f\n  Shape "sphere" "float radius" [%.2f]\nAttributeEnd'
                   % (rgb(5, 30), r(-3, 3), r(2.5, 4), r(-1, 2), r(.2, .5)))
    out.append('Texture "img_a" "spectrum" "imagemap" "string filename" "tex_a.png" "float uscale" [%.2f] "float vscale" [%.2f] %s'
               % (r(.5, 4), r(.5, 4), '"bool trilinear" ["true"]' if rng.random() < .5 else ""))
    out.append('Texture "img_b" "spectrum" "imagemap" "string filename" "tex_b.tga" "string wrap" "%s"' % rng.choice(["repeat", "black", "clamp"]))
    out.append('Texture "img_s" "spectrum" "scale" "texture tex1" "img_b" "rgb tex2" %s' % rgb(.3, 1))
    out.append('Texture "bump_raw" "float" "imagemap" "string filename" "tex_a.png" "float uscale" [2] "float vscale" [2]')
    out.append('Texture "bump" "float" "scale" "texture tex1" "bump_raw" "float tex2" [%.3f]' % r(.01, .1))
    out.append('Texture "mask" "float" "imagemap" "string filename" "alpha.png" "bool gamma" ["false"] "float uscale" [%.1f]' % r(1, 3))
    out.append('Texture "sig_raw" "float" "imagemap" "string filename" "tex_b.tga" "string wrap" "black"')
    out.append('Texture "sig_map" "float" "scale" "texture tex1" "sig_raw" "float tex2" [60]')
    out.append('Texture "gl_rough" "float" "scale" "texture tex1" "sig_raw" "float tex2" [.3]')
    out.append('Texture "chk" "spectrum" "checkerboard" "float uscale" [%.1f] "float vscale" [%.1f] "rgb tex1" %s "rgb tex2" %s %s'
               % (r(1, 9), r(1, 9), rgb(), rgb(0, .3), '"string aamode" "none"' if rng.random() < .3 else ""))

    def material():
        k = int(rng.integers(0, 15))
        bump = ' "texture bumpmap" "bump"' if rng.random() < .25 else ""
        if k == 0:
            base = 'Material "matte" "rgb Kd" %s "float sigma" [%.1f]%s' % (rgb(), r(0, 40) if rng.random() < .5 else 0, bump)
            if rng2.random() < .3:   # (round 3: sigma from a float map, 0 .. 60 degrees)
                base = 'Material "matte" "rgb Kd" %s "texture sigma" "sig_map"%s' % (rgb(), bump)
            return base
        if k == 1: return 'Material "plastic" "rgb Kd" %s "rgb Ks" %s "float roughness" [%.3f]%s' % (rgb(), rgb(.05, .5), r(.01, .4), bump)
        if k == 2:
            base = 'Material "glass" "rgb Kr" %s "rgb Kt" %s "float index" [%.2f]' % (rgb(.5, 1), rgb(.5, 1), r(1.2, 1.8))
            if rng2.random() < .3:   # (round 3: roughness from a float map that is 0 outside [0,1]^2: specular there, rough inside)
                pick = int(rng2.integers(0, 3))
                if pick != 1: base += ' "texture uroughness" "gl_rough"'
                if pick != 0: base += ' "texture vroughness" "gl_rough"'
            return base
        if k == 3: return 'Material "mirror" "rgb Kr" %s' % rgb(.5, .95)
        if k == 4: return ('Material "uber" "rgb Kd" %s "rgb Ks" %s "rgb Kr" %s "rgb Kt" %s "float roughness" [%.3f] "rgb opacity" %s%s'
                           % (rgb(), rgb(.05, .4), rgb(0, .3), rgb(0, .3), r(.02, .4), rgb(.6, 1) if rng.random() < .4 else "[1 1 1]", bump))
        if k == 5:
            base = 'Material "metal" "float roughness" [%.3f]' % r(.005, .2)
            if rng2.random() < .4:   # (round 3: eta and / or k from image maps)
                pick = int(rng2.integers(0, 3))
                if pick != 1: base += ' "texture eta" "%s"' % rng2.choice(["img_a", "img_b"])
                if pick != 0: base += ' "texture k" "%s"' % rng2.choice(["img_a", "img_b"])
            return base
        if k == 6: return 'Material "substrate" "rgb Kd" %s "rgb Ks" %s "float uroughness" [%.3f] "float vroughness" [%.3f]' % (rgb(), rgb(.05, .5), r(.02, .3), r(.02, .3))
        if k == 7: return 'Material "translucent" "rgb Kd" %s "rgb Ks" %s "rgb reflect" %s "rgb transmit" %s' % (rgb(), rgb(.05, .4), rgb(.2, .7), rgb(.2, .7))
        if k == 8:
            base = 'Material "disney" "rgb color" %s "float metallic" [%.2f] "float roughness" [%.2f] "float clearcoat" [%.2f] "float sheen" [%.2f]' % (rgb(), r(0, 1), r(.1, .8), r(0, 1), r(0, 1))
            if rng2.random() < .5:   # (round 3: an image-textured colour, thick or thin, with specular transmission and the tints)
                base = base.replace('"rgb color" ' + base.split('"rgb color" ')[1].split(']')[0] + ']', '"texture color" "%s"' % rng2.choice(["img_a", "img_b"]))
                base += ' "float speculartint" [%.2f] "float sheentint" [%.2f]' % (float(rng2.uniform(0, 1)), float(rng2.uniform(0, 1)))
                if rng2.random() < .5: base += ' "float spectrans" [%.2f]' % float(rng2.uniform(.1, .9))
                if rng2.random() < .4: base += ' "bool thin" ["true"] "float flatness" [%.2f] "float difftrans" [%.2f]' % (float(rng2.uniform(0, 1)), float(rng2.uniform(0, 1.6)))
            if rng2.random() < .3:   # (round 3: the roughness from a float map)
                import re as _re
                base = _re.sub(r'"float roughness" \[[0-9.]+\]', '"texture roughness" "gl_rough"', base) + ' "float anisotropic" [%.2f]' % float(rng2.uniform(0, .9))
            return base
        if k == 9: return 'Material "matte" "texture Kd" "img_a"%s' % bump
        if k == 10: return 'Material "plastic" "texture Kd" "img_s" "rgb Ks" %s "float roughness" [%.3f]%s' % (rgb(.05, .4), r(.02, .3), bump)
        if k == 11: return 'Material "uber" "texture Kd" "img_b" "texture Ks" "img_a" "rgb Kr" %s "float roughness" [%.3f]' % (rgb(0, .2), r(.05, .3))
        if k == 12: return 'Material "substrate" "texture Kd" "img_a" "rgb Ks" %s' % rgb(.05, .4)
        if k == 13: return 'Material "glass" "texture Kt" "img_b" "rgb Kr" %s' % rgb(.5, 1)
        return 'Material "matte" "texture Kd" "chk"%s' % bump

    out.append('AttributeBegin\n  %s\n  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-8 0 -8  8 0 -8  8 0 8  -8 0 8] "float uv" [0 0 4 0 4 4 0 4]\nAttributeEnd' % material())
    out.append('AttributeBegin\n  %s\n  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-8 0 5  8 0 5  8 8 5  -8 8 5] "float uv" [0 0 2 0 2 1 0 1]\nAttributeEnd' % material())
    for _ in range(int(rng.integers(3, 8))):
        out.append("AttributeBegin")
        out.append("  " + material())
        out.append("  Translate %.2f %.2f %.2f" % (r(-3.5, 3.5), r(.5, 2.5), r(-2, 3)))
        out.append("  Rotate %.1f %.2f %.2f %.2f" % (r(0, 360), r(-1, 1), r(-1, 1) + 1e-3, r(-1, 1)))
        if rng.random() < .2:
            out.append("  Scale -1 1 1")
        if rng.random() < .2:
            out.append("  ReverseOrientation")
        kind = rng.random()
        if kind < .3:
            out.append('  Shape "sphere" "float radius" [%.2f]%s' % (r(.3, .9), ' "float phimax" [%.0f]' % r(120, 340) if rng.random() < .3 else ""))
        elif kind < .6:
            out.append("  " + _curved_patch(int(rng.integers(2, 6)), int(rng.integers(2, 5))))
        else:
            alpha = ' "texture alpha" "mask"' if rng.random() < .35 else (' "texture shadowalpha" "mask"' if rng.random() < .2 else "")
            out.append('  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-1 -.8 0  1 -.8 0  1 .8 0  -1 .8 0] "float uv" [0 0 1 0 1 1 0 1]%s' % alpha)
        out.append("AttributeEnd")
    # object instancing (api.cpp:1544-1615): a named object of two shapes with their own materials, instanced under
    # rotated, scaled and mirrored transforms
    if rng.random() < .6:
        out.append('AttributeBegin\n  %s\nObjectBegin "thing"' % material())
        out.append('  Shape "sphere" "float radius" [%.2f]' % r(.15, .35))
        out.append('  %s\n  Translate 0 %.2f 0\n  %s' % (material(), r(.3, .6), _curved_patch(3, 2)))
        out.append("ObjectEnd\nAttributeEnd")
        for _ in range(int(rng.integers(1, 4))):
            out.append("AttributeBegin\n  Translate %.2f %.2f %.2f\n  Rotate %.1f 0 1 0\n  Scale %.2f %.2f %.2f" %
                       (r(-3, 3), r(.3, 2), r(-2, 2), r(0, 360), r(.5, 1.5) * (-1 if rng.random() < .3 else 1), r(.5, 1.5), r(.5, 1.5)))
            out.append('  ObjectInstance "thing"\nAttributeEnd')
    out.append("WorldEnd")
    return "\n".join(out) + "\n"


def _piz_wav2_encode(a, nx, ox, ny, oy, mx):
    """2D wavelet encoding in place (list of ints), finest level first: the inverse of the reader's Wav2Decode."""
    w14 = mx < (1 << 14)

    def s16(v):
        v &= 0xffff
        return v - 0x10000 if v & 0x8000 else v

    def enc(x, y):
        if w14:
            a_, b_ = s16(x), s16(y)
            return ((a_ + b_) >> 1) & 0xffff, (a_ - b_) & 0xffff
        ao = (x + 0x8000) & 0xffff
        m = (ao + y) >> 1
        d = ao - y
        if d < 0:
            m = (m + 0x8000) & 0xffff
        return m & 0xffff, d & 0xffff
    n = min(nx, ny)
    p, p2 = 1, 2
    while p2 <= n:
        py, ey = 0, oy * (ny - p2)
        oy1, oy2, ox1, ox2 = oy * p, oy * p2, ox * p, ox * p2
        while py <= ey:
            px, ex = py, py + ox * (nx - p2)
            while px <= ex:
                p01, p10, p11 = px + ox1, px + oy1, px + oy1 + ox1
                i00, i01 = enc(a[px], a[p01])
                i10, i11 = enc(a[p10], a[p11])
                a[px], a[p10] = enc(i00, i10)
                a[p01], a[p11] = enc(i01, i11)
                px += ox2
            if nx & p:
                p10 = px + oy1
                i00, a[p10] = enc(a[px], a[p10])
                a[px] = i00
            py += oy2
        if ny & p:
            px, ex = py, py + ox * (nx - p2)
            while px <= ex:
                p01 = px + ox1
                i00, a[p01] = enc(a[px], a[p01])
                a[px] = i00
                px += ox2
        p, p2 = p2, p2 << 1


def _piz_huf_compress(vals):
    """Huffman coding of 16-bit values in the layout the reader's HufDecode takes: canonical codes from code lengths, a run-length
    marker symbol one past the largest value, the table of 6-bit lengths with zero runs."""
    import heapq, struct
    from collections import Counter
    freq = Counter(vals)
    im, iM = min(freq), max(freq) + 1
    freq[iM] = 1   # the run-length marker
    heap = [(f, s, (s,)) for s, f in freq.items()]
    heapq.heapify(heap)
    length = {s: 0 for s in freq}
    if len(heap) == 1:
        length[heap[0][1]] = 1
    while len(heap) > 1:
        f1, s1, g1 = heapq.heappop(heap)
        f2, s2, g2 = heapq.heappop(heap)
        for s_ in g1 + g2:
            length[s_] += 1
        heapq.heappush(heap, (f1 + f2, min(s1, s2), g1 + g2))
    assert max(length.values()) <= 58
    count = [0] * 59
    for l in length.values():
        count[l] += 1
    base, c = [0] * 59, 0
    for i in range(58, 0, -1):
        nc = (c + count[i]) >> 1
        base[i] = c
        c = nc
    code, nxt = {}, list(base)
    for s_ in sorted(length):
        l = length[s_]
        code[s_] = (nxt[l], l)
        nxt[l] += 1
    bits = []   # (value, width) pairs, most significant bit first

    def flush(pairs):
        acc, n, out = 0, 0, bytearray()
        for v, wdt in pairs:
            acc = (acc << wdt) | v
            n += wdt
            while n >= 8:
                out.append((acc >> (n - 8)) & 0xff)
                n -= 8
            acc &= (1 << n) - 1
        if n:
            out.append((acc << (8 - n)) & 0xff)
        return bytes(out), sum(wdt for _, wdt in pairs)
    table, i = [], im
    while i <= iM:
        l = length.get(i, 0)
        if l == 0:
            run = 1
            while i + run <= iM and length.get(i + run, 0) == 0 and run < 261:
                run += 1
            if run >= 6:
                table += [(63, 6), (run - 6, 8)]
            elif run >= 2:
                table.append((59 + run - 2, 6))
            else:
                table.append((0, 6))
            i += run
        else:
            table.append((l, 6))
            i += 1
    tbytes, _ = flush(table)
    i = 0
    while i < len(vals):
        run = 1
        while i + run < len(vals) and vals[i + run] == vals[i] and run < 256:
            run += 1
        bits.append(code[vals[i]])
        if run >= 4:
            bits += [code[iM], (run - 1, 8)]
            i += run
        else:
            i += 1
    dbytes, nbits = flush(bits)
    return struct.pack("<IIIII", im, iM, len(tbytes), nbits, 0) + tbytes + dbytes


def _piz_compress(planes, nx, ny, size):
    """planes: per channel the block's 16-bit words (numpy uint16, [ny][nx * size]); size = words per pixel."""
    import struct
    import numpy as np
    allv = np.concatenate(planes)
    bitmap = np.zeros(8192, np.uint8)
    present = np.unique(allv)
    for v in present:
        bitmap[v >> 3] |= 1 << (v & 7)
    bitmap[0] &= 0xfe   # (zero is always there)
    nz = np.nonzero(bitmap)[0]
    lo, hi = (int(nz[0]), int(nz[-1])) if len(nz) else (8191, 0)
    lut = np.zeros(65536, np.int64)
    k = 0
    for i in range(65536):
        if i == 0 or (bitmap[i >> 3] & (1 << (i & 7))):
            lut[i] = k
            k += 1
    maxv = k - 1
    vals = []
    for pl in planes:
        a = [int(lut[v]) for v in pl]
        for j in range(size):
            sub = a[j::size] if size > 1 else a
            _piz_wav2_encode(sub, nx, 1, ny, nx, maxv)
            if size > 1:
                a[j::size] = sub
        vals += a
    huf = _piz_huf_compress(vals)
    head = struct.pack("<HH", lo, hi) + (bitmap[lo:hi + 1].tobytes() if lo <= hi else b"")
    return head + struct.pack("<i", len(huf)) + huf


def write_exr(path, img, compression="zip", dtype="half", data_window_origin=(0, 0), keep_larger=False):
    """Minimal scan-line OpenEXR writer for tests: img [h, w, 3] float; channels B, G, R (alphabetical, as the format wants)
    as HALF or FLOAT; compression "none", "zips" (1 line per chunk) or "zip" (16 lines per chunk)."""
    import struct, zlib
    import numpy as np
    h, w, _ = img.shape
    px = img.astype(np.float16 if dtype == "half" else np.float32)
    ptype = 1 if dtype == "half" else 2
    comp = {"none": 0, "zips": 2, "zip": 3, "piz": 4}[compression]
    lines_per = 32 if comp == 4 else (16 if comp == 3 else 1)
    x0, y0 = data_window_origin

    def attr(name, typ, data):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(data)) + data
    chlist = b"".join(n.encode() + b"\0" + struct.pack("<iBBBBii", ptype, 0, 0, 0, 0, 1, 1) for n in ("B", "G", "R")) + b"\0"
    box = struct.pack("<iiii", x0, y0, x0 + w - 1, y0 + h - 1)
    header = (attr("channels", "chlist", chlist) + attr("compression", "compression", bytes([comp])) + attr("dataWindow", "box2i", box) +
              attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", b"\0") + attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) +
              attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0)) + attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0")
    chunks = []
    for c0 in range(0, h, lines_per):
        raw = b"".join(px[y, :, k].tobytes() for y in range(c0, min(c0 + lines_per, h)) for k in (2, 1, 0))
        data = raw
        if comp == 4:
            nl = min(c0 + lines_per, h) - c0
            size = 1 if dtype == "half" else 2
            # the block as PIZ wants it: channel by channel, its lines, 16-bit words
            planes = [np.frombuffer(b"".join(px[y, :, k].tobytes() for y in range(c0, c0 + nl)), np.uint16).copy() for k in (2, 1, 0)]
            z = _piz_compress(planes, w, nl, size)
            data = z if (len(z) < len(raw) or (keep_larger and len(z) != len(raw))) else raw   # (keep_larger: tests of the decoder on data that does not shrink)
        elif comp:
            t = np.frombuffer(raw, np.uint8)
            re = np.concatenate([t[0::2], t[1::2]]).astype(np.int32)
            d = re.copy()
            d[1:] = (re[1:] - re[:-1] + 128 + 256) & 255
            z = zlib.compress(d.astype(np.uint8).tobytes(), 6)
            data = z if len(z) < len(raw) else raw
        chunks.append(struct.pack("<ii", y0 + c0, len(data)) + data)
    start = 8 + len(header) + 8 * len(chunks)
    offs, p = [], start
    for c in chunks:
        offs.append(p); p += len(c)
    with open(path, "wb") as f:
        f.write(struct.pack("<II", 20000630, 2) + header + b"".join(struct.pack("<Q", o) for o in offs) + b"".join(chunks))


MIS_SPAN_SCENE = """
LookAt %(ox)g 2.5 %(cz)g  %(ox)g 2 %(oz)g  0 1 0
Camera "perspective" "float fov" [40]
Film "image" "integer xresolution" [%(res)d] "integer yresolution" [%(res)d]
Sampler "halton" "integer pixelsamples" [%(spp)d]
Integrator "path" "integer maxdepth" [%(depth)d] "string lightsamplestrategy" "uniform"
WorldBegin
Translate %(ox)g 0 %(oz)g
# (a) an emitter lying IN the ceiling: both of its triangles are coplanar with the two ceiling triangles, so a ray that
#     reaches it meets the ceiling at the same t and the closest hit is whichever the reference tests first
Material "matte" "rgb Kd" [.6 .6 .6]
Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-6 5 -6  6 5 -6  6 5 8  -6 5 8]
AttributeBegin
  AreaLightSource "diffuse" "rgb L" [6 6 5]
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-2 5 -1  2 5 -1  2 5 3  -2 5 3]
AttributeEnd
# (b) a one-sided emitter on the left wall that faces the wall (its light goes into a 0.001 gap), and a two-sided one on the right
Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-6 0 -6  -6 0 8  -6 5 8  -6 5 -6]
Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [6 0 -6  6 5 -6  6 5 8  6 0 8]
AttributeBegin
  AreaLightSource "diffuse" "rgb L" [9 3 3]
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-5.999 1 0  -5.999 1 3  -5.999 3 3  -5.999 3 0]
AttributeEnd
AttributeBegin
  AreaLightSource "diffuse" "rgb L" [3 3 9] "bool twosided" ["true"]
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [5.5 1 0  5.5 3 0  5.5 3 3  5.5 1 3]
AttributeEnd
# (c) a sphere light sunk half into the back wall (geometry inside the span of its bounds), a partial one (a ray through the
#     cut-away part misses it) with reversed orientation, and one inside a glass shell (a quadric on every ray towards it)
Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-6 0 8  6 0 8  6 5 8  -6 5 8]
AttributeBegin
  AreaLightSource "diffuse" "rgb L" [5 8 5]
  Translate -3 2.5 8
  Shape "sphere" "float radius" [.8]
AttributeEnd
AttributeBegin
  AreaLightSource "diffuse" "rgb L" [8 8 3]
  Translate 3 1.2 4
  Rotate 40 1 0 0
  ReverseOrientation
  Shape "sphere" "float radius" [.7] "float phimax" [250] "float zmax" [.4]
AttributeEnd
AttributeBegin
  AreaLightSource "diffuse" "rgb L" [10 10 10]
  Translate 0 1.5 5
  Shape "sphere" "float radius" [.4]
AttributeEnd
AttributeBegin
  Material "glass" "float index" [1.4]
  Translate 0 1.5 5
  Shape "sphere" "float radius" [.8]
AttributeEnd
# floor (glossy: its BSDF samples aim at the emitters), a matte block that hides part of every emitter from part of the floor
Material "plastic" "rgb Kd" [.4 .4 .45] "rgb Ks" [.4 .4 .4] "float roughness" [.15]
Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-6 0 -6  6 0 -6  6 0 8  -6 0 8]
Material "matte" "rgb Kd" [.7 .5 .3]
Shape "trianglemesh" "integer indices" [0 1 2 0 2 3  4 5 6 4 6 7  0 1 5 0 5 4  2 3 7 2 7 6  1 2 6 1 6 5  0 3 7 0 7 4]
  "point P" [-1.5 0 1  1.5 0 1  1.5 0 2  -1.5 0 2  -1.5 2.2 1  1.5 2.2 1  1.5 2.2 2  -1.5 2.2 2]
WorldEnd
"""


def mis_span_scene(res=64, spp=16, depth=5, offset=0.0):
    """Emitters whose BSDF-sampled (MIS) rays cannot be settled by a visibility query alone: coplanar with other geometry,
    sunk into a wall, behind a quadric, partial, one-sided and facing away (k_trav MODE 3 / k_resolve_overflow). `offset`
    moves the whole scene and the camera that far from the origin along x and z (coarser floats: a coplanar emitter and
    ceiling then differ by more in their computed hit distances)."""
    return MIS_SPAN_SCENE % dict(res=res, spp=spp, depth=depth, ox=offset, oz=offset, cz=offset - 9)
