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
