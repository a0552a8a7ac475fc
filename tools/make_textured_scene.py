"""Write the image-textured test scene (tests/scenes_text.py: textured_zoo) and its texture files to a directory, with
textures large enough to matter for the caches. Usage: python tools/make_textured_scene.py <dir> [texture size] [res]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import scenes_text as st

out = sys.argv[1]
size = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
res = int(sys.argv[3]) if len(sys.argv) > 3 else 700
os.makedirs(out, exist_ok=True)
st.write_png(os.path.join(out, "tex_a.png"), st._texture_image(size, size // 2, 1), with_alpha=True)
st.write_tga(os.path.join(out, "tex_b.tga"), st._texture_image(size * 3 // 4, size * 5 // 8, 2), rle=True)   # resampled to a power of two
c = (st._texture_image(size // 2, size // 2, 3).astype(np.float32) / 255.0) ** 2 * 1.5
with open(os.path.join(out, "tex_c.pfm"), "wb") as f:
    f.write(b"PF\n%d %d\n-1.0\n" % (c.shape[1], c.shape[0]))
    f.write(c[::-1].astype(np.float32).tobytes())
with open(os.path.join(out, "textured-zoo.pbrt"), "w") as f:
    f.write(st.textured_zoo(res=res, spp=256, depth=5))
print(os.path.join(out, "textured-zoo.pbrt"))
