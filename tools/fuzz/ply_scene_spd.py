import sys, os, tempfile, random
W = os.environ.get('MIPT_ASAN_TREE', '/tmp/mipt_asan'); sys.path.insert(0, W); sys.path.insert(0, os.path.join(W, 'tests'))
import numpy as np
import pbrt_v3_spectral_amd as pt
import test_frontend as tf
d=tempfile.mkdtemp()
verts=[(0,0,0),(1,0,0),(1,1,0),(0,1,0),(0.5,0.5,1)]; faces=[(0,1,2),(0,2,3),(0,1,4,2)]
normals=[(0,0,1)]*5; uvs=[(0,0),(1,0),(1,1),(0,1),(.5,.5)]
rnd=random.Random(2); n=0
for fmt in ("ascii","binary_little_endian","binary_big_endian"):
    tf._write_ply(os.path.join(d,'m.ply'), fmt, verts, faces, normals, uvs)
    data=open(os.path.join(d,'m.ply'),'rb').read()
    for it in range(500):
        b=bytearray(data)
        if rnd.random()<0.3: b=b[:rnd.randrange(0,len(b))]
        else:
            for _ in range(rnd.randrange(1,5)):
                b[rnd.randrange(0,len(b))]=rnd.randrange(256)
        open(os.path.join(d,'f.ply'),'wb').write(bytes(b))
        s=pt.Scene(text='Camera "perspective"\nWorldBegin\nShape "plymesh" "string filename" "f.ply"\nWorldEnd\n', base_dir=d); n+=1
# PLY headers that lie about their elements: duplicated / negative / absurd element counts
props="property float x\nproperty float y\nproperty float z\n"; face="element face 1\nproperty list uchar int vertex_indices\n"
for header in ("element vertex 3\n"+props+face+"element vertex 1\n"+props, "element vertex -5\n"+props+face,
               "element vertex 4000000000000\n"+props+face, "element vertex 3\n"+props+"element face 9000000000000\nproperty list uchar int vertex_indices\n"):
    open(os.path.join(d,'f.ply'),'w').write("ply\nformat ascii 1.0\n"+header+"end_header\n0 0 0\n1 0 0\n0 1 0\n3 0 1 2\n")
    s=pt.Scene(text='Camera "perspective"\nWorldBegin\nShape "plymesh" "string filename" "f.ply"\nWorldEnd\n', base_dir=d); n+=1
    assert s.errors
# .pbrt text mutations
import scenes_text as st
txt=st.material_zoo(res=8, spp=1)
for it in range(400):
    b=bytearray(txt.encode())
    for _ in range(rnd.randrange(1,6)):
        i=rnd.randrange(0,len(b)); b[i]=rnd.choice(b'[]"# 0123456789.-eE\n' + bytes([rnd.randrange(32,127)]))
    try: pt.Scene(text=bytes(b).decode('latin1'))
    except Exception: pass
    n+=1
# spd files
for it in range(200):
    open(os.path.join(d,'s.spd'),'wb').write(bytes(rnd.randrange(256) if rnd.random()<.2 else rnd.choice(b"0123456789. -+e#\n") for _ in range(rnd.randrange(0,200))))
    pt.Scene(text='Camera "perspective"\nWorldBegin\nLightSource "point" "spectrum I" "s.spd"\nWorldEnd\n', base_dir=d); n+=1
print("inputs loaded without a crash:", n)
