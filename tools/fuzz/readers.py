import sys, os, tempfile, random
W = os.environ.get('MIPT_ASAN_TREE', '/tmp/mipt_asan'); sys.path.insert(0, W); sys.path.insert(0, os.path.join(W, 'tests'))
import numpy as np
import scenes_text as st, pbrt_v3_spectral_amd as pt
d=tempfile.mkdtemp()
img=st._texture_image(16,8,1)
st.write_png(os.path.join(d,'a.png'), img, with_alpha=True)
st.write_tga(os.path.join(d,'a.tga'), img, rle=True)
st.write_tga(os.path.join(d,'b.tga'), img, rle=False)
st.write_exr(os.path.join(d,'a.exr'), img.astype(np.float32)/255, compression='zip', dtype='half')
st.write_exr(os.path.join(d,'b.exr'), img.astype(np.float32)/255, compression='none', dtype='float')
big=np.tile(img.astype(np.float32)/255, (5,1,1))[:37,:15]   # two PIZ blocks, odd sizes
st.write_exr(os.path.join(d,'c.exr'), big, compression='piz', dtype='half', keep_larger=True)
st.write_exr(os.path.join(d,'d.exr'), big[:9], compression='piz', dtype='float', keep_larger=True)
with open(os.path.join(d,'a.pfm'),'wb') as f:
    f.write(b"PF\n16 8\n-1.0\n"); f.write((img.astype(np.float32)/255).tobytes())
rnd=random.Random(1)
scene='Camera "perspective"\nWorldBegin\nTexture "t" "spectrum" "imagemap" "string filename" "%s"\nLightSource "infinite" "string mapname" "%s"\nWorldEnd\n'
n=0
for name in ('a.png','a.tga','b.tga','a.exr','b.exr','c.exr','d.exr','a.pfm'):
    data=open(os.path.join(d,name),'rb').read()
    ext=name.split('.')[-1]
    for it in range(400):
        b=bytearray(data)
        mode=rnd.random()
        if mode<0.3: b=b[:rnd.randrange(0,len(b))]
        else:
            for _ in range(rnd.randrange(1,6)):
                i=rnd.randrange(0,min(len(b), 120 if rnd.random()<0.4 else len(b)))
                b[i]=rnd.randrange(256)
        fn='m.'+ext
        open(os.path.join(d,fn),'wb').write(bytes(b))
        s=pt.Scene(text=scene%(fn,fn), base_dir=d)
        n+=1
print("mutations loaded without a crash:", n)
