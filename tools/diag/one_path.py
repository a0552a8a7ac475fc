"""Diagnostic (GPU box): print one camera sample's path records, device and oracle. Usage: one_path.py <scene> <spp> <x> <y> <k>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import pbrt_v3_spectral_amd as pt, oracle_binding as ob
from sample_divergence import mk
name, spp, x, y, k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
s = mk(name, -1, spp)
integ = pt.CreatePathIntegrator(s)
d = integ.debug_path(x, y, k)
o = ob.path_log(s, x, y, k)
np.set_printoptions(precision=6, linewidth=200)
for v in range(max(len(d), len(o))):
    for nm, rec in (("dev", d), ("ora", o)):
        if v < len(rec):
            r = rec[v]
            print(nm, "vertex", v, "bounces %g prim %g dims %g->%g ended %g t %g" % (r[0], r[1], r[2], r[15], r[3], r[7]))
            print("    beta", r[20:51])
            print("    L   ", r[51:82])
li, _ = ob.li(s, np.array([[x, y, k]]))
print("oracle Li", li)
