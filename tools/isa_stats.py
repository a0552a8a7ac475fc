"""Instruction counts per device function of pt_kernels.hip (code size matters: the shading kernels are
instruction-fetch bound). Usage: python tools/isa_stats.py [extra hipcc flags]"""
import re, collections, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = "/tmp/pt_isa.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-I" + root + "/include",
                       "--offload-device-only", "-S", root + "/pbrt-v3-spectral_amd/csrc/device/pt_kernels.hip", "-o", out] + sys.argv[1:],
                      stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^[_A-Za-z0-9.$]+:\s+; @", l)]
starts.append((len(lines), "end"))
vg = dict(re.findall(r"\.set (\S+)\.num_vgpr, (\S.*)", "\n".join(lines)))
for (i, n), (j, _) in zip(starts, starts[1:]):
    c = collections.Counter()
    for l in lines[i:j]:
        if l.startswith(".Lfunc_end"):
            break
        m = re.match(r"\t([a-z_0-9]+)\s", l + " ")
        if m and not l.startswith("\t."):
            c[m.group(1)] += 1
    short = re.sub(r"N3dpt6DScene.*", "", n).replace("_ZN12_GLOBAL__N_1", "")
    g = lambda p: sum(v for k, v in c.items() if k.startswith(p))
    print("%-62s insts %6d gload %4d gstore %3d scratch %4d f64 %5d call %3d vgpr %s" % (
        short[:62], sum(c.values()), g(("global_load", "flat_load", "buffer_load")), g(("global_store", "flat_store", "buffer_store")),
        g("scratch_"), sum(v for k, v in c.items() if "f64" in k), c["s_swappc_b64"], vg.get(n, "?")[:12]))
