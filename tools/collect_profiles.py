#!/usr/bin/env python3
"""Copy what tools/profile_configs.sh left under gpurun_out/<tag>_<workload>/ into profiles/ (the committed evidence) and
rebuild profiles/pmc_traffic.json (HBM bytes per k_trav<0> launch of each workload, which bench.py quotes as roofline.traffic).
Usage: python tools/collect_profiles.py <tag> <round prefix, e.g. r2>"""
import json, os, re, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
NAMES = {"killeroo": "killeroo-simple", "cornell": "cornell-glass", "procedural": "procedural-10000000tris", "matzoo": "matzoo", "texzoo": "textured-zoo"}
traffic_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
try:
    traffic = json.load(open(traffic_path))
    if "kernel" in traffic:   # round-1 layout (one workload)
        traffic = {}
except (OSError, ValueError):
    traffic = {}
for w, wl in NAMES.items():
    src = os.path.join(ROOT, "gpurun_out", "%s_%s" % (tag, w))
    if not os.path.isdir(src):
        continue
    for f, dst in (("bench.json", "bench"), ("trace_bench.json", "trace_bench"), ("kernel_stats.csv", "kernel_stats"), ("pmc_summary.txt", "pmc_hbm_summary")):
        if os.path.exists(os.path.join(src, f)):
            ext = os.path.splitext(f)[1]
            shutil.copy(os.path.join(src, f), os.path.join(ROOT, "profiles", "%s_%s_%s%s" % (rnd, w, dst, ext)))
    txt = open(os.path.join(src, "pmc_summary.txt")).read()
    m = re.search(r"k_trav<0[^\n]*\n((?:   [^\n]*\n)+)", txt)
    vals = dict(re.findall(r"   (\S+)\s+mean/dispatch (\S+)", m.group(1)))
    n = int(re.search(r"FETCH_SIZE[^\n]*\(n=(\d+)\)", m.group(1)).group(1))
    bench = json.load(open(os.path.join(src, "bench.json")))
    fetch_kb, write_kb = float(vals["FETCH_SIZE"]), float(vals["WRITE_SIZE"])
    traffic[wl] = {
        "kernel": "k_trav<0>", "workload_spp": bench["config"]["spp"], "streams": 1, "launches_per_frame": n,
        "FETCH_SIZE_kb_per_launch": fetch_kb, "WRITE_SIZE_kb_per_launch": write_kb,
        "k_trav0_hbm_bytes_per_launch": int((2 * fetch_kb + write_kb) * 1024),
        "l2_hit_rate": round(float(vals["TCC_HIT_sum"]) / (float(vals["TCC_HIT_sum"]) + float(vals["TCC_MISS_sum"])), 3),
        "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE tallies 128-B reads at 64 B, MI355X_MICROARCH.md HBM section)",
        "source": "profiles/%s_%s_pmc_hbm_summary.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, tools/profile_configs.sh)" % (rnd, w)}
    # the shading instances: HBM bytes per launch of each k_shade<NL, TM>, and all of them per shaded vertex
    shade = {}
    for name, body in re.findall(r"(k_shade<[^\n]*)\n((?:   [^\n]*\n)+)", txt):
        v = dict(re.findall(r"   (\S+)\s+mean/dispatch (\S+)", body))
        nn = re.search(r"\(n=(\d+)\)", body)
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            shade[name.strip()] = {"hbm_bytes_per_launch": int((2 * float(v["FETCH_SIZE"]) + float(v["WRITE_SIZE"])) * 1024), "launches": int(nn.group(1))}
    tot = sum(e["hbm_bytes_per_launch"] * e["launches"] for e in shade.values())
    verts = None
    try:
        p1 = json.load(open(os.path.join(src, "pass1.json")))   # the counter pass's own bench line: one frame
        verts = p1["roofline"].get("shaded_vertices")
    except (OSError, ValueError, KeyError):
        pass
    traffic[wl]["k_shade"] = {"instances": shade, "hbm_bytes_per_frame": tot, "shaded_vertices_per_frame": verts,
                              "hbm_bytes_per_vertex": round(tot / verts, 1) if verts else None, "algorithmic_bytes_per_vertex": 960}
    r = bench["roofline"]
    print("%-12s %.1f Mray/s  %.1f ms/frame  k_trav<0>: algorithmic %.2f GB/launch, HBM traffic %.2f GB/launch (%.2fx), frac %.3f" %
          (w, bench["value"], bench["ms_per_step"], r["bytes_per_ray"] * r["rays_per_launch"] / 1e9, traffic[wl]["k_trav0_hbm_bytes_per_launch"] / 1e9,
           traffic[wl]["k_trav0_hbm_bytes_per_launch"] / (r["bytes_per_ray"] * r["rays_per_launch"]), r["frac"]))
json.dump(traffic, open(traffic_path, "w"), indent=1)
