#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json: Mray/s at fixed spp).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (config.workload): killeroo-simple.pbrt, 700x700, PathIntegrator maxdepth 5,
Halton, box filter, SampledSpectrum-31, 1024 spp -- BASELINE.json configs[1]. A *step*
is one complete render of that frame (all 1024 samples of every pixel through the
wavefront pipeline, film accumulated on the device); K steps render the frame K times.
The scene (BVH, meshes, tables) and the film are resident in HBM before the timed region
starts; nothing crosses PCIe inside it.

With N > 1 the film's 16x16 tiles are sharded over the ranks (tile_id % N == rank, same
Halton indices as the 1-GPU render), each rank accumulates into its own device film and
ONE RCCL sum-reduce of that film, in place ([H, W, 32] floats: 62.7 MB at 700x700), closes every
step (strong scaling: total work is the fixed 1024-spp frame). Rank 0 parses the scene and builds
the BVH once; the other ranks load its binary cache. `per_rank` in the JSON line gives each rank's
render and reduce seconds per step and the imbalance max/mean of the render times. A ray = one Scene::Intersect or Scene::IntersectP call
(src/core/scene.cpp:40-55), counted on the device.

The JSON line also carries
  roofline:      closest-hit traversal (k_trav<0>), algorithmic bytes
                 B_ray = 32*N_node + 48*N_tri + 64 with N_node/N_tri counted by the kernel,
                 divided by the kernel's average launch duration (HIP events on the stream
                 each launch goes to, inside mi_pt_render), against the 8 TB/s HBM peak.
                 By default one sub-renderer (one stream) renders the frame, so each launch has
                 the GPU to itself while it is timed; --streams K runs K sub-renderers
                 concurrently (a few % more throughput, but then a timed launch shares the CUs).
  cpu_baseline:  the CPU oracle (a port of the reference algorithm, oracle/) timed on the
                 host cores on a bounded sample of the same workload (rank 0, N = 1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 measured)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=1024, help="samples per pixel of the frame (configs[1]: 1024)")
    ap.add_argument("--scene", default=os.path.join(ROOT, "scenes", "killeroo-simple.pbrt"))
    ap.add_argument("--procedural-tris", type=int, default=0,
                    help="render the seeded procedural scene of tools/make_procedural_scene.py with this many "
                         "triangles instead of --scene (BASELINE configs 4/5 stand-in)")
    ap.add_argument("--pool", type=int, default=0, help="resident path slots (0 = library default)")
    ap.add_argument("--cpu-samples", type=int, default=80_000_000,
                    help="camera samples the CPU oracle renders for cpu_baseline (0 = skip)")
    ap.add_argument("--streams", type=int, default=0,
                    help="concurrent sub-renderers (0 = MIPT_STREAMS from the environment, else 1)")
    ap.add_argument("--exclusive-spp", type=int, default=0,
                    help="with --streams > 1: spp of an extra one-stream pass that times the traversal kernel alone (0 = skip)")
    ap.add_argument("--dump-film", default=None, help="rank 0 saves the (reduced) film of the last step as .npy: [H, W, 32] = 31 bins + weight")
    ap.add_argument("--pmc-traffic", type=float, default=None,
                    help="HBM bytes per k_extend launch from a separate rocprofv3 --pmc pass")
    a = ap.parse_args()

    # concurrent sub-renderer streams need their own hardware queues (HIP maps streams onto 4 by default)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    # One sub-renderer per GPU at every N: every timed launch has the GPU to itself (the roofline figure), and with
    # round 2's kernels concurrent pools no longer pay (one-GPU rehearsal of the shards, tools/shard_scaling.py:
    # slowest 1/2 shard 0.433 s with one pool, 0.453 s with four; 1/4 shard 0.229 / 0.228 s; 1/8 shard 0.122 / 0.121 s).
    if a.streams > 0:
        os.environ["MIPT_STREAMS"] = str(a.streams)
    os.environ.setdefault("MIPT_STREAMS", "1")
    import torch
    import pbrt_v3_spectral_amd as pt
    import importlib.util
    spec = importlib.util.spec_from_file_location("ptdist", os.path.join(ROOT, "pbrt-v3-spectral_amd", "distributed.py"))
    ptdist = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ptdist)

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    rank, world, local_rank = ptdist.init_from_env()
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (a.gpus, a.gpus))
    n_dev = torch.cuda.device_count()
    local_rank = local_rank % max(1, n_dev)   # rehearsal on a 1-GPU box: ranks share device 0
    torch.cuda.set_device(local_rank)

    total_spp = a.spp
    workload = ("procedural-%dtris" % a.procedural_tris) if a.procedural_tris > 0 else os.path.basename(a.scene)
    # The scene is parsed and its BVH built ONCE per job: rank 0 loads the .pbrt text and saves the flat scene as a binary
    # file, the other ranks read the arrays back (mi_scene_save_cache / mi_scene_load_cache) -- eight ranks parsing
    # 200 MB of text and building the same 10M-triangle BVH side by side is what the reference's single process never does.
    # The cache and the generated scene text live in a directory of this job's own (mkdtemp: mode 0700, an unguessable name --
    # never a predictable path in a shared /tmp that another user could have planted), whose name rank 0 hands to the others.
    import shutil
    import tempfile
    job_dir = tempfile.mkdtemp(prefix="mipt_job_") if rank == 0 else None
    job_dir = ptdist.broadcast_string(job_dir)
    cache = os.path.join(job_dir, "scene.bin")
    t_load = time.perf_counter()
    if rank == 0:
        if a.procedural_tris > 0:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import make_procedural_scene as mps
            a.scene = os.path.join(job_dir, "procedural_%d.pbrt" % a.procedural_tris)
            with open(a.scene, "w") as fh:
                mps.write_scene(fh, a.procedural_tris, 700, total_spp, 7, 5)
        scene = pt.Scene(a.scene, spp=total_spp)
        if world > 1:
            scene.save_cache(cache)
    ptdist.barrier()
    if rank != 0:
        scene = pt.Scene(cache=cache)
    t_load = time.perf_counter() - t_load
    ptdist.barrier()
    if rank == 0:
        shutil.rmtree(job_dir, ignore_errors=True)
    integ = pt.CreatePathIntegrator(scene, local_rank)
    w, h = scene.film_size
    film32 = ptdist.device_film_tensor(integ)   # [H, W, 32] view of the renderer's resident film: reduced in place

    def render(si, sc):
        integ.Render(shard_index=si, shard_count=sc, path_pool=a.pool, download=False)

    frame = ptdist.ShardedFrame(render, film32, rank, world)
    step = frame.step   # render this rank's tiles, then ONE RCCL sum-reduce of the film over xGMI (no-op for N = 1)

    for k in range(a.warmup):
        step()
    frame.render_s = frame.reduce_s = 0.0
    frame.steps = 0

    keys = ("camera_rays", "regular_rays", "shadow_rays", "extend_rays", "extend_nodes", "extend_tri_tests",
            "iterations", "bvh_nodes_visited", "tri_tests", "total_paths")
    acc = dict.fromkeys(keys, 0)
    t_kernel = [0.0] * 7
    ptdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(a.steps):
        step()
        c = integ.counters.as_dict()
        for key in keys:
            acc[key] += c[key]
        tk = integ.timings()
        for i in range(7):
            t_kernel[i] += tk[i]
    ptdist.barrier()
    torch.cuda.synchronize()
    dt = ptdist.max_over_ranks(time.perf_counter() - t0)

    per_rank = frame.per_rank_timings()
    sums = ptdist.sum_over_ranks([acc[k] for k in keys])
    tot = dict(zip(keys, sums))
    rays = tot["regular_rays"] + tot["shadow_rays"]
    mrays = rays / dt / 1e6
    msamples = tot["camera_rays"] / dt / 1e6

    # ---- roofline of the dominant kernel class on this rank (closest-hit traversal)
    n_launch = max(1, acc["iterations"])
    ext_rays = max(1, acc["extend_rays"])
    n_node = acc["extend_nodes"] / ext_rays
    n_tri = acc["extend_tri_tests"] / ext_rays
    b_ray = 32.0 * n_node + 48.0 * n_tri + 64.0
    bytes_per_launch = b_ray * ext_rays / n_launch
    avg_launch_s = t_kernel[6] / n_launch
    achieved = bytes_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
    traffic = a.pmc_traffic
    if traffic is None:   # HBM bytes per k_trav<0> launch from the committed rocprofv3 --pmc passes of this workload
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
                tr = json.load(fh).get(workload.replace(".pbrt", ""))
            if tr and tr.get("workload_spp") == a.spp and tr.get("streams") == int(os.environ.get("MIPT_STREAMS", "0")) \
                    and world == 1 and a.pool == 0:
                traffic = tr["k_trav0_hbm_bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            traffic = None
    roofline = {"bound": "hbm", "kernel": "k_trav<0>", "concurrent_streams": int(os.environ.get("MIPT_STREAMS", "1")), "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "bytes_per_ray": round(b_ray, 1), "nodes_per_ray": round(n_node, 2), "tri_tests_per_ray": round(n_tri, 2),
                "rays_per_launch": round(ext_rays / n_launch), "avg_launch_ms": round(avg_launch_s * 1e3, 4),
                "launches": n_launch, "shaded_vertices": int(acc["total_paths"]),
                "kernel_time_s": {"generate": round(t_kernel[1], 4), "trav0": round(t_kernel[6], 4), "extend": round(t_kernel[2], 4),
                                  "shade": round(t_kernel[3], 4), "shadow": round(t_kernel[4], 4),
                                  "mis": round(t_kernel[5], 4), "render_loop": round(t_kernel[0], 4)}}

    # per kernel class: algorithmic bytes / summed class time on this rank. The class times come
    # from HIP events that bracket the traversal kernel together with its resolve kernel, and the classes of
    # the concurrent streams overlap, so these are lower bounds of what each class reaches alone.
    other_rays = max(1, acc["regular_rays"] + acc["shadow_rays"] - acc["extend_rays"])
    other_bytes = 32.0 * (acc["bvh_nodes_visited"] - acc["extend_nodes"]) + 48.0 * (acc["tri_tests"] - acc["extend_tri_tests"]) \
        + (64.0 + 24.0) * other_rays   # (until round 3: + 248 B, when k_resolve_shadow still moved the light sample's spectrum: DESIGN 4)
    vertices = max(1, acc["total_paths"])

    def _cls(nbytes, secs):
        g = nbytes / secs / 1e9 if secs > 0 else 0.0
        return {"achieved": round(g, 1), "frac": round(g / HBM_PEAK_GBS, 4), "unit": "GB/s", "seconds": round(secs, 4)}
    roofline["classes"] = {
        "extend (k_trav<0> + k_resolve_extend)": _cls(b_ray * ext_rays, t_kernel[2]),
        "shadow + mis (k_trav<1,2> + resolves), B_ray + 24 B": _cls(other_bytes, t_kernel[4] + t_kernel[5]),
        "shade (k_shade), 0.96 KB/vertex": _cls(960.0 * vertices, t_kernel[3]),
        "generate (k_generate), 408 B/sample": _cls(408.0 * acc["camera_rays"], t_kernel[1]),
    }

    # The same kernel with the GPU to itself: one untimed 64-spp pass through a one-stream
    # integrator (rank 0, N = 1), so the concurrent-launch figure above has its reference.
    if rank == 0 and world == 1 and a.exclusive_spp > 0 and roofline["concurrent_streams"] > 1:
        os.environ["MIPT_STREAMS"] = "1"
        solo = pt.CreatePathIntegrator(scene, local_rank)
        os.environ["MIPT_STREAMS"] = str(roofline["concurrent_streams"])
        for _ in range(2):   # first pass allocates the pool
            solo.Render(spp=a.exclusive_spp, path_pool=1 << 23, download=False)
        sc_, st_ = solo.counters.as_dict(), solo.timings()
        s_rays = max(1, sc_["extend_rays"])
        s_bray = 32.0 * sc_["extend_nodes"] / s_rays + 48.0 * sc_["extend_tri_tests"] / s_rays + 64.0
        s_gbs = s_bray * s_rays / st_[6] / 1e9
        roofline["exclusive"] = {"streams": 1, "achieved": round(s_gbs, 1), "frac": round(s_gbs / HBM_PEAK_GBS, 4),
                                 "avg_launch_ms": round(st_[6] / max(1, sc_["iterations"]) * 1e3, 4),
                                 "launches": sc_["iterations"], "sample": "%d spp pass, 8M-slot pool" % a.exclusive_spp,
                                 "mray_per_s": round((sc_["regular_rays"] + sc_["shadow_rays"]) / st_[0] / 1e6, 1)}
        del solo

    cpu_baseline = None
    if rank == 0 and world == 1 and a.cpu_samples > 0:
        import oracle_binding as ob
        cores = os.cpu_count() or 1
        try:
            cores = len(os.sched_getaffinity(0))
        except Exception:
            pass
        cores = min(cores, 16)   # a 1-GPU box gives this job a 16-core CPU share
        # a bounded, unbiased sample of the frame: every S-th film tile with all its samples
        n_shards = max(1, round(w * h * total_spp / a.cpu_samples))
        ofilm, oweight, oc, secs = ob.render(scene, n_threads=cores, shard_index=0, shard_count=n_shards)
        orays = oc.regular_rays + oc.shadow_rays
        cpu_baseline = {"value": round(orays / secs / 1e6, 2), "unit": "Mray/s", "cores": cores, "kind": "port",
                        "msamples_per_s": round(oc.camera_rays / secs / 1e6, 3),
                        "sample": "every %d-th 16x16 film tile of the same frame with all %d spp: %d camera samples, %.1f s"
                                  % (n_shards, total_spp, oc.camera_rays, secs)}
        if workload == "killeroo-simple.pbrt":
            # the reference binary itself, measured by the survey in its container: not re-measurable
            # here (the reference does not build in this image without stand-ins for its absent glog submodule)
            cpu_baseline["reference_container"] = {"value": 7.5, "unit": "Mray/s", "threads": 8, "msamples_per_s": 1.28,
                                                   "workload": "killeroo-simple 700x700, 64 spp, maxdepth 5",
                                                   "source": "BASELINE.md section 2"}

    if rank == 0 and a.dump_film:
        import numpy as np
        np.save(a.dump_film, film32.cpu().numpy())
    if rank == 0:
        line = {
            "metric": "Mray/s (%s, PathIntegrator maxdepth %d, Halton, SampledSpectrum-31, %d spp)"
                      % (workload.replace(".pbrt", ""), int(scene.desc.integrator.max_depth), total_spp),
            "value": round(mrays, 1), "unit": "Mray/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic: %s scene, Halton samples" % ("seeded procedural" if a.procedural_tris > 0 else "bundled"),
            "config": {"workload": "%s %dx%d, %d spp per step (one full frame), film tiles sharded over %d GPU(s)"
                                   % (workload, w, h, total_spp, world),
                       "triangles": int(scene.stats["n_triangles"]), "scene_load_s": round(t_load, 2),
                       "spp": total_spp, "resolution": [w, h], "max_depth": int(scene.desc.integrator.max_depth),
                       "path_pool_slots": integ.pool_info()[0], "path_pool_gb": round(integ.pool_info()[1] / 1e9, 2)},
            "msamples_per_s": round(msamples, 2), "rays": int(rays), "camera_samples": int(tot["camera_rays"]),
            "seconds": round(dt, 4),
            "film_mean_per_sample": round(float(film32[..., :31].mean().item()) / total_spp, 6),
            "per_rank": per_rank,
            "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(line))


if __name__ == "__main__":
    main()
