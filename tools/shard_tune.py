"""Time one shard of an N-way split of the 1024-spp frame for several stream counts / pool sizes.
Usage: shard_tune.py [N=8] [streams,comma,separated] [pools in M slots,comma,separated]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import pbrt_v3_spectral_amd as pt

scene = pt.Scene(os.path.join(ROOT, "scenes", "killeroo-simple.pbrt"), spp=1024)
w, h = scene.film_size
film = torch.zeros((h, w, pt.NSPEC), dtype=torch.float32, device="cuda")
weight = torch.zeros((h, w), dtype=torch.float32, device="cuda")
NSH = int(sys.argv[1]) if len(sys.argv) > 1 else 8
STREAMS = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [1, 2, 4]
POOLS = [int(x) << 20 for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else [4 << 20, 8 << 20, 16 << 20, 32 << 20]
for k in STREAMS:
    os.environ["MIPT_STREAMS"] = str(k)
    integ = pt.CreatePathIntegrator(scene, 0)
    for pool in POOLS:
        ts = []
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            integ.Render(shard_index=NSH // 2, shard_count=NSH, path_pool=pool, film_out=film.data_ptr(), weight_out=weight.data_ptr())
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        print("streams %d pool %3dM  shard %d/%d: %.4f s (first %.4f)  iterations %d" % (k, pool >> 20, NSH // 2, NSH, min(ts[1:]), ts[0], integ.counters.iterations))
    del integ
