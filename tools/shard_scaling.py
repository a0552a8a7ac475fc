"""Strong-scaling rehearsal on one GPU: time shard r of N of the 1024-spp frame for each r
(what rank r of an N-GPU job would do, without the film reduce) against the full frame."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import pbrt_v3_spectral_amd as pt

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
scene = pt.Scene(os.path.join(ROOT, "scenes", "killeroo-simple.pbrt"), spp=spp)
integ = pt.CreatePathIntegrator(scene, 0)
w, h = scene.film_size
film = torch.zeros((h, w, pt.NSPEC), dtype=torch.float32, device="cuda")
weight = torch.zeros((h, w), dtype=torch.float32, device="cuda")

def run(si, sc):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    integ.Render(shard_index=si, shard_count=sc, film_out=film.data_ptr(), weight_out=weight.data_ptr())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if os.environ.get("SHARD_VERBOSE"):
        tk = integ.timings()
        print("   shard %d/%d: wall %.4f  render loop %.4f  kernels %.4f (gen %.4f ext %.4f shade %.4f shadow %.4f mis %.4f)  iterations %d"
              % (si, sc, dt, tk[0], sum(tk[1:6]), tk[1], tk[2], tk[3], tk[4], tk[5], integ.counters.as_dict()["iterations"]))
    return dt

run(0, 1)
full = min(run(0, 1) for _ in range(2))
print("full frame %.4f s" % full)
for n in (2, 4, 8):
    run(0, n)
    ts = [run(r, n) for r in range(n)]
    print("N=%d  slowest shard %.4f s  mean %.4f  ideal %.4f  -> speed-up %.2f (eff %.2f)"
          % (n, max(ts), sum(ts) / n, full / n, full / max(ts), full / max(ts) / n))
