"""Timing experiment: do sweeps (VALU-bound) and WTA (HBM-bound) of different pair groups overlap when they
are issued on different streams?  G contexts of B/G pairs each, one stream each, vs one context of B pairs."""
import argparse, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stereo_vision_amd as sv

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--groups", type=int, default=2)
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--stagger", type=int, default=1)
ap.add_argument("--width", type=int, default=1241)
ap.add_argument("--height", type=int, default=376)
ap.add_argument("--disp", type=int, default=128)
a = ap.parse_args()
W, H, B, G = a.width, a.height, a.batch, a.groups
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(1)
L = torch.randint(0, 256, (B, H, W), dtype=torch.uint8, generator=g).to(dev)
R = torch.randint(0, 256, (B, H, W), dtype=torch.uint8, generator=g).to(dev)
out = torch.zeros((B, H, W), dtype=torch.uint16, device=dev)
n = B // G
engs = [sv.StereoSGM(sv.StereoSGMParams(disp_range=a.disp, window_sz=5, penalty1=3, penalty2=60, max_batch=n)) for _ in range(G)]
streams = [torch.cuda.Stream() for _ in range(G)]

def one_step():
    for k in range(G):
        engs[k].compute_device(L[k * n].data_ptr(), R[k * n].data_ptr(), out[k * n].data_ptr(), n, W, H,
                               stream=streams[k].cuda_stream)

def run(steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(steps):
        one_step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

run(5)
if a.stagger and G > 1:
    # start the streams a part of a phase apart: stream k first runs k * stagger pairs alone
    for k in range(1, G):
        m = min(n, k * a.stagger)
        engs[k].compute_device(L[k * n].data_ptr(), R[k * n].data_ptr(), out[k * n].data_ptr(), m, W, H,
                               stream=streams[k].cuda_stream)
ms = run(a.steps)
print("groups %d x %d pairs: %.3f ms per %d pairs -> %.0f frames/s" % (G, n, ms, B, B / ms * 1e3))
