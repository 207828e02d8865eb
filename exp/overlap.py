"""Experiment: two engines on two streams, so one half's HBM-bound WTA overlaps the other half's VALU-bound sweeps.
usage: python exp/overlap.py B_per_engine n_engines steps"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import stereo_vision_amd as sv
from bench import synthetic_pair
B, NE, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
W, H, D = 1241, 376, 128
pairs = [synthetic_pair(1234 + i, H, W, 100) for i in range(8)]
Lh = np.stack([pairs[i % 8][0] for i in range(B)]); Rh = np.stack([pairs[i % 8][1] for i in range(B)])
engs, bufs, streams = [], [], []
for e in range(NE):
    engs.append(sv.StereoSGM(sv.StereoSGMParams(disp_range=D, window_sz=5, penalty1=3, penalty2=60, num_paths=8, max_batch=B)))
    bufs.append((torch.from_numpy(Lh).cuda(), torch.from_numpy(Rh).cuda(), torch.zeros((B, H, W), dtype=torch.int16, device="cuda")))
    streams.append(torch.cuda.Stream())
def step():
    for e in range(NE):
        dL, dR, dO = bufs[e]
        engs[e].compute_device(dL.data_ptr(), dR.data_ptr(), dO.data_ptr(), B, W, H, streams[e].cuda_stream)
for _ in range(2): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps): step()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("B=%d x %d engines: %.1f frames/s, %.3f ms per %d pairs" % (B, NE, B * NE * steps / dt, dt / steps * 1e3, B * NE))
same = all(torch.equal(bufs[0][2], bufs[e][2]) for e in range(NE))
print("outputs identical across engines:", same)
