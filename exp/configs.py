"""Timings of the other BASELINE configurations (not bench lines): float cost volume at KITTI size,
4K D=256 census, Tsukuba D=16 4 paths.  Inputs resident in HBM."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import stereo_vision_amd as sv
from bench import synthetic_pair

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n

# float volume, KITTI size
H, W, D, B = 376, 1241, 128, 8
C = (torch.rand((B, H, W, D), device="cuda") * 4).contiguous()
out = torch.zeros((B, H, W), dtype=torch.int16, device="cuda")
e = sv.SGM(sv.SGMParams(disp_range=D, penalty1=3, penalty2=60, max_batch=B))
dt = timeit(lambda: e.compute_cost_device(C.data_ptr(), out.data_ptr(), B, W, H, 0))
print("float volume 1241x376x128, B=%d: %.2f ms per step, %.1f volumes/s, alg %.2f TB/s (96 B/voxel)" % (B, dt * 1e3, B / dt, 96 * H * W * D * B / dt / 1e12))
e.close(); del C

# SGMStereo pipeline (new_sgm) on a float volume, KITTI size
B = 4
C = (torch.rand((B, H, W, D), device="cuda") * 4).contiguous()
out = torch.zeros((B, H, W), dtype=torch.int16, device="cuda")
e2 = sv.SGMStereo(max_batch=B)
e2.SetSmoothnessCostParameters(3, 60)
dt = timeit(lambda: e2.compute_cost_device(C.data_ptr(), out.data_ptr(), B, W, H, D, 0))
e2.engine.set_profiling(True)
e2.compute_cost_device(C.data_ptr(), out.data_ptr(), B, W, H, D, 0)
print("SGMStereo float volume 1241x376x128, B=%d: %.2f ms per step, %.1f volumes/s; kernel classes (ms): %s" %
      (B, dt * 1e3, B / dt, {k: round(v[0], 3) for k, v in e2.engine.kernel_stats().items()}))
del C, e2

# 4K census D=256
H, W, D = 2160, 3840, 256
for B in (1, 2, 4, 8):
    l, r = synthetic_pair(7, H, W, 240)
    L = torch.from_numpy(np.stack([l] * B)).cuda(); R = torch.from_numpy(np.stack([r] * B)).cuda()
    out = torch.zeros((B, H, W), dtype=torch.int16, device="cuda")
    e = sv.StereoSGM(sv.StereoSGMParams(disp_range=D, window_sz=5, penalty1=3, penalty2=60, max_batch=B))
    dt = timeit(lambda: e.compute_device(L.data_ptr(), R.data_ptr(), out.data_ptr(), B, W, H, 0), 3)
    V = (W - 4) * (H - 4) * D
    print("4K 3840x2160 D=256 census, B=%d: %.1f ms per step, %.2f frames/s, %.0f MDE/s, alg %.2f TB/s (41 B/voxel)" % (B, dt * 1e3, B / dt, B * W * H * D / dt / 1e6, 41 * V * B / dt / 1e12))
    e.close()

# Tsukuba-size D=16, 4 paths
H, W, D, B = 288, 384, 16, 64
l, r = synthetic_pair(3, H, W, 12)
L = torch.from_numpy(np.stack([l] * B)).cuda(); R = torch.from_numpy(np.stack([r] * B)).cuda()
out = torch.zeros((B, H, W), dtype=torch.int16, device="cuda")
e = sv.StereoSGM(sv.StereoSGMParams(disp_range=D, window_sz=5, penalty1=3, penalty2=60, num_paths=4, max_batch=B))
dt = timeit(lambda: e.compute_device(L.data_ptr(), R.data_ptr(), out.data_ptr(), B, W, H, 0))
print("Tsukuba-size 384x288 D=16 4 paths, B=%d: %.2f ms per step, %.0f frames/s" % (B, dt * 1e3, B / dt))
