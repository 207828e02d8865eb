import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stereo_vision_amd as sv
from bench import synthetic_pair
H, W, D = 376, 1241, 128
n = 128
l, r = synthetic_pair(1, H, W, 100)
L = np.stack([l] * n); R = np.stack([r] * n)
e = sv.StereoSGM(sv.StereoSGMParams(disp_range=D, window_sz=5, max_batch=32))
e.compute_batch(L[:32], R[:32])
t = time.perf_counter(); out = e.compute_batch(L, R); dt = time.perf_counter() - t
print("host-buffer batch path (pageable numpy, n=%d): %.1f pairs/s" % (n, n / dt))
import torch
Lp = torch.from_numpy(L).pin_memory(); Rp = torch.from_numpy(R).pin_memory()
Op = torch.empty((n, H, W), dtype=torch.int16).pin_memory()
t = time.perf_counter()
e.check(e.lib.sgmx_compute_census_u8_batch(e.h, n, Lp.data_ptr(), Rp.data_ptr(), W, H, Op.data_ptr()))
dt = time.perf_counter() - t
print("host-buffer batch path (pinned host memory, n=%d): %.1f pairs/s, equal: %s" % (n, n / dt, bool((Op.numpy().view(np.uint16) == out).all())))
t = time.perf_counter()
for i in range(20): e.compute(l, r)
dt = time.perf_counter() - t
print("single-pair host calls: %.1f pairs/s (%.2f ms per call)" % (20 / dt, dt / 20 * 1e3))
for flags, name in ((0, "fast"), (1, "materialized")):
    for mb in (1, 2, 4, 8):
        e = sv.StereoSGM(sv.StereoSGMParams(disp_range=D, window_sz=5, max_batch=mb, flags=flags))
        e.compute_batch(L[:mb], R[:mb])
        t = time.perf_counter()
        for i in range(10): e.compute_batch(L[:mb], R[:mb])
        dt = (time.perf_counter() - t) / 10
        e.set_profiling(True); e.compute_batch(L[:mb], R[:mb]); st = e.kernel_stats(); e.set_profiling(False)
        print("%s B=%d: %.2f ms per call, %.1f pairs/s, kernels ms: %s" % (name, mb, dt * 1e3, mb / dt, {k: round(v[0], 3) for k, v in st.items()}))
        e.close()
