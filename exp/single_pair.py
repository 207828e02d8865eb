"""One KITTI-size pair per call (the reference's own call pattern, sgm_single/main.cc:43-44): host entry point with
reused buffers, device-resident call, per-class kernel time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import stereo_vision_amd as sv
from bench import synthetic_pair
H, W, D = 376, 1241, 128
l, r = synthetic_pair(1234, H, W, 100)
e = sv.StereoSGM(sv.StereoSGMParams(disp_range=D, window_sz=5, penalty1=3, penalty2=60, max_batch=1))
out = np.empty((H, W), np.uint16)
def host():
    e.check(e.lib.sgmx_compute_census_u8(e.h, l.ctypes.data, r.ctypes.data, W, H, W, W, out.ctypes.data, 2 * W))
for _ in range(5): host()
N = 50
t0 = time.perf_counter()
for _ in range(N): host()
print("host entry point, pageable buffers: %.3f ms per pair" % ((time.perf_counter() - t0) / N * 1e3))
dl, dr = torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()
do = torch.zeros((H, W), dtype=torch.int16, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def dev():
    e.compute_device(dl.data_ptr(), dr.data_ptr(), do.data_ptr(), 1, W, H, st)
for _ in range(5): dev()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N): dev()
torch.cuda.synchronize()
print("device-resident, back to back: %.3f ms per pair" % ((time.perf_counter() - t0) / N * 1e3))
t0 = time.perf_counter()
for _ in range(N):
    dev(); torch.cuda.synchronize()
print("device-resident, synchronised per pair: %.3f ms per pair" % ((time.perf_counter() - t0) / N * 1e3))
e.kernel_stats(); e.set_profiling(True)
for _ in range(10): dev()
s = e.kernel_stats()
print("kernel ms per pair:", {k: round(v[0] / 10, 4) for k, v in s.items()}, "sum %.3f" % (sum(v[0] for v in s.values()) / 10))
d = e.direction_stats()  # (cleared by the kernel_stats call above: run again for the split)
for _ in range(10): dev()
print("sweep ms per direction:", [round(ms / max(1, n), 4) for ms, n in e.direction_stats()])
e.kernel_stats()
