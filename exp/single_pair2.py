"""Why does bench.py's single_pair_ms read higher than exp/single_pair.py?  Same call, with and without a second, large
context alive in the process, and with the bench's own input pair."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import stereo_vision_amd as sv
from bench import synthetic_pair
H, W, D = 376, 1241, 128
def measure(tag, l, r, reps=50):
    e = sv.StereoSGM(sv.StereoSGMParams(disp_range=D, window_sz=5, penalty1=3, penalty2=60, max_batch=1))
    out = np.empty((H, W), np.uint16)
    f = lambda: e.check(e.lib.sgmx_compute_census_u8(e.h, l.ctypes.data, r.ctypes.data, W, H, W, W, out.ctypes.data, 2 * W))
    for _ in range(5): f()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    print("%-50s %.3f ms" % (tag, (time.perf_counter() - t0) / reps * 1e3), flush=True)
    e.close()
l, r = synthetic_pair(1234, H, W, 100)
measure("alone", l, r)
big = sv.StereoSGM(sv.StereoSGMParams(disp_range=D, window_sz=5, penalty1=3, penalty2=60, max_batch=32))
L = torch.zeros((32, H, W), dtype=torch.uint8, device="cuda"); O = torch.zeros((32, H, W), dtype=torch.int16, device="cuda")
big.compute_device(L.data_ptr(), L.data_ptr(), O.data_ptr(), 32, W, H, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
measure("with a 32-pair context alive (16 GB of scratch)", l, r)
ls = np.ascontiguousarray(np.stack([l] * 4)[2]); rs = np.ascontiguousarray(np.stack([r] * 4)[2])
measure("inputs that are slices of a larger array's copy", ls, rs)
measure("30 repetitions", l, r, 30)
big.close()
measure("after closing the large context", l, r)
