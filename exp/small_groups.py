"""Device-resident calls of 1..12 KITTI-size pairs (or --size W H D): ms per call, back to back.  SGMX_LIB selects a build
(e.g. one with -DSGMX_FUSE=0: no directions side by side)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import stereo_vision_amd as sv
from bench import synthetic_pair
W, H, D = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (1241, 376, 128)
l, r = synthetic_pair(1234, H, W, min(100, D - 1))
res = []
for n in ((1, 2, 3) if W * H > 4000000 else (1, 2, 3, 4, 6, 8, 12)):
    e = sv.StereoSGM(sv.StereoSGMParams(disp_range=D, window_sz=5, penalty1=3, penalty2=60, max_batch=n))
    dl = torch.from_numpy(np.stack([l] * n)).cuda(); dr = torch.from_numpy(np.stack([r] * n)).cuda()
    do = torch.zeros((n, H, W), dtype=torch.int16, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    f = lambda: e.compute_device(dl.data_ptr(), dr.data_ptr(), do.data_ptr(), n, W, H, st)
    for _ in range(5): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 6 if W * H > 4000000 else 40
    for _ in range(reps): f()
    torch.cuda.synchronize()
    res.append("%d: %.3f" % (n, (time.perf_counter() - t0) / reps * 1e3))
    e.close()
print("%dx%d D=%d, ms per call by pairs per call:" % (W, H, D), "  ".join(res))
