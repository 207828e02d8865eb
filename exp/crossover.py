"""Small launch groups: materialised (one path per wave) against the fast formulation, KITTI size, inputs resident."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import stereo_vision_amd as sv
from bench import synthetic_pair
W, H, D = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (1241, 376, 128)
BS = [int(v) for v in sys.argv[4:]] or [1, 2, 3, 4, 6, 8]
l, r = synthetic_pair(1, H, W, min(100, D - 4))
for B in BS:
    L = torch.from_numpy(np.stack([l] * B)).cuda(); R = torch.from_numpy(np.stack([r] * B)).cuda()
    out = torch.zeros((B, H, W), dtype=torch.int16, device="cuda")
    res = []
    for flags in (1, 2):
        e = sv.StereoSGM(sv.StereoSGMParams(disp_range=D, window_sz=5, penalty1=3, penalty2=60, max_batch=B, flags=flags))
        f = lambda: e.compute_device(L.data_ptr(), R.data_ptr(), out.data_ptr(), B, W, H, 0)
        f(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10): f()
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t) / 10 * 1e3)
        e.close()
    print("B=%d: materialised %.2f ms, fast %.2f ms" % (B, res[0], res[1]))
