"""Three steps of the default bench workload (B KITTI-size pairs), for kernel traces."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import stereo_vision_amd as sv
from bench import synthetic_pair
H, W, D, B = 376, 1241, 128, int(sys.argv[1]) if len(sys.argv) > 1 else 32
FLAGS = int(sys.argv[2]) if len(sys.argv) > 2 else 0  # 1 = materialised formulation
ps = [synthetic_pair(1234 + i, H, W, 100) for i in range(8)]
L = torch.from_numpy(np.stack([ps[i % 8][0] for i in range(B)])).cuda(); R = torch.from_numpy(np.stack([ps[i % 8][1] for i in range(B)])).cuda()
out = torch.zeros((B, H, W), dtype=torch.int16, device="cuda")
e = sv.StereoSGM(sv.StereoSGMParams(disp_range=D, window_sz=5, penalty1=3, penalty2=60, max_batch=B, flags=FLAGS))
for _ in range(3):
    e.compute_device(L.data_ptr(), R.data_ptr(), out.data_ptr(), B, W, H, 0)
torch.cuda.synchronize()
