"""Two steps of the 4K D=256 census configuration (B pairs), for kernel traces."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import stereo_vision_amd as sv
from bench import synthetic_pair
H, W, D, B = 2160, 3840, 256, int(sys.argv[1]) if len(sys.argv) > 1 else 2
l, r = synthetic_pair(7, H, W, 240)
L = torch.from_numpy(np.stack([l] * B)).cuda(); R = torch.from_numpy(np.stack([r] * B)).cuda()
out = torch.zeros((B, H, W), dtype=torch.int16, device="cuda")
e = sv.StereoSGM(sv.StereoSGMParams(disp_range=D, window_sz=5, penalty1=3, penalty2=60, max_batch=B))
for _ in range(2):
    e.compute_device(L.data_ptr(), R.data_ptr(), out.data_ptr(), B, W, H, 0)
torch.cuda.synchronize()
