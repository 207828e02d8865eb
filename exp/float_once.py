"""One warm-up and two timed steps of the float-volume path (KITTI size, B=8), for kernel traces."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stereo_vision_amd as sv
H, W, D, B = 376, 1241, 128, int(sys.argv[1]) if len(sys.argv) > 1 else 8
C = (torch.rand((B, H, W, D), device="cuda") * 4).contiguous()
out = torch.zeros((B, H, W), dtype=torch.int16, device="cuda")
e = sv.SGM(sv.SGMParams(disp_range=D, penalty1=3, penalty2=60, max_batch=B))
for _ in range(3):
    e.compute_cost_device(C.data_ptr(), out.data_ptr(), B, W, H, 0)
torch.cuda.synchronize()
