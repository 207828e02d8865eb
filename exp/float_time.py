import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stereo_vision_amd as sv
H, W, D, B = 376, 1241, 128, 8
C = (torch.rand((B, H, W, D), device="cuda") * 4).contiguous()
out = torch.zeros((B, H, W), dtype=torch.int16, device="cuda")
e = sv.SGM(sv.SGMParams(disp_range=D, penalty1=3, penalty2=60, max_batch=B))
f = lambda: e.compute_cost_device(C.data_ptr(), out.data_ptr(), B, W, H, 0)
f(); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5): f()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 5
print("float volume B=%d: %.2f ms per step, %.1f volumes/s" % (B, dt * 1e3, B / dt))
