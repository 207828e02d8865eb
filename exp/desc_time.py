"""Time of descriptor_cost_kernel at the BASELINE CNN size (1241x376, K=64, D=128) through sgmx_compute_desc_f32,
HIP events of the data-cost class; algorithmic bytes 2*H*W*K*4 in + H*W*Dp*4 out."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stereo_vision_amd as sv
H, W, K, D = 376, 1241, int(sys.argv[1]) if len(sys.argv) > 1 else 64, 128
rng = np.random.default_rng(1)
fl = rng.standard_normal((H, W, K)).astype(np.float32)
fr = rng.standard_normal((H, W, K)).astype(np.float32)
e = sv.SGM(sv.SGMParams(disp_range=D, penalty1=3, penalty2=60))
e.compute_desc(fl, fr)
e.kernel_stats()
e.set_profiling(True)
for _ in range(5):
    e.compute_desc(fl, fr)
st = e.kernel_stats()
ms, n = st["hamming_volume"]
byt = 2 * H * W * K * 4 + H * W * 128 * 4
print("descriptor_cost_kernel %dx%d K=%d D=%d: %.3f ms per launch (%d launches), %.0f GB/s of %.1f MB algorithmic bytes; "
      "aggregate %.2f ms, wta %.2f ms" % (W, H, K, D, ms / n, n, byt / (ms / n * 1e-3) / 1e9, byt / 1e6,
                                          st["aggregate"][0] / n, st["wta"][0] / n))
