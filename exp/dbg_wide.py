"""Where the fast path's raw disparity differs from the oracle for a wide image at D > 256 (debug aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from conftest import synth_pair
import stereo_vision_amd as sv
from oracle import oracle as o
o.build()
for (H, W, D, wsz, P1, P2, paths) in [(14, 700, 512, 3, 5, 255, 8), (16, 520, 270, 5, 7, 60, 8), (14, 640, 384, 5, 3, 60, 8)]:
    left, right = synth_pair(H * 1000 + W + D, H, W, min(D, W // 2))
    e = sv.StereoSGM(sv.StereoSGMParams(disp_range=D, window_sz=wsz, penalty1=P1, penalty2=P2, num_paths=paths, flags=2))
    disp = e.compute(left, right)
    want = o.sgm_census(left, right, D, wsz, P1, P2, paths, taps=True)
    st = e.stages(H, W)
    print("case", (H, W, D), "C", np.array_equal(st["C"], want["C"]), "S", np.array_equal(st["S"].astype(np.uint32), want["S"]))
    bad = np.argwhere(st["raw"] != want["raw"])
    print("  raw mismatches:", len(bad), "of", st["raw"].size)
    if len(bad):
        xs = np.unique(bad[:, 1]); ys = np.unique(bad[:, 0])
        print("  x range", xs.min(), xs.max(), "first xs", xs[:20], "ys", ys[:20])
        for (y, x) in bad[:8]:
            print("   ", y, x, "got", st["raw"][y, x], st["raw"][y, x] / 256.0, "want", want["raw"][y, x], want["raw"][y, x] / 256.0)
