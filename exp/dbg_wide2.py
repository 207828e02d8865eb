"""Debug builds (SGMX_DBG_WTA = 1: right-view argmin, 2: left argmin, 3: right-view min cost as the raw image) against numpy on the oracle's S."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from conftest import synth_pair
import stereo_vision_amd as sv
from oracle import oracle as o
o.build()
mode = int(sys.argv[1])
for (H, W, D, wsz, P1, P2, paths) in [(14, 700, 512, 3, 5, 255, 8), (16, 520, 270, 5, 7, 60, 8)]:
    left, right = synth_pair(H * 1000 + W + D, H, W, min(D, W // 2))
    e = sv.StereoSGM(sv.StereoSGMParams(disp_range=D, window_sz=wsz, penalty1=P1, penalty2=P2, num_paths=paths, flags=2))
    e.compute(left, right)
    want = o.sgm_census(left, right, D, wsz, P1, P2, paths, taps=True)
    m = (wsz - 1) // 2
    got = e.stages(H, W)["raw"][m:H - m, m:W - m].astype(np.int64)
    S = want["S"].astype(np.int64)
    h, w, _ = S.shape
    if mode == 2:
        ref = S.argmin(axis=2)
    elif mode in (4, 5, 6):
        am = S.argmin(axis=2)
        ok = (am >= 1) & (am < D - 1)
        idx = np.clip(am + (mode - 5), 0, D - 1)
        ref = np.take_along_axis(S, idx[..., None], axis=2)[..., 0]
        ref = np.where(ok, ref, got)
    else:
        ref = np.zeros((h, w), np.int64)
        for y in range(h):
            for xr in range(w):
                n = min(D, w - xr)
                v = S[y, xr + np.arange(n), np.arange(n)]
                ref[y, xr] = v.argmin() if mode == 1 else v.min()
    bad = np.argwhere(got != ref)
    print("mode", mode, "case", (H, W, D), "mismatches", len(bad))
    for (y, x) in bad[:12]:
        print("   y", y, "x", x, "got", got[y, x], "ref", ref[y, x], "w - x", w - x)
