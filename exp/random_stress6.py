"""One-off: long paths (widths up to 2600, heights up to 300), every lane layout, 4 / 8 paths, batches of 2, against the oracle."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import numpy as np
import stereo_vision_amd as sv
from conftest import synth_pair
from oracle import oracle as o
o.build()
count = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
bad = 0
for k in range(count):
    wsz = int(rng.choice([3, 5, 7]))
    if k % 2:
        H, W = int(rng.integers(8, 40)), int(rng.integers(600, 2600))
    else:
        H, W = int(rng.integers(150, 300)), int(rng.integers(40, 400))
    D = int(rng.choice([16, 32, 64, 100, 128, 130, 256, 300, 512]))
    P1 = int(rng.integers(0, 20)); P2 = int(rng.integers(P1, 200))
    paths = int(rng.choice([4, 8]))
    pairs = [synth_pair(k * 17 + j, H, W, max(1, min(D, W // 2))) for j in range(2)]
    e = sv.StereoSGM(sv.StereoSGMParams(disp_range=D, window_sz=wsz, penalty1=P1, penalty2=P2, num_paths=paths, max_batch=2))
    got = e.compute_batch(np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs]))
    e.close()
    for j, (l, r) in enumerate(pairs):
        if not np.array_equal(got[j], o.sgm_census(l, r, D, wsz, P1, P2, paths)):
            bad += 1
            print("MISMATCH", (H, W, D, wsz, P1, P2, paths), "pair", j, flush=True)
    print(k + 1, (H, W, D, wsz, P1, P2, paths), "ok" if not bad else bad, flush=True)
print("done:", count, "configurations,", bad, "mismatches")
sys.exit(1 if bad else 0)
