"""One-off: small and odd shapes (cropped widths and heights of a few pixels, D far above the width), batches of 1..5
pairs, 4 and 8 paths, every lane layout; and a few mid-size images.  Final disparity against the oracle.
usage: exp/random_stress3.py [count] [seed]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import numpy as np
import stereo_vision_amd as sv
from conftest import synth_pair
from oracle import oracle as o
o.build()
count = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rng = np.random.default_rng(seed)
bad = 0
for k in range(count):
    wsz = int(rng.choice([1, 3, 5, 7, 9]))
    if k % 10 == 9:  # mid-size
        H, W = int(rng.integers(60, 140)), int(rng.integers(300, 700))
    else:
        H, W = wsz + int(rng.integers(0, 14)), wsz + int(rng.integers(0, 44))
    D = int(rng.choice([2, 3, 16, 31, 32, 33, 64, 65, 100, 128, 129, 200, 256, 257, 300, 384, 385, 512]))
    P1 = int(rng.integers(0, 30))
    P2 = int(rng.integers(P1, 256)) if rng.random() < 0.7 else int(rng.integers(P1, P1 + 12))
    paths = int(rng.choice([4, 8]))
    n = int(rng.integers(1, 6))
    pairs = [synth_pair(k * 131 + j, H, W, max(1, min(D, W // 2))) for j in range(n)]
    e = sv.StereoSGM(sv.StereoSGMParams(disp_range=D, window_sz=wsz, penalty1=P1, penalty2=P2, num_paths=paths,
                                        max_batch=int(rng.integers(1, n + 1))))
    got = e.compute_batch(np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs]))
    e.close()
    for j, (l, r) in enumerate(pairs):
        if not np.array_equal(got[j], o.sgm_census(l, r, D, wsz, P1, P2, paths)):
            bad += 1
            print("MISMATCH", (H, W, D, wsz, P1, P2, paths), "n", n, "pair", j, flush=True)
    if k % 50 == 49:
        print(k + 1, "configurations,", bad, "mismatches", flush=True)
print("done:", count, "configurations,", bad, "mismatches")
sys.exit(1 if bad else 0)
