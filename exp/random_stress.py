"""One-off: many more seeded random configurations than tests/test_parity_gpu.py keeps (same generator, same check:
final disparity of the volume-free formulation against the oracle), in batches of three pairs so that the rows of a
wave walk different pairs' geometry too.  usage: exp/random_stress.py [count] [seed]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import numpy as np
import stereo_vision_amd as sv
from conftest import synth_pair
from test_parity_gpu import _random_configs
from oracle import oracle as o
o.build()
count = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 777
bad = 0
for k, (H, W, D, wsz, P1, P2, paths) in enumerate(_random_configs(count, seed)):
    pairs = [synth_pair(H * 977 + W * 13 + D + j, H, W, max(1, min(D, W // 2))) for j in range(3)]
    e = sv.StereoSGM(sv.StereoSGMParams(disp_range=D, window_sz=wsz, penalty1=P1, penalty2=P2, num_paths=paths, max_batch=3))
    got = e.compute_batch(np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs]))
    e.close()
    for j, (l, r) in enumerate(pairs):
        if not np.array_equal(got[j], o.sgm_census(l, r, D, wsz, P1, P2, paths)):
            bad += 1
            print("MISMATCH", (H, W, D, wsz, P1, P2, paths), "pair", j, flush=True)
    if k % 25 == 24:
        print(k + 1, "configurations,", bad, "mismatches", flush=True)
print("done:", count, "configurations,", bad, "mismatches")
sys.exit(1 if bad else 0)
