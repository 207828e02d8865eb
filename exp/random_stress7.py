"""One-off: the no-allocation entry point against the allocating device entry point on random sizes, disparity ranges,
path counts, pair counts and max_batch values that fit its 64 MiB scratch (tests/test_zz_no_alloc.py's comparison)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import numpy as np, torch
import stereo_vision_amd as sv
from conftest import synth_pair
import test_zz_no_alloc as Z
count = int(sys.argv[1]) if len(sys.argv) > 1 else 80
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 21)
bad = done = 0
while done < count:
    D = int(rng.choice([8, 16, 32, 33, 64, 100, 128, 200, 256]))
    paths = int(rng.choice([4, 8]))
    H, W = int(rng.integers(8, 120)), int(rng.integers(8, 260))
    RP = 1 if D <= 32 else 2 if D <= 64 else 4 if D <= 128 else 8
    if (H - 4) * (W - 4) * 32 * RP * paths > 40 << 20:  # one pair has to fit the scratch with room to spare
        continue
    n = int(rng.integers(1, 6))
    pairs = [synth_pair(done * 13 + j, H, W, max(1, min(D, W // 2))) for j in range(n)]
    got, want = Z._both(sv, torch, pairs, D, paths, int(rng.integers(1, 6)))
    done += 1
    if not np.array_equal(got, want):
        bad += 1
        print("MISMATCH", (H, W, D, paths, n), flush=True)
print("noalloc:", count, "configurations,", bad, "mismatches")
sys.exit(1 if bad else 0)
