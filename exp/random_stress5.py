"""One-off: the Hamming library (matrix, matchers, cross matchers, allocating / threaded forms) on random code
lengths, thresholds and set sizes against the restatement and the reference's own core/hamming.c (when built)."""
import sys, os, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import numpy as np
import stereo_vision_amd as sv
from oracle import oracle as o
o.build()
count = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
e = sv.StereoSGM(sv.StereoSGMParams())
bad = 0
for k in range(count):
    ncodes = int(rng.choice([1, 2, 3, 4, 5, 8, 12, 16, 17, 32, 33, 64]))
    n1, n2 = int(rng.integers(0, 700)), int(rng.integers(0, 700))
    ht = int(rng.integers(0, 4 * ncodes + 2))
    a = rng.integers(0, 256, size=(n1, ncodes)).astype(np.uint8)
    b = rng.integers(0, 256, size=(n2, ncodes)).astype(np.uint8)
    if n1 > 3 and n2 > 3:
        b[n2 // 2] = a[n1 // 3]
        a[n1 - 1] = a[1]
    try:
        if n1 and n2:
            assert np.array_equal(e.hamming_matrix(a, b), o.hamming_matrix(a, b)), "matrix"
        for cross in (False, True):
            if cross:
                n = e.crossmatch_hamming_count(a, ht)
                idx, hams = e.crossmatch_hamming(a, ht)
            else:
                n = e.match_hamming_count(a, b, ht)
                idx, hams = e.match_hamming_thres(a, b, ht)
            want = o.match_hamming(a, b, ht, cross)
            assert n == len(want[1]), "count"
            assert np.array_equal(idx, want[0]) and np.array_equal(hams, want[1]), "matches"
            ref = o.ref_match_hamming(a, b, ht, cross) if (n1 and (cross or n2)) else None
            if ref is not None:
                assert np.array_equal(idx, ref[0]) and np.array_equal(hams, ref[1]), "reference"
    except AssertionError:
        bad += 1
        print("FAIL", (ncodes, n1, n2, ht), traceback.format_exc().splitlines()[-1], flush=True)
print("hamming:", count, "configurations,", bad, "failures")
sys.exit(1 if bad else 0)
