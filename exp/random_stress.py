"""Seeded random configurations far beyond what tests/ keeps, every result against the oracle (census, float, SGMStereo,
descriptor and Hamming-library entry points) or against the allocating entry point (no-allocation call).

usage: exp/random_stress.py MODE [count] [seed]

  census3   the suite's generator (tests/test_parity_gpu._random_configs), three pairs per launch
  both      the suite's seeded tests themselves: both census formulations, float volumes
  small     cropped sizes of a few pixels, D far above the width, 1..5 pairs, max_batch 1..n; every tenth mid-size
  long      long paths: wide-and-flat and tall-and-narrow images (diagonals that wrap several times), two pairs
  tall      heights of 2..12 times the width: wrapped diagonals with many wraps per path, three pairs
  ns        recon::SGMStereo pipeline and the descriptor path
  hamming   the Hamming library (matrix, matchers, cross matchers) against the restatement and core/hamming.c
  noalloc   sgmx_compute_census_u8_dev_noalloc against sgmx_compute_census_u8_dev
"""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import numpy as np
import stereo_vision_amd as sv
from conftest import synth_pair
from oracle import oracle as o

o.build()
mode = sys.argv[1] if len(sys.argv) > 1 else "census3"
DEFAULTS = {"census3": (200, 777), "both": (100, 31337), "small": (200, 5), "long": (40, 3), "tall": (40, 9), "ns": (60, 99),
            "hamming": (150, 11), "noalloc": (80, 21)}
count = int(sys.argv[2]) if len(sys.argv) > 2 else DEFAULTS[mode][0]
seed = int(sys.argv[3]) if len(sys.argv) > 3 else DEFAULTS[mode][1]
rng = np.random.default_rng(seed)
bad = 0


def census_case(cfg, n, max_batch, seed0):
    """n pairs of one configuration through compute_batch, each against the oracle."""
    global bad
    H, W, D, wsz, P1, P2, paths = cfg
    pairs = [synth_pair(seed0 + j, H, W, max(1, min(D, W // 2))) for j in range(n)]
    e = sv.StereoSGM(sv.StereoSGMParams(disp_range=D, window_sz=wsz, penalty1=P1, penalty2=P2, num_paths=paths, max_batch=max_batch))
    got = e.compute_batch(np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs]))
    e.close()
    for j, (l, r) in enumerate(pairs):
        if not np.array_equal(got[j], o.sgm_census(l, r, D, wsz, P1, P2, paths)):
            bad += 1
            print("MISMATCH", cfg, "n", n, "pair", j, flush=True)


def guarded(name, fn, *args):
    global bad
    try:
        fn(*args)
    except AssertionError:
        bad += 1
        print("FAIL", name, args[-1], traceback.format_exc().splitlines()[-1], flush=True)


if mode == "census3":
    from test_parity_gpu import _random_configs
    for k, cfg in enumerate(_random_configs(count, seed)):
        census_case(cfg, 3, 3, cfg[0] * 977 + cfg[1] * 13 + cfg[2])
        if k % 25 == 24:
            print(k + 1, "configurations,", bad, "mismatches", flush=True)
elif mode == "both":
    import test_parity_gpu as T
    for name, gen, fn in (("census both formulations", T._random_configs, T.test_seeded_random_configurations),
                          ("float volumes", T._random_float_configs, T.test_seeded_random_float_volumes)):
        for case in gen(count, seed):
            guarded(name, fn, o, case)
        print(name, ":", count, "configurations,", bad, "failures so far", flush=True)
elif mode == "small":
    for k in range(count):
        wsz = int(rng.choice([1, 3, 5, 7, 9]))
        if k % 10 == 9:  # mid-size
            H, W = int(rng.integers(60, 140)), int(rng.integers(300, 700))
        else:
            H, W = wsz + int(rng.integers(0, 14)), wsz + int(rng.integers(0, 44))
        D = int(rng.choice([2, 3, 16, 31, 32, 33, 64, 65, 100, 128, 129, 200, 256, 257, 300, 384, 385, 512]))
        P1 = int(rng.integers(0, 30))
        P2 = int(rng.integers(P1, 256)) if rng.random() < 0.7 else int(rng.integers(P1, P1 + 12))
        n = int(rng.integers(1, 6))
        census_case((H, W, D, wsz, P1, P2, int(rng.choice([4, 8]))), n, int(rng.integers(1, n + 1)), k * 131)
        if k % 50 == 49:
            print(k + 1, "configurations,", bad, "mismatches", flush=True)
elif mode in ("long", "tall"):
    for k in range(count):
        wsz = int(rng.choice([3, 5, 7]))
        if mode == "tall":
            W = wsz + int(rng.integers(0, 60))
            H = wsz - 1 + (W - wsz + 1) * int(rng.integers(2, 13)) + int(rng.integers(0, 5))
        elif k % 2:
            H, W = int(rng.integers(8, 40)), int(rng.integers(600, 2600))
        else:
            H, W = int(rng.integers(150, 300)), int(rng.integers(40, 400))
        D = int(rng.choice([16, 32, 64, 100, 128, 130, 256, 300, 512]))
        P1 = int(rng.integers(0, 20))
        cfg = (H, W, D, wsz, P1, int(rng.integers(P1, 200)), int(rng.choice([4, 8])))
        census_case(cfg, 3 if mode == "tall" else 2, 3 if mode == "tall" else 2, k * 17)
        print(k + 1, cfg, "ok" if not bad else bad, flush=True)
elif mode == "ns":
    import test_sgmstereo_gpu as N
    import test_drivers_gpu as G
    for case in N._random_ns_configs(count, seed):
        guarded("sgmstereo", N.test_seeded_random_configurations, o, case)
    print("sgmstereo:", count, "configurations,", bad, "failures", flush=True)
    for k in range(count):
        case = (int(rng.integers(1, 14)), int(rng.integers(2, 320)), int(rng.choice([1, 3, 4, 8, 30, 32, 36, 64, 70])),
                int(rng.choice([2, 16, 33, 100, 128, 129, 200, 256, 300, 512])))
        guarded("descriptor", G.test_descriptor_path, o, case)
elif mode == "hamming":
    e = sv.StereoSGM(sv.StereoSGMParams())

    def one(case):
        ncodes, n1, n2, ht = case
        a = rng.integers(0, 256, size=(n1, ncodes)).astype(np.uint8)
        b = rng.integers(0, 256, size=(n2, ncodes)).astype(np.uint8)
        if n1 > 3 and n2 > 3:
            b[n2 // 2] = a[n1 // 3]
            a[n1 - 1] = a[1]
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

    for k in range(count):
        ncodes = int(rng.choice([1, 2, 3, 4, 5, 8, 12, 16, 17, 32, 33, 64]))
        guarded("hamming", one, (ncodes, int(rng.integers(0, 700)), int(rng.integers(0, 700)), int(rng.integers(0, 4 * ncodes + 2))))
elif mode == "noalloc":
    import torch
    import test_zz_no_alloc as Z
    done = 0
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
else:
    raise SystemExit(__doc__)
print("done:", mode, count, "configurations,", bad, "mismatches / failures")
sys.exit(1 if bad else 0)
