"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same
inputs.  Bit-exact on the census/Hamming path; the float path is held to
north_star's tolerance (aggregated costs within 1e-5 relative, >= 99.9 % identical
disparities) and in practice is bit-exact too."""
import numpy as np
import pytest

from conftest import synth_pair

pytestmark = pytest.mark.gpu

sv = pytest.importorskip("stereo_vision_amd")


def make(cls=None, **kw):
    cls = cls or sv.StereoSGM
    return cls(sv.StereoSGMParams(**kw))


# ---------------------------------------------------------------- stage taps
@pytest.mark.parametrize("wsz", [1, 3, 5, 7, 9])
def test_census_tap(oracle, wsz):
    rng = np.random.default_rng(wsz)
    img = rng.integers(0, 256, size=(37, 53)).astype(np.uint8)
    e = make(window_sz=wsz, disp_range=16)
    assert np.array_equal(e.census(img), oracle.census(img, wsz))


@pytest.mark.parametrize("D", [2, 16, 100, 128, 129, 255, 256, 270, 512])
def test_hamming_volume_tap(oracle, D):
    rng = np.random.default_rng(D)
    cl = rng.integers(0, 1 << 32, size=(5, 300), dtype=np.uint64).astype(np.uint32)
    cr = rng.integers(0, 1 << 32, size=(5, 300), dtype=np.uint64).astype(np.uint32)
    e = make(disp_range=D)
    assert np.array_equal(e.hamming_volume(cl, cr), oracle.hamming_volume(cl, cr, D))


def test_hamming_volume_tap_vs_reference_popcount(oracle, tsukuba):
    """The GPU's data-cost volume (stereo_sgm.cc:43,75-101) on the census codes of the reference's Tsukuba fixture against
    the volume whose popcounts come from the reference's own compiled core/hamming.c (oracle/_ref)."""
    from test_oracle_golden import _cost_volume_from_reference_hamming
    left, right = tsukuba
    e = make(disp_range=16)
    cl = np.ascontiguousarray(e.census(left[90:120]))
    cr = np.ascontiguousarray(e.census(right[90:120]))
    want = _cost_volume_from_reference_hamming(oracle, cl, cr, 16)
    if want is None:
        pytest.skip("oracle/_ref/libhamming_ref.so not built (reference tree not mounted)")
    assert np.array_equal(e.hamming_volume(cl, cr), want)
    e.close()


@pytest.mark.parametrize("ncodes", [4, 8, 16, 5, 32])
def test_hamming_matrix(oracle, ncodes):
    rng = np.random.default_rng(ncodes)
    a = rng.integers(0, 256, size=(301, ncodes)).astype(np.uint8)
    b = rng.integers(0, 256, size=(77, ncodes)).astype(np.uint8)
    e = make()
    got = e.hamming_matrix(a, b)
    assert np.array_equal(got, oracle.hamming_matrix(a, b))
    ref = oracle.ref_hamming_matrix(a, b)  # the reference's own core/hamming.c, when built
    if ref is not None:
        assert np.array_equal(got, ref)


@pytest.mark.parametrize("ncodes,ht", [(4, 10), (8, 24), (16, 52), (5, 16), (4, 0), (32, 300)])
@pytest.mark.parametrize("cross", [False, True])
def test_hamming_matchers(oracle, ncodes, ht, cross):
    """match_hamming_count / _thres_prealloc and the crossmatch forms (core/hamming.h:73-104): same
    matches in the same order as the restatement and as the reference's own core/hamming.c."""
    rng = np.random.default_rng(ncodes + 7 * ht)
    a = rng.integers(0, 256, size=(333, ncodes)).astype(np.uint8)
    b = rng.integers(0, 256, size=(209, ncodes)).astype(np.uint8)
    b[17] = a[5]
    a[200] = a[11]
    e = make()
    if cross:
        n = e.crossmatch_hamming_count(a, ht)
        idx, hams = e.crossmatch_hamming(a, ht)
    else:
        n = e.match_hamming_count(a, b, ht)
        idx, hams = e.match_hamming_thres(a, b, ht)
    want = oracle.match_hamming(a, b, ht, cross)
    assert n == len(want[1])
    assert np.array_equal(idx, want[0]) and np.array_equal(hams, want[1])
    ref = oracle.ref_match_hamming(a, b, ht, cross)  # the reference's own core/hamming.c, when built
    if ref is not None:
        assert np.array_equal(idx, ref[0]) and np.array_equal(hams, ref[1])


def test_hamming_matchers_capacity_and_empty(oracle):
    """Too small a buffer: nothing written, the needed count reported, an error returned.  Empty sets: 0."""
    import ctypes as C
    rng = np.random.default_rng(3)
    a = rng.integers(0, 256, size=(50, 4)).astype(np.uint8)
    e = make()
    n = e.match_hamming_count(a, a, 12)
    assert n >= 50
    idx = np.full((n, 2), -1, np.int32)
    hams = np.zeros((n,), np.uint16)
    got = C.c_size_t(0)
    rc = e.lib.sgmx_match_hamming_thres_prealloc(e.h, a.ctypes.data, a.ctypes.data, 50, 50, 12, 4, idx.ctypes.data,
                                                 hams.ctypes.data, n - 1, C.byref(got))
    assert rc != 0 and got.value == n and (idx == -1).all()
    assert e.match_hamming_count(a[:0], a, 12) == 0
    assert e.crossmatch_hamming_count(a[:1], 32) == 0


# ---------------------------------------------------------------- full pipeline, every stage
CASES = [
    # H, W, D, wsz, P1, P2, paths
    (40, 75, 16, 5, 3, 60, 8),
    (40, 75, 16, 5, 3, 60, 4),
    (33, 130, 64, 3, 1, 20, 8),
    (29, 141, 100, 5, 7, 60, 8),
    (31, 150, 128, 5, 3, 60, 8),
    (31, 150, 128, 1, 10, 250, 8),
    (23, 97, 128, 5, 20, 20, 4),
    (27, 160, 255, 3, 3, 60, 8),
    (27, 160, 256, 5, 15, 100, 8),
    (21, 90, 270, 5, 7, 60, 8),
    (19, 70, 512, 3, 5, 5000, 8),
    (19, 70, 512, 3, 5, 255, 8),
    # wider than D: the unmasked steady-state loops of the grouped layouts (8, 12, 16 packed pairs per lane)
    (22, 430, 256, 5, 3, 60, 8),
    (18, 470, 255, 3, 7, 60, 4),
    (16, 520, 270, 5, 7, 60, 8),
    (14, 700, 512, 3, 5, 255, 8),
    (14, 640, 384, 5, 3, 60, 8),
    (12, 9, 16, 5, 0, 0, 8),
    (64, 7, 32, 3, 3, 7044, 8),
    (64, 7, 32, 3, 200, 255, 4),
    (9, 300, 128, 9, 3, 60, 8),
]


@pytest.mark.parametrize("flags", [0, 1], ids=["fast", "materialized"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c)))
def test_pipeline_stages_bit_exact(oracle, case, flags):
    """Every stage of both census formulations: the default one (C recomputed from the census
    images, one byte per voxel and direction; used when P2 <= 255) and the materialised C/S one."""
    H, W, D, wsz, P1, P2, paths = case
    left, right = synth_pair(H * 1000 + W + D, H, W, min(D, W // 2))
    e = make(disp_range=D, window_sz=wsz, penalty1=P1, penalty2=P2, num_paths=paths, flags=flags)
    disp = e.compute(left, right)
    want = oracle.sgm_census(left, right, D, wsz, P1, P2, paths, taps=True)
    st = e.stages(H, W)
    assert np.array_equal(st["C"], want["C"]), "data cost"
    assert np.array_equal(st["S"].astype(np.uint32), want["S"]), "summed path costs"
    assert np.array_equal(st["raw"], want["raw"]), "WTA / LR / subpixel"
    assert np.array_equal(disp, want["disp"]), "median"
    m = (wsz - 1) // 2
    assert np.array_equal(e.data_cost_image(H, W), oracle.data_cost_wta(want["C"], m))


def _random_configs(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        wsz = int(rng.choice([1, 3, 5, 7]))
        H = int(rng.integers(wsz, 48)) + 1
        W = int(rng.integers(wsz, 420)) + 1
        D = int(rng.choice([int(rng.integers(2, 513)), 16, 32, 33, 64, 65, 128, 129, 256, 257, 384, 385, 512]))
        P1 = int(rng.integers(0, 40))
        P2 = int(rng.integers(P1, 256))
        out.append((H, W, D, wsz, P1, P2, int(rng.choice([4, 8]))))
    return out


@pytest.mark.parametrize("case", _random_configs(48, 20261110), ids=lambda c: "x".join(map(str, c)))
def test_seeded_random_configurations(oracle, case):
    """Sizes, disparity ranges, windows and penalties drawn from a seeded generator (layout boundaries of the packed
    lanes included): final disparity of the default formulation against the oracle, and the materialised
    formulation on the same input."""
    H, W, D, wsz, P1, P2, paths = case
    left, right = synth_pair(H * 977 + W * 13 + D, H, W, max(1, min(D, W // 2)))
    want = oracle.sgm_census(left, right, D, wsz, P1, P2, paths)
    for flags in (0, 1):
        e = make(disp_range=D, window_sz=wsz, penalty1=P1, penalty2=P2, num_paths=paths, flags=flags)
        assert np.array_equal(e.compute(left, right), want), "flags=%d" % flags
        e.close()


@pytest.mark.parametrize("case", [(16, 156, 16, 5, 2, 4, 4), (17, 105, 16, 7, 6, 11, 4), (36, 327, 64, 3, 7, 26, 4),
                                  (39, 50, 31, 1, 8, 23, 4), (42, 276, 32, 7, 20, 26, 4), (11, 214, 65, 7, 8, 28, 4),
                                  (20, 90, 128, 5, 3, 31, 4), (20, 90, 128, 5, 3, 31, 8), (20, 90, 200, 5, 3, 15, 4)],
                         ids=lambda c: "x".join(map(str, c)))
def test_four_paths_with_small_penalties(oracle, case):
    """P2 <= 31 lets the final kernel add the G bytes of 8 volumes at a time without carry; with 4 paths only 4
    volumes exist (found by exp/random_stress.py: the sum used to take in the registers of the 4 that were never
    loaded).  Three pairs per launch, final disparity against the oracle.  stereo_sgm.h:105-117."""
    H, W, D, wsz, P1, P2, paths = case
    pairs = [synth_pair(H * 977 + W * 13 + D + j, H, W, max(1, min(D, W // 2))) for j in range(3)]
    e = make(disp_range=D, window_sz=wsz, penalty1=P1, penalty2=P2, num_paths=paths, max_batch=3)
    got = e.compute_batch(np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs]))
    e.close()
    for j, (left, right) in enumerate(pairs):
        assert np.array_equal(got[j], oracle.sgm_census(left, right, D, wsz, P1, P2, paths)), "pair %d" % j


@pytest.mark.parametrize("D", [16, 40, 100, 128, 130, 256, 270, 384])
def test_horizontal_sweep_block_boundaries(oracle, D):
    """The horizontal sweeps keep their census windows in registers and shift them by one word per step, in blocks
    of 8 steps (12 with 257..384 disparities) that need two blocks + 1 pixels of path ahead of them (aggregate3_kernel, HZ): cropped widths around
    every block boundary, disparity ranges of every lane layout with and without pad disparities, three pairs per
    launch so that the 16-lane rows of a wave walk different image rows.  stereo_sgm.cc:191-268."""
    for w in (15, 16, 17, 18, 19, 24, 25, 26, 27, 33, 34, 35, 37, 38, 41, 42, 49, 50):
        H, W = 11, w + 4
        pairs = [synth_pair(w * 31 + D + k, H, W, max(1, min(D, W // 2))) for k in range(3)]
        e = make(disp_range=D, window_sz=5, penalty1=3, penalty2=60, max_batch=3)
        got = e.compute_batch(np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs]))
        for k, (left, right) in enumerate(pairs):
            assert np.array_equal(got[k], oracle.sgm_census(left, right, D, 5, 3, 60, 8)), "w=%d pair %d" % (w, k)
        e.close()


@pytest.mark.parametrize("shape", [(40, 3840), (2160, 64)], ids=["3840x40", "64x2160"])
def test_4k_geometry_strips_vs_oracle(oracle, shape):
    """BASELINE.json configs[4]'s geometry (3840x2160, D=256, 8 paths) in strips the oracle finishes in seconds: a
    3840-wide strip (horizontal paths of 3 836 steps: the shift-register blocks, the launch split; diagonals that wrap
    after 36 rows) and a 2160-tall strip (vertical paths of 2 156 steps, diagonals that wrap every 60 rows), at RP = 8,
    three pairs per launch.  Final disparity against the oracle.  stereo_sgm.cc:120-135,191-268."""
    H, W = shape
    D = 256
    pairs = [synth_pair(4000 + H + k, H, W, min(240, W // 2)) for k in range(3)]
    e = make(disp_range=D, window_sz=5, penalty1=3, penalty2=60, max_batch=3)
    got = e.compute_batch(np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs]))
    e.close()
    for k, (left, right) in enumerate(pairs):
        assert np.array_equal(got[k], oracle.sgm_census(left, right, D, 5, 3, 60, 8)), "pair %d" % k


@pytest.mark.parametrize("D", [16, 128, 256, 270])
def test_wrapped_diagonal_boundaries(oracle, D):
    """The diagonal sweeps walk wrapped diagonals (aggregate3_kernel, WRAP): a 16-lane row continues, after the image
    column its diagonal ends in, with the diagonal that starts one row further, in steady-state blocks that stop short
    of the wrap.  Cropped widths and heights around the block sizes, wider than tall, square, and taller than wide
    (several wraps per row), three pairs per launch.  stereo_sgm.cc:197-267 (start sets), stereo_sgm.h:119-147."""
    for w, h in ((9, 9), (10, 9), (9, 10), (13, 30), (16, 33), (17, 17), (18, 70), (23, 8), (31, 12), (40, 41), (3, 20), (2, 9), (1, 7)):
        H, W = h + 4, w + 4
        pairs = [synth_pair(w * 37 + h * 5 + D + k, H, W, max(1, min(D, W // 2))) for k in range(3)]
        want = [oracle.sgm_census(left, right, D, 5, 3, 60, 8) for left, right in pairs]
        for flags in (0, 1):  # the volume-free kernels (WRAP) and the materialised ones (WrapWalk)
            e = make(disp_range=D, window_sz=5, penalty1=3, penalty2=60, max_batch=3, flags=flags)
            got = e.compute_batch(np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs]))
            for k in range(3):
                assert np.array_equal(got[k], want[k]), "w=%d h=%d pair %d flags %d" % (w, h, k, flags)
            e.close()


@pytest.mark.parametrize("case", [(9, 9, 16), (30, 13, 64), (70, 18, 40), (41, 40, 128), (20, 3, 8), (130, 37, 100), (64, 200, 130)],
                         ids=lambda c: "x".join(map(str, c)))
def test_float_volume_wrapped_diagonals(oracle, case):
    """Float volumes taller than wide, square and wider than tall: the diagonal sweeps of the one-path-per-wave kernels
    walk wrapped diagonals with several wraps per path (WrapWalk, sgmx_kernels.hpp).  Summed costs bit-exact against
    the restatement of sgm.cc:103-118,177-254, disparities apart from the reference's undefined pixels."""
    H, W, D = case
    C = _float_volume(H * 7 + W + D, H, W, D, True)
    e = make(sv.SGM, disp_range=D, penalty1=3, penalty2=60, num_paths=8)
    disp = e.compute_cost(C)
    want = oracle.sgm_cost_f32(C, 3, 60, 8, taps=True)
    S = e.stages(H, W)["S"]
    finite = np.isfinite(want["S"])
    assert np.array_equal(np.isfinite(S), finite)
    assert np.array_equal(S[finite], want["S"][finite])
    assert int(np.sum(disp != want["disp"])) <= 9 * want["n_undefined"]
    e.close()


@pytest.mark.parametrize("case", _random_configs(64, 777), ids=lambda c: "x".join(map(str, c)))
def test_seeded_random_configurations_three_pairs(oracle, case):
    """A second seeded draw (the one exp/random_stress.py starts with), three different pairs per launch: the four
    rows of a wave then walk paths of different pairs."""
    H, W, D, wsz, P1, P2, paths = case
    pairs = [synth_pair(H * 977 + W * 13 + D + j, H, W, max(1, min(D, W // 2))) for j in range(3)]
    e = make(disp_range=D, window_sz=wsz, penalty1=P1, penalty2=P2, num_paths=paths, max_batch=3)
    got = e.compute_batch(np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs]))
    e.close()
    for j, (left, right) in enumerate(pairs):
        assert np.array_equal(got[j], oracle.sgm_census(left, right, D, wsz, P1, P2, paths)), "pair %d" % j


@pytest.mark.parametrize("flags", [0, 1], ids=["fast", "materialized"])
@pytest.mark.parametrize("shape", [(5, 5), (6, 40), (40, 6), (5, 300), (6, 6), (7, 5), (5, 7)])
def test_degenerate_shapes(oracle, shape, flags):
    """Cropped images of 1 x 1, one row, one column, 2 x 2; D larger than the width."""
    H, W = shape
    rng = np.random.default_rng(H * 7 + W)
    left = rng.integers(0, 256, size=(H, W)).astype(np.uint8)
    right = rng.integers(0, 256, size=(H, W)).astype(np.uint8)
    e = make(disp_range=64, window_sz=5, flags=flags)
    assert np.array_equal(e.compute(left, right), oracle.sgm_census(left, right, 64, 5, 3, 60, 8))


def test_strided_input_and_reuse(oracle):
    """Row pitch != width (cv::Mat ROI style), and one context reused across sizes."""
    e = make(disp_range=32, window_sz=3)
    for seed, (H, W) in enumerate([(30, 64), (18, 100), (30, 64)]):
        l, r = synth_pair(seed, H, W + 9, 12)
        lv, rv = l[:, 4:4 + W], r[:, 4:4 + W]
        want = oracle.sgm_census(np.ascontiguousarray(lv), np.ascontiguousarray(rv), 32, 3, 3, 60, 8)
        assert np.array_equal(e.compute(lv, rv), want)
        # and a disparity image with a row pitch of its own (cv::Mat ROI as the output)
        out = np.full((H, W + 7), 0xABCD, np.uint16)
        e.check(e.lib.sgmx_compute_census_u8(e.h, lv.ctypes.data, rv.ctypes.data, W, H, lv.strides[0], rv.strides[0],
                                             out.ctypes.data, out.strides[0]))
        assert np.array_equal(out[:, :W], want) and (out[:, W:] == 0xABCD).all()


# ---------------------------------------------------------------- golden fixtures
def _golden_keys():
    from test_oracle_golden import KNOWN
    return sorted(KNOWN)


@pytest.mark.parametrize("flags", [0, 1], ids=["fast", "materialized"])
@pytest.mark.parametrize("key", _golden_keys(), ids=lambda k: "%s-D%d-w%d-p%d" % k)
def test_reference_known_answers(oracle, tsukuba, kitti, key, flags):
    """All six known answers SURVEY.md 8c records for recon::StereoSGM::compute on the reference's own fixtures
    (Tsukuba D=16 wsz 5 / 3, 8 / 4 paths = BASELINE config 1; KITTI 1241x376 D=128 wsz 5 / 3, 8 / 4 paths = BASELINE
    config 2 at full size), both formulations; the two "per-direction reset" hashes are what the output must NOT be."""
    from test_oracle_golden import KNOWN, RESET_FNV
    name, D, wsz, paths = key
    left, right = tsukuba if name == "tsukuba" else kitti
    e = make(disp_range=D, window_sz=wsz, num_paths=paths, flags=flags)
    disp = e.compute(left, right)
    e.close()
    s, nz, fnv = KNOWN[key]
    assert int(disp.sum(dtype=np.uint64)) == s and int(np.count_nonzero(disp)) == nz
    assert oracle.fnv_u16(disp) == fnv
    if key in RESET_FNV:
        assert oracle.fnv_u16(disp) != RESET_FNV[key]


@pytest.mark.parametrize("flags", [0, 1], ids=["fast", "materialized"])
def test_kitti_stages_vs_oracle(oracle, kitti, flags):
    left, right = kitti
    e = make(disp_range=128, window_sz=5, flags=flags)
    disp = e.compute(left, right)
    want = oracle.sgm_census(left, right, 128, 5, 3, 60, 8, taps=True)
    st = e.stages(*left.shape)
    assert np.array_equal(st["S"].astype(np.uint32), want["S"])
    assert np.array_equal(disp, want["disp"])


# ---------------------------------------------------------------- batch / device entry points
def test_batch_chunks_equal_single(oracle):
    n, H, W = 5, 34, 120
    pairs = [synth_pair(100 + i, H, W, 40) for i in range(n)]
    L = np.stack([p[0] for p in pairs])
    R = np.stack([p[1] for p in pairs])
    e = make(disp_range=64, window_sz=5, max_batch=2)  # 5 pairs -> groups of 2, 2, 1
    got = e.compute_batch(L, R)
    for i in range(n):
        assert np.array_equal(got[i], oracle.sgm_census(L[i], R[i], 64, 5, 3, 60, 8)), i


def test_deprecated_flag_is_accepted(oracle):
    """SGMX_FLAG_NO_SMALL_GROUPS (2) of the round-1 header is a no-op today; contexts created with it still work."""
    left, right = synth_pair(5, 30, 90, 20)
    e = make(disp_range=32, flags=2)
    assert np.array_equal(e.compute(left, right), oracle.sgm_census(left, right, 32, 5, 3, 60, 8))
    e.close()
    with pytest.raises(sv.SgmxError):
        make(disp_range=32, flags=4)


def test_profiling_modes(oracle):
    """sgmx_set_profiling: 1 brackets every launch (all classes report), 2 only the direction sweeps with shared boundary
    events (per-direction times, the other classes report nothing); the results do not change."""
    pairs = [synth_pair(300 + i, 40, 160, 40) for i in range(3)]
    L, R = np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs])
    e = make(disp_range=64, window_sz=5, max_batch=3)
    want = e.compute_batch(L, R)
    for mode, classes in ((1, {"census", "aggregate", "wta", "median"}), (2, {"aggregate"})):
        e.set_profiling(mode)
        for _ in range(2):
            assert np.array_equal(e.compute_batch(L, R), want)
        dirs = e.direction_stats()
        stats = e.kernel_stats()
        assert {k for k, v in stats.items() if v[1]} == classes
        assert stats["aggregate"][1] == 16 and all(n == 2 and ms > 0 for ms, n in dirs)
        assert abs(sum(ms for ms, _ in dirs) - stats["aggregate"][0]) < 1e-6
    e.set_profiling(0)
    e.compute_batch(L, R)
    assert all(v[1] == 0 for v in e.kernel_stats().values())
    with pytest.raises(sv.SgmxError):
        e.set_profiling(3)
    e.close()


@pytest.mark.parametrize("D,paths", [(64, 8), (16, 8), (130, 8), (64, 4)])
def test_directions_side_by_side_equal_one_after_the_other(oracle, D, paths):
    """Small launch groups run the diagonal directions 0 and 5 side by side with the horizontal directions 1 and 6 that
    follow them (Agg2Args::fuse: analytic start values L_{k-1} + 2 C on the start column, a second border ring, a second
    stream); under profiling the directions run one after the other.  Same disparities and the same summed costs, for
    1..5 pairs per call, on the caller's null stream and on a stream of its own; against the oracle."""
    import torch
    H, W = 52, 170
    pairs = [synth_pair(400 + i, H, W, 40) for i in range(5)]
    want = [oracle.sgm_census(l, r, D, 5, 3, 60, paths, taps=True) for l, r in pairs]
    side = torch.cuda.Stream()
    for n in (1, 2, 5):
        L = torch.from_numpy(np.stack([p[0] for p in pairs[:n]])).cuda()
        R = torch.from_numpy(np.stack([p[1] for p in pairs[:n]])).cuda()
        e = make(disp_range=D, window_sz=5, penalty1=3, penalty2=60, num_paths=paths, max_batch=n)
        for mode, stream in ((0, 0), (1, 0), (0, side.cuda_stream)):  # side by side / one after the other / own stream
            e.set_profiling(mode)
            out = torch.full((n, H, W), -1, dtype=torch.int16, device="cuda")
            torch.cuda.synchronize()
            for _ in range(2):
                e.compute_device(L.data_ptr(), R.data_ptr(), out.data_ptr(), n, W, H, stream)
            torch.cuda.synchronize()
            got = out.cpu().numpy().view(np.uint16)
            for k in range(n):
                assert np.array_equal(got[k], want[k]["disp"]), "n=%d mode=%d pair %d" % (n, mode, k)
            assert np.array_equal(e.stages(H, W, n - 1)["S"], want[n - 1]["S"])
            e.kernel_stats()
        e.close()


def test_taps_address_the_last_launch_group(oracle):
    """The stage taps reach the last launch group only; sgmx_last_launch_group says which pairs of the call those
    are.  5 pairs with max_batch 2 run as 2 + 2 + 1: tap index 0 is pair 4 of the call."""
    n, H, W = 5, 34, 120
    pairs = [synth_pair(100 + i, H, W, 40) for i in range(n)]
    e = make(disp_range=64, window_sz=5, max_batch=2)
    e.compute_batch(np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs]))
    assert e.last_launch_group() == (4, 1)
    want = oracle.sgm_census(pairs[4][0], pairs[4][1], 64, 5, 3, 60, 8, taps=True)
    assert np.array_equal(e.stages(H, W, 0)["S"], want["S"])
    with pytest.raises(sv.SgmxError):
        e.stages(H, W, 1)
    e.compute_batch(np.stack([p[0] for p in pairs[:4]]), np.stack([p[1] for p in pairs[:4]]))
    assert e.last_launch_group() == (2, 2)
    want = oracle.sgm_census(pairs[3][0], pairs[3][1], 64, 5, 3, 60, 8, taps=True)
    assert np.array_equal(e.stages(H, W, 1)["S"], want["S"])
    e.close()


def test_adaptive_groups_mix_formulations(oracle):
    """max_batch 5 over 9 pairs: a group of 5 then one of 4, then a single-pair call on the same
    context, and the stage taps of that call: all bit-exact."""
    n, H, W = 9, 30, 140
    pairs = [synth_pair(300 + i, H, W, 50) for i in range(n)]
    e = make(disp_range=100, window_sz=5, max_batch=5)
    got = e.compute_batch(np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs]))
    for i in range(n):
        assert np.array_equal(got[i], oracle.sgm_census(pairs[i][0], pairs[i][1], 100, 5, 3, 60, 8)), i
    one = e.compute(*pairs[4])
    assert np.array_equal(one, got[4])
    st = e.stages(H, W)
    want = oracle.sgm_census(pairs[4][0], pairs[4][1], 100, 5, 3, 60, 8, taps=True)
    assert np.array_equal(st["C"], want["C"]) and np.array_equal(st["S"].astype(np.uint32), want["S"])


def test_device_entry_point_with_torch_memory(oracle):
    import torch
    n, H, W = 3, 40, 96
    pairs = [synth_pair(200 + i, H, W, 30) for i in range(n)]
    L = torch.from_numpy(np.stack([p[0] for p in pairs])).cuda()
    R = torch.from_numpy(np.stack([p[1] for p in pairs])).cuda()
    out = torch.zeros((n, H, W), dtype=torch.int16, device="cuda")
    e = make(disp_range=128, window_sz=5, max_batch=4)
    e.compute_device(L.data_ptr(), R.data_ptr(), out.data_ptr(), n, W, H, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint16)
    for i in range(n):
        assert np.array_equal(got[i], oracle.sgm_census(pairs[i][0], pairs[i][1], 128, 5, 3, 60, 8)), i


def test_idempotent_and_deterministic(kitti):
    e = make(disp_range=128, window_sz=5)
    a = e.compute(*kitti)
    b = e.compute(*kitti)
    assert np.array_equal(a, b)


def test_error_paths():
    e = make(disp_range=16, window_sz=5)
    with pytest.raises(sv.SgmxError):
        e.compute(np.zeros((3, 3), np.uint8), np.zeros((3, 3), np.uint8))  # smaller than the window
    with pytest.raises(ValueError):
        e.compute(np.zeros((8, 8), np.uint8), np.zeros((8, 9), np.uint8))
    with pytest.raises(sv.SgmxError):
        e.stages(8, 8, index=5)


# ---------------------------------------------------------------- float data-cost path
def _float_volume(seed, H, W, D, ties=False):
    rng = np.random.default_rng(seed)
    C = (rng.random((H, W, D), dtype=np.float32) * 4).astype(np.float32)
    if ties:
        C = np.round(C * 4) / 4  # exact ties
    xs = np.arange(W)[None, :, None]
    ds = np.arange(D)[None, None, :]
    C = np.where(ds > xs, np.float32(np.finfo(np.float32).max), C)  # sgm.cc:55
    return np.ascontiguousarray(C, np.float32)


@pytest.mark.parametrize("case", [(24, 70, 16, 8, False), (20, 150, 128, 8, False), (20, 150, 128, 8, True),
                                  (17, 140, 100, 4, False), (14, 90, 255, 8, False), (11, 60, 300, 8, True),
                                  (9, 640, 300, 8, False)],  # wider than D: winners past d = 255 (low 16 bits of 256 d)
                         ids=lambda c: "x".join(map(str, c)))
def test_float_cost_path(oracle, case):
    H, W, D, paths, ties = case
    C = _float_volume(H + W + D, H, W, D, ties)
    e = make(sv.SGM, disp_range=D, penalty1=3, penalty2=60, num_paths=paths)
    disp = e.compute_cost(C)
    want = oracle.sgm_cost_f32(C, 3, 60, paths, taps=True)
    S = e.stages(H, W)["S"]
    finite = np.isfinite(want["S"])
    assert np.array_equal(np.isfinite(S), finite)
    rel = np.abs(S[finite] - want["S"][finite]) / np.maximum(np.abs(want["S"][finite]), 1e-30)
    assert rel.max() <= 1e-5  # north_star tolerance
    assert np.array_equal(S[finite], want["S"][finite])  # and in fact bit-exact
    same = np.mean(disp == want["disp"])
    assert same >= 0.999
    # pixels whose subpixel value is undefined in the reference (NaN cast) are the only allowed difference
    assert int(np.sum(disp != want["disp"])) <= 9 * want["n_undefined"]


def _random_float_configs(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        H = int(rng.integers(1, 40))
        W = int(rng.integers(2, 380))
        D = int(rng.choice([int(rng.integers(2, 513)), 64, 65, 128, 129, 256, 257, 512]))
        P1 = int(rng.integers(0, 20))
        out.append((H, W, D, P1, int(rng.integers(P1, 200)), int(rng.choice([4, 8])), bool(rng.integers(0, 2))))
    return out


@pytest.mark.parametrize("case", _random_float_configs(24, 77), ids=lambda c: "x".join(map(str, c)))
def test_seeded_random_float_volumes(oracle, case):
    """Float-volume path on seeded random sizes / ranges / penalties (with and without exact ties): aggregated
    costs within north_star's 1e-5 (bit-exact in practice), disparities identical apart from the reference's
    undefined pixels."""
    H, W, D, P1, P2, paths, ties = case
    C = _float_volume(H * 131 + W + D, H, W, D, ties)
    e = make(sv.SGM, disp_range=D, penalty1=P1, penalty2=P2, num_paths=paths)
    disp = e.compute_cost(C)
    want = oracle.sgm_cost_f32(C, P1, P2, paths, taps=True)
    S = e.stages(H, W)["S"]
    finite = np.isfinite(want["S"])
    assert np.array_equal(np.isfinite(S), finite)
    assert np.array_equal(S[finite], want["S"][finite])
    assert int(np.sum(disp != want["disp"])) <= 9 * want["n_undefined"]
    e.close()


def test_seeded_random_batches(oracle):
    """Launch groups of random pair counts (1 .. 11, contexts of random max_batch) through the host batch entry point:
    every pair equals its own oracle result, whatever its place in a group."""
    rng = np.random.default_rng(99)
    for trial in range(6):
        H, W = int(rng.integers(8, 40)), int(rng.integers(20, 200))
        D = int(rng.choice([16, 48, 128, 200]))
        n, mb = int(rng.integers(1, 12)), int(rng.integers(1, 9))
        pairs = [synth_pair(1000 * trial + i, H, W, max(1, min(D, W // 2))) for i in range(n)]
        e = make(disp_range=D, window_sz=5, penalty1=3, penalty2=60, max_batch=mb)
        got = e.compute_batch(np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs]))
        for i, (l, r) in enumerate(pairs):
            assert np.array_equal(got[i], oracle.sgm_census(l, r, D, 5, 3, 60, 8)), (trial, i, n, mb)
        e.close()


# ---------------------------------------------------------------- caller-side pre-filter
@pytest.mark.parametrize("shape", [(376, 1241), (37, 53), (1, 9), (9, 1), (1, 1), (2, 300)])
@pytest.mark.parametrize("sigma", [0.7, 0.0, 1.5])
def test_gaussian_blur3(oracle, shape, sigma):
    """cv::GaussianBlur(img, img, Size(3,3), sigma) of dense_stereo/main.cc:83-85 (OpenCV arithmetic:
    parity unpinned by the reference, bit-exact against the oracle's restatement)."""
    rng = np.random.default_rng(shape[0] * 31 + shape[1])
    img = rng.integers(0, 256, size=shape).astype(np.uint8)
    e = make(disp_range=16, window_sz=3)
    assert np.array_equal(e.gaussian_blur3(img, sigma), oracle.gaussian3(img, sigma))
    wide = rng.integers(0, 256, size=(shape[0], shape[1] + 5)).astype(np.uint8)
    view = wide[:, 2:2 + shape[1]]  # row pitch != width
    assert np.array_equal(e.gaussian_blur3(view, sigma), oracle.gaussian3(np.ascontiguousarray(view), sigma))


def test_depth_from_disparity(oracle):
    """CalculateDepthImage of create_rgbd_dataset/main.cc:12-27 (KITTI-like focal length and baseline)."""
    rng = np.random.default_rng(3)
    disp = rng.integers(0, 120 * 256, size=(37, 91)).astype(np.uint16)
    disp[disp < 3 * 256] = 0
    disp[0, :4] = [255, 256, 257, 768]  # d < 1 stays 0; d = 1 is the first converted value
    e = make(disp_range=16, window_sz=3)
    want = oracle.depth_image(disp, 721.5, 0.54)
    assert want is not None and np.array_equal(e.depth_image(disp, 721.5, 0.54), want)
    assert want[0, 0] == 0 and want[0, 1] == round(100 * 721.5 * 0.54)
    assert oracle.depth_image(disp, 7215.0, 0.54) is None
    with pytest.raises(sv.SgmxError):
        e.depth_image(disp, 7215.0, 0.54)  # 100 f b / d >= 65536: the reference throws


def test_float_path_takes_penalties_beyond_the_uint16_guard(oracle):
    """The uint16 path-cost guard (P2 <= 7044) belongs to the census entry points only: recon::SGM / SGMStereo keep
    float costs and take any penalty (sgm.h:129-157), while a census call on such a context is SGMX_ERR_RANGE."""
    rng = np.random.default_rng(5)
    h, w, D = 18, 70, 32
    C = (rng.random((h, w, D), dtype=np.float32) * 4).astype(np.float32)
    C[:, np.arange(D)[None, :] > np.arange(w)[:, None]] = np.finfo(np.float32).max
    e = sv.SGM(sv.SGMParams(disp_range=D, penalty1=100, penalty2=20000))
    got = e.compute_cost(C)
    want = oracle.sgm_cost_f32(C, 100, 20000, 8, taps=True)
    assert int((got != want["disp"]).sum()) <= 9 * want["n_undefined"]
    e.close()
    c = sv.StereoSGM(sv.StereoSGMParams(disp_range=D, window_sz=5, penalty1=100, penalty2=20000))
    with pytest.raises(sv.SgmxError) as ei:
        c.compute(np.zeros((20, 40), np.uint8), np.zeros((20, 40), np.uint8))
    assert ei.value.code == 2
    c.close()


def test_wide_image_beyond_the_wta_lds_tables_is_invalid_arg():
    """check_size prices the WTA kernel's LDS with the formula the launch uses: D = 512 (16 packed pairs per lane:
    16 KiB tile) leaves room for 9 216 columns, not the 9 728 of narrow ranges."""
    e = sv.StereoSGM(sv.StereoSGMParams(disp_range=512, window_sz=1, penalty1=3, penalty2=60))
    img = np.zeros((2, 9300), np.uint8)
    with pytest.raises(sv.SgmxError) as ei:
        e.compute(img, img)
    assert ei.value.code == 1 and "LDS" in str(ei.value)
    ok = np.zeros((2, 9200), np.uint8)
    assert e.compute(ok, ok).shape == (2, 9200)
    e.close()


def test_dev_calls_on_two_streams_share_one_context(oracle):
    """Two _dev calls on different non-blocking streams through one context: the second waits for the first's
    launch group (shared scratch) instead of overwriting its volumes."""
    import torch
    H, W = 60, 200
    a = synth_pair(901, H, W, 40)
    b = synth_pair(902, H, W, 40)
    e = make(disp_range=64, window_sz=5, max_batch=1)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    ins = [torch.from_numpy(np.stack(x)).cuda() for x in (a, b)]
    outs = [torch.zeros((H, W), dtype=torch.int16, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    for _ in range(3):
        e.compute_device(ins[0][0].data_ptr(), ins[0][1].data_ptr(), outs[0].data_ptr(), 1, W, H, s1.cuda_stream)
        e.compute_device(ins[1][0].data_ptr(), ins[1][1].data_ptr(), outs[1].data_ptr(), 1, W, H, s2.cuda_stream)
    torch.cuda.synchronize()
    for (l, r), o in zip((a, b), outs):
        assert np.array_equal(o.cpu().numpy().view(np.uint16), oracle.sgm_census(l, r, 64, 5, 3, 60, 8))


@pytest.mark.parametrize("ncodes,ht", [(4, 10), (8, 24), (16, 52), (5, 14)])
def test_hamming_allocating_and_threaded_forms(oracle, ncodes, ht):
    """match_hamming_thres, crossmatch_hamming, compute_hamming_thread, match_hamming_thres_nt (core/hamming.h:84-96,
    108-112) through the C ABI against the restatement and, when built, against the reference's own core/hamming.c."""
    rng = np.random.default_rng(ncodes * 7 + ht)
    a = rng.integers(0, 256, size=(300, ncodes)).astype(np.uint8)
    b = rng.integers(0, 256, size=(270, ncodes)).astype(np.uint8)
    e = make(disp_range=16)
    for cross in (False, True):
        want = oracle.match_hamming(a, b, ht, cross)
        got = e.crossmatch_hamming_alloc(a, ht, bufsize=3) if cross else e.match_hamming_thres_alloc(a, b, ht, bufsize=3)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        ref = oracle.ref_match_hamming_alloc(a, b, ht, cross, bufsize=3)
        if ref is not None:
            assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    assert np.array_equal(e.hamming_matrix_thread(a, b), oracle.hamming_matrix(a, b))
    k, h = e.match_hamming_thres_nt(a, b, ht)
    wk, wh = oracle.match_hamming_blocked(a, b, ht)
    assert len(h) > 0 and np.array_equal(k, wk) and np.array_equal(h, wh)
    ref = oracle.ref_match_hamming_thres_nt(a, b, ht)
    if ref is not None:
        assert np.array_equal(k, ref[0]) and np.array_equal(h, ref[1])
    # no match at all, and the shapes whose block arithmetic leaves the sets in the reference
    k0, h0 = e.match_hamming_thres_nt(a[:5], 255 - a[:5], 0)
    assert k0.shape == (0, 2) and h0.shape == (0,)
    with pytest.raises(sv.SgmxError):
        e.match_hamming_thres_nt(a, b[:100], ht)
    e.close()


def test_hamming_compat_library_is_a_drop_in_for_core_hamming(oracle):
    """libhamming_sgmx.so (include/sgmx_hamming_compat.h) exports core/hamming.h's own names and signatures; the same
    ctypes prototypes bound to it and to the reference's core/hamming.c build give the same results."""
    import ctypes as C
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ref_path = os.path.join(root, "oracle", "_ref", "libhamming_ref.so")
    if not os.path.exists(ref_path):
        pytest.skip("oracle/_ref/libhamming_ref.so not built (reference tree not mounted)")
    u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")

    def bind(path):
        L = C.CDLL(path)
        L.hamming.argtypes, L.hamming.restype = [u8p, u8p, C.c_int], C.c_uint16
        L.compute_hamming.argtypes, L.compute_hamming.restype = [C.c_void_p, u8p, u8p, C.c_int, C.c_int, C.c_int], None
        L.match_hamming_count.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)]
        L.match_hamming_count.restype = None
        L.match_hamming_thres_prealloc.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.match_hamming_thres_prealloc.restype = C.c_size_t
        L.crossmatch_hamming_count.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)]
        L.crossmatch_hamming_count.restype = None
        L.crossmatch_hamming_prealloc.argtypes = [u8p, C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.crossmatch_hamming_prealloc.restype = C.c_size_t
        return L

    def drive(L, a, b, ht):
        out = [int(L.hamming(a[3:4].copy(), b[5:6].copy(), a.shape[1]))]
        dis = np.empty((b.shape[0], a.shape[0]), np.uint16)
        L.compute_hamming(dis.ctypes.data, a, b, a.shape[0], b.shape[0], a.shape[1])
        n = C.c_size_t(0)
        L.match_hamming_count(a, b, a.shape[0], b.shape[0], ht, a.shape[1], C.byref(n))
        idx, hams = np.zeros((n.value, 2), np.int32), np.zeros(n.value, np.uint16)
        assert L.match_hamming_thres_prealloc(a, b, a.shape[0], b.shape[0], ht, a.shape[1], idx.ctypes.data, hams.ctypes.data) == n.value
        m = C.c_size_t(0)
        L.crossmatch_hamming_count(a, a.shape[0], ht, a.shape[1], C.byref(m))
        cidx, chams = np.zeros((m.value, 2), np.int32), np.zeros(m.value, np.uint16)
        assert L.crossmatch_hamming_prealloc(a, a.shape[0], ht, a.shape[1], cidx.ctypes.data, chams.ctypes.data) == m.value
        return out + [dis, idx, hams, cidx, chams]

    rng = np.random.default_rng(77)
    a = rng.integers(0, 256, size=(90, 8)).astype(np.uint8)
    b = rng.integers(0, 256, size=(70, 8)).astype(np.uint8)
    mine = drive(bind(os.path.join(root, "stereo_vision_amd", "libhamming_sgmx.so")), a, b, 26)
    ref = drive(bind(ref_path), a, b, 26)
    assert mine[0] == ref[0] and len(mine[3]) > 0 and len(mine[5]) > 0
    for x, y in zip(mine[1:], ref[1:]):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("shape", [(1, 1), (7, 3), (8, 33), (9, 32), (2047, 31), (2048, 5), (2049, 64), (4100, 65), (6151, 40)],
                         ids=lambda s: "%dx%d" % s)
@pytest.mark.parametrize("ncodes", [4, 8, 16, 32, 12])
def test_hamming_matrix_tile_boundaries(oracle, shape, ncodes):
    """compute_hamming (core/hamming.c:167-236) through the register-tiled kernel (8 signatures a_i per thread, 2048
    per workgroup, 32 rows b_j per workgroup, 16-byte stores that are only 2-byte aligned when na is odd): set sizes
    around every tile edge, against the restatement and the reference's own core/hamming.c.  ncodes = 12 runs the
    generic kernel."""
    na, nb = shape
    rng = np.random.default_rng(na * 131 + nb + ncodes)
    a = rng.integers(0, 256, size=(na, ncodes)).astype(np.uint8)
    b = rng.integers(0, 256, size=(nb, ncodes)).astype(np.uint8)
    e = make()
    got = e.hamming_matrix(a, b)
    assert np.array_equal(got, oracle.hamming_matrix(a, b))
    ref = oracle.ref_hamming_matrix(a, b)
    if ref is not None:
        assert np.array_equal(got, ref)
    # the context's buffers are kept from call to call: a smaller and a larger call on the same context
    a2 = rng.integers(0, 256, size=(max(1, na // 3), ncodes)).astype(np.uint8)
    assert np.array_equal(e.hamming_matrix(a2, b), oracle.hamming_matrix(a2, b))
    a3 = rng.integers(0, 256, size=(na + 77, ncodes)).astype(np.uint8)
    assert np.array_equal(e.hamming_matrix(a3, b), oracle.hamming_matrix(a3, b))
    e.close()


@pytest.mark.parametrize("ncodes,ht", [(4, 9), (8, 22), (16, 50), (32, 110)])
@pytest.mark.parametrize("shape", [(1, 1), (7, 64), (8, 65), (9, 63), (33, 130), (257, 500), (4097, 40), (6200, 33)],
                         ids=lambda s: "%dx%d" % s)
def test_hamming_matchers_query_tiles(oracle, shape, ncodes, ht):
    """The matchers' multi-query kernel (a wave owns 8 consecutive queries, 32 per workgroup): query counts around the
    tile edges, base sets around the 64-lane chunks, more than 4096 queries (the prefix sum over the per-query counts
    then runs as three launches), plain and cross matching, against the restatement and the reference's own
    core/hamming.c (core/hamming.c:300-322, 385-407, 725-765, 810-839)."""
    n1, n2 = shape
    rng = np.random.default_rng(n1 * 7 + n2 + ncodes)
    a = rng.integers(0, 256, size=(n1, ncodes)).astype(np.uint8)
    b = rng.integers(0, 256, size=(n2, ncodes)).astype(np.uint8)
    if n1 > 2 and n2 > 2:
        b[n2 // 2] = a[n1 // 3]
        a[n1 - 1] = a[1]
    e = make()
    for cross in (False, True):
        want = oracle.match_hamming(a, b, ht, cross)
        n = e.crossmatch_hamming_count(a, ht) if cross else e.match_hamming_count(a, b, ht)
        idx, hams = e.crossmatch_hamming(a, ht) if cross else e.match_hamming_thres(a, b, ht)
        assert n == len(want[1])
        assert np.array_equal(idx, want[0]) and np.array_equal(hams, want[1])
        ref = oracle.ref_match_hamming(a, b, ht, cross)
        if ref is not None:
            assert np.array_equal(idx, ref[0]) and np.array_equal(hams, ref[1])
    e.close()


def test_hamming_compat_library_from_several_threads(oracle):
    """The reference's functions are re-entrant; the drop-in library serialises its calls on one shared context
    (include/sgmx_hamming_compat.h).  Eight threads call compute_hamming and match_hamming_count on their own inputs
    at the same time: every result equals the restatement's."""
    import ctypes as C
    import os
    from concurrent.futures import ThreadPoolExecutor
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
    L = C.CDLL(os.path.join(root, "stereo_vision_amd", "libhamming_sgmx.so"))
    L.hamming.argtypes, L.hamming.restype = [u8p, u8p, C.c_int], C.c_uint16
    L.compute_hamming.argtypes, L.compute_hamming.restype = [C.c_void_p, u8p, u8p, C.c_int, C.c_int, C.c_int], None
    L.match_hamming_count.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)]
    L.match_hamming_count.restype = None

    def work(t):
        rng = np.random.default_rng(900 + t)
        ok = True
        for rep in range(6):
            na, nb, nc = int(rng.integers(1, 900)), int(rng.integers(1, 300)), int(rng.choice([4, 8, 16, 5]))
            a = rng.integers(0, 256, size=(na, nc)).astype(np.uint8)
            b = rng.integers(0, 256, size=(nb, nc)).astype(np.uint8)
            dis = np.empty((nb, na), np.uint16)
            L.compute_hamming(dis.ctypes.data, a, b, na, nb, nc)
            want = oracle.hamming_matrix(a, b)
            n = C.c_size_t(0)
            L.match_hamming_count(a, b, na, nb, 3 * nc, nc, C.byref(n))
            ok = ok and np.array_equal(dis, want) and n.value == int((want <= 3 * nc).sum())
            ok = ok and int(L.hamming(a[:1].copy(), b[:1].copy(), nc)) == int(want[0, 0])
        return ok

    with ThreadPoolExecutor(8) as ex:  # ctypes calls release the GIL
        assert all(ex.map(work, range(8)))


def test_hamming_allocating_form_argument_errors_leave_null():
    """sgmx.h: on an argument error *hmptr is NULL (nothing to release), whatever the caller's pointer held before."""
    import ctypes as C
    e = make()
    hm = C.c_void_p(0xDEAD0)  # what an uninitialised hammatch_t* of the caller might hold
    n = C.c_size_t(5)
    a = np.zeros((4, 8), np.uint8)
    rc = e.lib.sgmx_match_hamming_thres(e.h, a.ctypes.data, a.ctypes.data, 4, 4, 3, 0, 16, C.byref(hm), C.byref(n))  # ncodes = 0
    assert rc == 1 and not hm.value
    hm = C.c_void_p(0xDEAD0)
    rc = e.lib.sgmx_crossmatch_hamming(e.h, None, 4, 3, 8, 16, C.byref(hm), C.byref(n))  # null signatures
    assert rc == 1 and not hm.value
    e.close()
