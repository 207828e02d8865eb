"""CPU checks of the oracle's restatement of recon::SGMStereo (reconstruction/base/new_sgm/sgm_stereo.cc).

PARITY UNPINNED by the reference: it holds no fixture or output of this pipeline and the class cannot be
built here (OpenCV, Eigen).  The literal C restatement (row buffers, two raster passes, explicit flood
fill) is cross-checked against independent formulations of the same definitions: per-direction sweeps,
scipy's connected components, brute-force numpy."""
import numpy as np
import pytest


def smooth_volume(seed, H, W, D, noise=0.3):
    """Float data cost with structure: |d - truth| with a piecewise-constant truth, plus noise."""
    rng = np.random.default_rng(seed)
    truth = np.zeros((H, W), np.int32)
    for by in range(0, H, max(4, H // 3)):
        for bx in range(0, W, max(8, W // 4)):
            truth[by:by + max(4, H // 3), bx:bx + max(8, W // 4)] = rng.integers(1, max(2, min(D - 1, W // 3)))
    d = np.arange(D, dtype=np.float32)[None, None, :]
    C = np.minimum(np.abs(d - truth[..., None]), 6).astype(np.float32) + noise * rng.random((H, W, D), dtype=np.float32)
    return np.ascontiguousarray(C, np.float32), truth


NS_ORDER = [(1, 0), (1, 1), (0, 1), (-1, 1), (-1, 0), (-1, -1), (0, -1), (1, -1)]


@pytest.mark.parametrize("shape", [(12, 40, 16), (9, 33, 33), (1, 20, 8), (20, 8, 8), (2, 2, 2)])
def test_two_pass_aggregation_equals_eight_textbook_sweeps(oracle, shape):
    """PerformSGM's fused two-pass / four-path loop (sgm_stereo.cc:258-360) gives, bit for bit, the sum of
    eight independent path sweeps started from L = C, added in the order the loops visit them."""
    H, W, D = shape
    C, _ = smooth_volume(H * W, H, W, D)
    got = oracle.ns_perform_sgm(C, 3, 40)
    S = np.zeros_like(C)
    for dx, dy in NS_ORDER:
        buf = np.zeros_like(C)
        oracle.aggregate_dir_f32(C, dx, dy, 3, 40, buf)
        S = S + buf
    assert np.array_equal(got["S"], S)
    # WTA: first minimum; sub-pixel with the pipeline's + 0.5 (sgm_stereo.cc:383-400)
    bd = np.argmin(S, axis=2)
    y, x = np.nonzero((bd > 0) & (bd < D - 1))
    sp = bd.astype(np.float32)
    for yy, xx in zip(y, x):
        b = int(bd[yy, xx])
        l, c, r = S[yy, xx, b - 1], S[yy, xx, b], S[yy, xx, b + 1]
        q = np.float32(r - l) / np.float32(c - l if r < l else c - r)
        sp[yy, xx] = np.float32(np.float64(b) + np.float64(q) / 2.0 + 0.5)
    assert np.array_equal(got["subpix"], sp)
    assert np.array_equal(got["disp"], oracle.ns_speckle(bd.astype(np.int32)))


def test_right_cost_volume(oracle):
    rng = np.random.default_rng(1)
    H, W, D = 3, 11, 5
    L = rng.random((H, W, D), dtype=np.float32)
    R = oracle.ns_right_cost(L)
    for x in range(W):
        for d in range(D):
            want = L[:, x + d, d] if x + d < W else L[:, W - 1, W - 1 - x]
            assert np.array_equal(R[:, x, d], want), (x, d)
    with pytest.raises(ValueError):
        oracle.ns_right_cost(np.zeros((2, 4, 5), np.float32))  # the reference runs out of bounds when W < D


def test_left_cost_from_descriptors_repeats_the_last_valid_cost(oracle):
    rng = np.random.default_rng(2)
    fl = rng.standard_normal((4, 9, 3)).astype(np.float32)
    fr = rng.standard_normal((4, 9, 3)).astype(np.float32)
    C = oracle.ns_left_cost_desc(fl, fr, 6)
    main = oracle.descriptor_cost(fl, fr, 6)  # the recon::SGM variant: FLT_MAX where x < d
    for x in range(9):
        for d in range(6):
            assert np.array_equal(C[:, x, d], main[:, x, min(d, x)])


@pytest.mark.parametrize("seed", range(4))
def test_speckle_filter_against_scipy_components(oracle, seed):
    """Regions = connected components of the graph whose edges join 4-neighbours that are both non-zero and
    differ by at most 2; components of at most `max_size` pixels are zeroed (sgm_stereo.cc:416-488)."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    rng = np.random.default_rng(seed)
    H, W = 40, 57
    img = rng.integers(0, 6, (H, W)).astype(np.int32) * 3  # values 0, 3, 6, ...: never linked unless equal
    img[5:30, 10:40] = 7 + (np.arange(30)[None, :] // 8)   # a big region with steps of 1
    img[rng.random((H, W)) < 0.1] = 0
    idx = np.arange(H * W).reshape(H, W)
    rows, cols = [], []
    for a, b, ia, ib in [(img[:, :-1], img[:, 1:], idx[:, :-1], idx[:, 1:]), (img[:-1], img[1:], idx[:-1], idx[1:])]:
        m = (a != 0) & (b != 0) & (np.abs(a - b) <= 2)
        rows += ia[m].tolist()
        cols += ib[m].tolist()
    n, lab = connected_components(coo_matrix((np.ones(len(rows)), (rows, cols)), shape=(H * W, H * W)), directed=False)
    size = np.bincount(lab, minlength=n)
    for max_size in (5, 100):
        want = img.copy()
        want[(size[lab].reshape(H, W) <= max_size) & (img != 0)] = 0
        assert np.array_equal(oracle.ns_speckle(img, max_size, 2), want)


def test_lr_states_brute_force(oracle):
    rng = np.random.default_rng(3)
    H, W, D = 6, 30, 12
    left = rng.integers(0, D, (H, W)).astype(np.int32)
    right = rng.integers(0, D, (H, W)).astype(np.int32)
    for thr in (0, 1, 2):
        st = oracle.ns_lr_states(left, right, D, thr)
        for y in range(H):
            for x in range(W):
                ld = left[y, x]
                if ld > x:
                    want = "o"
                elif abs(ld - right[y, x - ld]) <= thr:
                    want = "c"
                elif any(d != ld and abs(ld - right[y, x - d]) <= thr for d in range(0, min(x, D) + 1)):
                    want = "m"
                else:
                    want = "o"
                assert chr(st[y, x]) == want


def test_interpolation_fills_occlusions_from_the_left(oracle):
    st = np.frombuffer(b"oocmoomcoo", np.uint8).reshape(1, 10).copy()
    d = np.arange(10, dtype=np.float32).reshape(1, 10) + 0.25
    s2, d2 = oracle.ns_interpolate(st, d)
    assert s2.tobytes() == b"oocmccmccc"
    assert d2[0].tolist() == [0.25, 1.25, 2.25, 3.25, 2.25, 2.25, 6.25, 7.25, 7.25, 7.25]
    # vectorised statement: nearest 'c' strictly to the left, 'm' skipped
    rng = np.random.default_rng(5)
    st = rng.choice(np.frombuffer(b"cmo", np.uint8), size=(7, 50)).astype(np.uint8)
    d = rng.random((7, 50), dtype=np.float32)
    s2, d2 = oracle.ns_interpolate(st, d)
    for y in range(7):
        last = None
        for x in range(50):
            if st[y, x] == ord("o") and last is not None:
                assert d2[y, x] == d[y, last] and s2[y, x] == ord("c")
            else:
                assert d2[y, x] == d[y, x] and s2[y, x] == st[y, x]
            if st[y, x] == ord("c"):
                last = x


def test_median5_replicate_border(oracle):
    rng = np.random.default_rng(9)
    img = rng.integers(0, 65536, size=(7, 9)).astype(np.uint16)
    pad = np.pad(img, 2, mode="edge")
    want = np.empty_like(img)
    for y in range(7):
        for x in range(9):
            want[y, x] = np.sort(pad[y:y + 5, x:x + 5].reshape(-1))[12]
    assert np.array_equal(oracle.median5(img), want)
    one = rng.integers(0, 65536, size=(1, 6)).astype(np.uint16)
    assert oracle.median5(one).shape == (1, 6)


def test_full_pipeline_recovers_the_planted_disparities(oracle):
    """End to end on a structured volume: the result is 256 * (truth + 0.5) up to the sub-pixel term on
    most pixels (the + 0.5 is the reference's own, sgm_stereo.cc:388-395)."""
    H, W, D = 40, 120, 32
    C, truth = smooth_volume(11, H, W, D)
    r = oracle.ns_compute(C, 3, 40, 1, taps=True)
    assert r["disp"].shape == (H, W) and r["disp"].dtype == np.uint16
    inner = np.s_[4:-4, 40:-8]
    err = np.abs(r["disp"][inner].astype(np.float64) / 256.0 - 0.5 - truth[inner])
    assert np.mean(err < 1.0) > 0.9
    assert set(np.unique(r["states"])) <= {ord("c"), ord("m"), ord("o")}
    assert np.array_equal(r["disp"], oracle.median5(r["raw"]))
