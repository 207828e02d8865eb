"""Pins the CPU oracle (oracle/sgm_oracle.c) to the reference.

Known answers: SURVEY.md section 8c -- outputs of the reference's own
recon::StereoSGM::compute (stereo_sgm.cc:15-188) on the reference's own fixtures
reconstruction/test/{tsukuba_l,tsukuba_r,left_000087,right_000087}.png, P1=3, P2=60,
recorded as sum / non-zero count / 64-bit FNV-style hash over the uint16 output.
The Hamming primitive is additionally pinned against the reference's core/hamming.c
compiled as-is (oracle/_ref/libhamming_ref.so) when that build is present.
"""
import numpy as np
import pytest

# (fixture, D, wsz, num_paths) -> (sum, nonzero, fnv)
KNOWN = {
    ("tsukuba", 16, 5, 8): (177734591, 105163, 0xFB2353D52A6B2EE2),
    ("tsukuba", 16, 5, 4): (177041394, 104788, 0x456B80E5D90EF209),
    ("tsukuba", 16, 3, 8): (178288796, 106672, 0xDE77C8FF19A1052F),
    ("kitti", 128, 5, 8): (2795828975, 415382, 0x4ACA717F5249B4D4),
    ("kitti", 128, 5, 4): (2783554156, 411634, 0x1124DB038A65BA05),
    ("kitti", 128, 3, 8): (2656778054, 412072, 0xE894A754560277B3),
}
# with the per-direction reset the reference does NOT do (regression sentinels for the carry-over)
RESET_FNV = {("tsukuba", 16, 5, 8): 0xC7390A4498379D69, ("kitti", 128, 5, 8): 0x3858AD135DE9605A}


@pytest.mark.parametrize("key", sorted(KNOWN))
def test_known_answers(oracle, tsukuba, kitti, key):
    name, D, wsz, paths = key
    left, right = tsukuba if name == "tsukuba" else kitti
    disp = oracle.sgm_census(left, right, D, wsz, 3, 60, paths)
    s, nz, fnv = KNOWN[key]
    assert disp.shape == left.shape and disp.dtype == np.uint16
    assert int(disp.sum(dtype=np.uint64)) == s
    assert int(np.count_nonzero(disp)) == nz
    assert oracle.fnv_u16(disp) == fnv


def test_carry_over_is_output_visible(oracle, tsukuba):
    key = ("tsukuba", 16, 5, 8)
    disp = oracle.sgm_census(*tsukuba, 16, 5, 3, 60, 8, reset_between_dirs=True)
    assert oracle.fnv_u16(disp) == RESET_FNV[key]
    assert oracle.fnv_u16(disp) != KNOWN[key][2]


def test_kitti_per_direction_maxima(oracle, kitti):
    # per-direction max L = 315,315,570,570,570,825,1080,1335; max S = 5325
    r = oracle.sgm_census(*kitti, 128, 5, 3, 60, 8, taps=True)
    assert r["Lmax"].tolist() == [315, 315, 570, 570, 570, 825, 1080, 1335]
    assert int(r["S"].max()) == 5325


def test_fnv_matches_independent_python():
    from oracle import oracle as o
    v = np.array([0, 1, 65535, 256, 77], np.uint16)
    h = 1469598103934665603
    for x in v.tolist():
        h = ((h ^ x) * 1099511628211) % (1 << 64)
    assert o.fnv_u16(v) == h


@pytest.mark.parametrize("dx,dy", [(-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1)])
@pytest.mark.parametrize("shape", [(7, 11, 5), (11, 7, 4), (1, 9, 3), (9, 1, 3), (2, 2, 2), (1, 1, 2)])
def test_path_walk_equals_edge_walk(oracle, dx, dy, shape):
    """The oracle's path-walk sweep and the literal restatement of the reference's four
    edge blocks (stereo_sgm.cc:197-267) agree, including on a dirty carry-over buffer."""
    h, w, D = shape
    rng = np.random.default_rng(h * 100 + w * 10 + D)
    C = rng.integers(0, 256, size=(h, w, D)).astype(np.uint8)
    buf0 = rng.integers(0, 500, size=(h, w, D)).astype(np.uint32)
    a = oracle.aggregate_dir_u32(C, dx, dy, 3, 20, buf0.copy())
    b = oracle.aggregate_dir_u32(C, dx, dy, 3, 20, buf0.copy(), edges=True)
    assert np.array_equal(a, b)


def _py_path_step(prior, local, P1, P2):
    D = len(prior)
    mp = min(prior)
    out = []
    for d in range(D):
        e = min(mp + P2, prior[d])
        if d > 0:
            e = min(e, prior[d - 1] + P1)
        if d < D - 1:
            e = min(e, prior[d + 1] + P1)
        out.append(local[d] + e - mp)
    return out


def test_aggregate_against_pure_python(oracle):
    """Horizontal left-to-right sweep against a direct transcription of the recurrence
    in stereo_sgm.h:119-147 (pure Python, tiny case)."""
    rng = np.random.default_rng(5)
    h, w, D = 3, 6, 7
    C = rng.integers(0, 40, size=(h, w, D)).astype(np.uint8)
    buf = oracle.aggregate_dir_u32(C, 1, 0, 2, 9, np.zeros((h, w, D), np.uint32))
    for y in range(h):
        prior = C[y, 0].astype(int).tolist()
        assert buf[y, 0].tolist() == prior
        for x in range(1, w):
            prior = _py_path_step(prior, C[y, x].astype(int).tolist(), 2, 9)
            assert buf[y, x].tolist() == prior


def test_census_bit_order_and_wrap(oracle):
    img = np.arange(49, dtype=np.uint8).reshape(7, 7)
    # 3x3: neighbours in row-major order, centre skipped, MSB first
    c3 = oracle.census(img, 3)
    assert c3.shape == (5, 5)
    assert int(c3[0, 0]) == 0b00001111  # the 4 later neighbours are brighter
    # 7x7 has 48 taps: only the low 32 bits survive (stereo_costs.cc:55)
    c7 = oracle.census(img, 7)
    assert c7.shape == (1, 1)
    assert int(c7[0, 0]) == (1 << 24) - 1
    # wsz=1: no taps, zero code, no crop
    assert not oracle.census(img, 1).any() and oracle.census(img, 1).shape == (7, 7)


def test_hamming_volume_invalid_is_255(oracle):
    rng = np.random.default_rng(2)
    cl = rng.integers(0, 1 << 24, size=(4, 9)).astype(np.uint32)
    cr = rng.integers(0, 1 << 24, size=(4, 9)).astype(np.uint32)
    C = oracle.hamming_volume(cl, cr, 6)
    for x in range(9):
        for d in range(6):
            if d > x:
                assert (C[:, x, d] == 255).all()
            else:
                want = [bin(int(a) ^ int(b)).count("1") for a, b in zip(cl[:, x], cr[:, x - d])]
                assert C[:, x, d].tolist() == want


@pytest.mark.parametrize("ncodes", [4, 8, 16, 5])
def test_hamming_matrix_vs_reference_build(oracle, ncodes):
    """oracle_hamming_matrix_u16 == core/hamming.c compute_hamming (reference build)."""
    rng = np.random.default_rng(ncodes)
    a = rng.integers(0, 256, size=(37, ncodes)).astype(np.uint8)
    b = rng.integers(0, 256, size=(23, ncodes)).astype(np.uint8)
    mine = oracle.hamming_matrix(a, b)
    want = np.array([[bin(int.from_bytes(bytes(x ^ y), "little")).count("1") for x in a] for y in b], np.uint16)
    assert np.array_equal(mine, want)
    ref = oracle.ref_hamming_matrix(a, b)
    if ref is None:
        pytest.skip("oracle/_ref/libhamming_ref.so not built (reference tree not mounted)")
    assert np.array_equal(mine, ref)


@pytest.mark.parametrize("ncodes,ht", [(4, 12), (8, 26), (16, 56), (5, 17), (4, 0), (4, 32)])
@pytest.mark.parametrize("cross", [False, True])
def test_hamming_matchers_vs_reference_build(oracle, ncodes, ht, cross):
    """oracle_match_hamming == core/hamming.c match_hamming_count / match_hamming_thres_prealloc /
    crossmatch_hamming_count / crossmatch_hamming_prealloc (reference build): same matches, same order."""
    rng = np.random.default_rng(ncodes * 100 + ht)
    a = rng.integers(0, 256, size=(61, ncodes)).astype(np.uint8)
    b = rng.integers(0, 256, size=(45, ncodes)).astype(np.uint8)
    b[7] = a[3]      # an exact match
    a[20] = a[11]    # and one inside the set, for the cross form
    idx, hams = oracle.match_hamming(a, b, ht, cross)
    bb = a if cross else b
    want = [(i, j, bin(int.from_bytes(bytes(a[i] ^ bb[j]), "little")).count("1"))
            for i in range(len(a)) for j in range(i + 1 if cross else 0, len(bb))]
    want = [(i, j, h) for i, j, h in want if h <= ht]
    assert [(int(i), int(j), int(h)) for (i, j), h in zip(idx, hams)] == want
    ref = oracle.ref_match_hamming(a, b, ht, cross)
    if ref is None:
        pytest.skip("oracle/_ref/libhamming_ref.so not built (reference tree not mounted)")
    assert np.array_equal(idx, ref[0]) and np.array_equal(hams, ref[1])


def test_median3_replicate_border(oracle):
    rng = np.random.default_rng(9)
    img = rng.integers(0, 65536, size=(6, 8)).astype(np.uint16)
    got = oracle.median3(img)
    pad = np.pad(img, 1, mode="edge")
    want = np.empty_like(img)
    for y in range(6):
        for x in range(8):
            want[y, x] = np.sort(pad[y:y + 3, x:x + 3].reshape(-1))[4]
    assert np.array_equal(got, want)


def test_float_path_matches_integer_path_on_integer_costs(oracle):
    """With integer-valued float costs every float op is exact, so the float build
    (sgm.h:129-157, sgm.cc:177-254) must reproduce the uint32 build's sums."""
    rng = np.random.default_rng(11)
    h, w, D = 9, 14, 8
    Cu = rng.integers(0, 64, size=(h, w, D)).astype(np.uint8)
    bu = np.zeros((h, w, D), np.uint32)
    bf = np.zeros((h, w, D), np.float32)
    Cf = Cu.astype(np.float32)
    for dx, dy in oracle.directions(8):
        oracle.aggregate_dir_u32(Cu, dx, dy, 3, 17, bu)
        oracle.aggregate_dir_f32(Cf, dx, dy, 3, 17, bf)
        assert np.array_equal(bu.astype(np.float32), bf)


def test_gaussian3_against_numpy(oracle):
    """Caller-side pre-blur (dense_stereo/main.cc:83-85): the C restatement against a direct numpy
    statement of the same fixed-point arithmetic.  PARITY UNPINNED by the reference (OpenCV is not
    in its tree and its version is not pinned); sigma = 0.7 gives the taps 54/149/54 (sum 257: the
    2.4-era integer path rounds the float taps independently)."""
    assert oracle.gaussian3_kernel_q8(0.7) == (149, 54)
    assert oracle.gaussian3_kernel_q8(0.0) == oracle.gaussian3_kernel_q8(0.8)  # automatic sigma for ksize 3
    rng = np.random.default_rng(3)
    for shape in [(9, 13), (1, 7), (7, 1), (2, 2), (1, 1)]:
        img = rng.integers(0, 256, size=shape).astype(np.uint8)
        k0, k1 = oracle.gaussian3_kernel_q8(0.7)
        pad = img.astype(np.int64)
        pad = np.pad(pad, ((1, 1) if shape[0] > 1 else (0, 0), (1, 1) if shape[1] > 1 else (0, 0)), mode="reflect")
        if shape[1] > 1:
            rows = k1 * (pad[:, :-2] + pad[:, 2:]) + k0 * pad[:, 1:-1]
        else:
            rows = (2 * k1 + k0) * pad
        if shape[0] > 1:
            acc = k1 * (rows[:-2] + rows[2:]) + k0 * rows[1:-1]
        else:
            acc = (2 * k1 + k0) * rows
        want = np.minimum((acc + (1 << 15)) >> 16, 255).astype(np.uint8)
        assert np.array_equal(oracle.gaussian3(img, 0.7), want), shape


def test_disparities_past_255_wrap_like_the_reference_builds(oracle):
    """disp_range > 256: value = 256 (d + ds) leaves uint16 and the reference's static_cast<uint16_t>(double)
    (stereo_sgm.h:309,312) is undefined; its x86-64 builds keep the low 16 bits.  The oracle restates that explicitly
    (cast_u16_like_x86); this pins the restatement to the formula on every pixel whose winner is d >= 256."""
    from conftest import synth_pair
    H, W, D, wsz = 14, 700, 512, 3
    left, right = synth_pair(H * 1000 + W + D, H, W, W // 2)
    r = oracle.sgm_census(left, right, D, wsz, 5, 255, 8, taps=True)
    S = r["S"].astype(np.int64)
    m = (wsz - 1) // 2
    raw = r["raw"][m:H - m, m:W - m]
    am = S.argmin(axis=2)
    big = np.argwhere((am >= 256) & (am < D - 1) & (raw != 0))
    assert len(big) > 20
    for y, x in big:
        d = int(am[y, x])
        cl, cc, cr = (np.float32(S[y, x, d - 1]), np.float32(S[y, x, d]), np.float32(S[y, x, d + 1]))
        ds = np.float32(0.5) * (cr - cl) / ((cc - cl) if cr < cl else (cc - cr))
        v = int(np.floor(256.0 * float(np.float32(d) + ds) + 0.5))  # std::round on a positive value
        assert v >= 65536 and int(raw[y, x]) == v % 65536


@pytest.mark.parametrize("ncodes,ht", [(4, 10), (8, 24), (16, 52), (5, 14)])
def test_hamming_allocating_and_threaded_forms_vs_reference_build(oracle, ncodes, ht):
    """The rest of core/hamming.h (84-96, 108-112) in the reference build itself: match_hamming_thres / crossmatch_hamming
    return the prealloc forms' matches as hammatch_t records; compute_hamming_thread == compute_hamming;
    match_hamming_thres_nt == the 128 x 128 block-pair order with block-local indices that the restatement spells out."""
    rng = np.random.default_rng(ncodes * 7 + ht)
    a = rng.integers(0, 256, size=(300, ncodes)).astype(np.uint8)
    b = rng.integers(0, 256, size=(270, ncodes)).astype(np.uint8)
    if oracle.ref_match_hamming(a, b, ht) is None:
        pytest.skip("oracle/_ref/libhamming_ref.so not built (reference tree not mounted)")
    for cross in (False, True):
        idx, hams = oracle.match_hamming(a, b, ht, cross)
        ridx, rh = oracle.ref_match_hamming_alloc(a, b, ht, cross, bufsize=3)  # small: the reference grows it
        assert np.array_equal(ridx, idx) and np.array_equal(rh, hams)
    assert np.array_equal(oracle.ref_hamming_matrix_thread(a, b), oracle.hamming_matrix(a, b))
    k, h = oracle.match_hamming_blocked(a, b, ht)
    rk, rh = oracle.ref_match_hamming_thres_nt(a, b, ht)
    assert len(h) > 0 and np.array_equal(rk, k) and np.array_equal(rh, h)


def _cost_volume_from_reference_hamming(oracle, cl, cr, D):
    """The data-cost loop of stereo_sgm.cc:75-101 with its popcount done by the reference's own compiled
    core/hamming.c: compute_hamming on the 4-byte census codes of one image row gives popcount(cL[i] ^ cR[j]) for every
    (i, j); C[y][x][d] is the entry (i, j) = (x, x - d) for d <= x and 255 (stereo_sgm.cc:43) otherwise."""
    h, w = cl.shape
    C = np.full((h, w, D), 255, np.uint8)
    for y in range(h):
        dis = oracle.ref_hamming_matrix(cl[y].view(np.uint8).reshape(w, 4), cr[y].view(np.uint8).reshape(w, 4))  # [j][i]
        if dis is None:
            return None
        for d in range(min(D, w)):
            xs = np.arange(d, w)
            C[y, xs, d] = dis[xs - d, xs]
    return C


def test_data_cost_volume_vs_reference_popcount(oracle, tsukuba):
    """the restatement's Hamming data-cost volume on the census codes of the reference's own
    Tsukuba fixture equals the volume built with core/hamming.c's compute_hamming (reference build, oracle/_ref)."""
    left, right = tsukuba
    cl, cr = oracle.census(left[100:112], 5), oracle.census(right[100:112], 5)
    want = _cost_volume_from_reference_hamming(oracle, np.ascontiguousarray(cl), np.ascontiguousarray(cr), 16)
    if want is None:
        pytest.skip("oracle/_ref/libhamming_ref.so not built (reference tree not mounted)")
    assert np.array_equal(oracle.hamming_volume(cl, cr, 16), want)
