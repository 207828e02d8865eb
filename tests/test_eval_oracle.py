"""CPU checks of the oracle's restatement of the KITTI evaluation (reconstruction/evaluation) against a
vectorised numpy statement of the same definitions."""
import numpy as np


def test_interpolate_background_definition(oracle):
    rng = np.random.default_rng(0)
    H, W = 9, 40
    d = np.where(rng.random((H, W)) < 0.5, rng.integers(1, 5000, (H, W)) / 256.0, -1).astype(np.float32)
    d[4] = -1
    d[0] = -1
    got = oracle.kitti_interpolate_background(d)
    want = d.copy()
    for v in range(H):
        valid = np.flatnonzero(d[v] >= 0)
        if valid.size == 0:
            continue
        for u in range(W):
            if d[v, u] >= 0:
                continue
            left, right = valid[valid < u], valid[valid > u]
            if left.size and right.size:
                want[v, u] = min(d[v, left[-1]], d[v, right[0]])
            else:
                want[v, u] = d[v, right[0]] if right.size else d[v, left[-1]]
    rows_ok = np.flatnonzero((want >= 0).all(axis=1))
    want[: rows_ok[0]] = want[rows_ok[0]]
    want[rows_ok[-1] + 1:] = want[rows_ok[-1]]
    assert np.array_equal(got, want)
    assert (got[4] < 0).all()  # inner empty rows are never filled


def test_error_statistics_definition(oracle):
    rng = np.random.default_rng(1)
    H, W = 30, 50
    gt = (rng.integers(256, 60 * 256, (H, W)) * (rng.random((H, W)) < 0.5)).astype(np.uint16)
    disp = (np.clip(gt.astype(np.int64) + rng.integers(-1500, 1500, (H, W)), 1, 65535) * (rng.random((H, W)) < 0.7)).astype(np.uint16)
    disp[gt == 0] = rng.integers(0, 2, (gt == 0).sum()) * 999
    out, ipol = oracle.kitti_eval_image(disp, gt, gt, want_ipol=True)
    g = gt / 256.0
    ip = ipol / 256.0
    m = gt > 0
    r = m & (disp > 0)
    err = np.abs(g - ip)
    for k in range(5):
        assert abs(out[2 * k] - (err[m] > k + 1).mean()) < 1e-6
        assert abs(out[2 * k + 1] - (err[r] > k + 1).mean()) < 1e-6
    assert abs(out[10] - r.sum() / m.sum()) < 1e-6
    assert abs(out[22] - err[m].mean()) < 1e-4 * max(1.0, err[m].mean())
    assert abs(out[23] - err[r].mean()) < 1e-4 * max(1.0, err[r].mean())
    assert np.array_equal(out[:11], out[11:22]) and np.array_equal(out[22:24], out[24:26])
