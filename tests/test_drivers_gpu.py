"""GPU tests of the callers and of the remaining BASELINE configurations: the descriptor (CNN)
path, the CLI / batch-driver counterparts, the KITTI-size float volume and the 3840x2160 D=256
stress case (through size-independent properties where the oracle would take minutes)."""
import os
import sys

import numpy as np
import pytest

from conftest import synth_pair

pytestmark = pytest.mark.gpu

sv = pytest.importorskip("stereo_vision_amd")
from stereo_vision_amd import cli, drivers  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _descs(seed, H, W, K, shift=3):
    rng = np.random.default_rng(seed)
    base = rng.standard_normal((H, W + shift, K)).astype(np.float32)
    fl = np.ascontiguousarray(base[:, shift:])
    fr = np.ascontiguousarray(base[:, :W] + 0.05 * rng.standard_normal((H, W, K)).astype(np.float32))
    return fl, fr


@pytest.mark.parametrize("case", [(18, 80, 8, 32), (12, 150, 64, 100), (9, 40, 1, 16),
                                  # every register tile of descriptor_cost_kernel (volume pitch 128 / 256 / 512), several
                                  # pixel tiles per row, K past one LDS chunk with and without a multiple of 4
                                  (7, 300, 38, 200), (6, 200, 12, 300), (5, 270, 68, 128)],
                         ids=lambda c: "x".join(map(str, c)))
def test_descriptor_path(oracle, case):
    """recon::SGM::Compute from descriptors (sgm.cc:76-84 on the GPU).  Parity unpinned by the
    reference (Eigen's .norm() order); checked against the oracle's left-to-right float sum."""
    H, W, K, D = case
    fl, fr = _descs(H + W, H, W, K)
    e = sv.SGM(sv.SGMParams(disp_range=D, penalty1=3, penalty2=60))
    disp = e.compute_desc(fl, fr)
    C = oracle.descriptor_cost(fl, fr, D)
    want = oracle.sgm_cost_f32(C, 3, 60, 8, taps=True)
    S = e.stages(H, W)["S"]
    fin = np.isfinite(want["S"])
    assert np.array_equal(np.isfinite(S), fin)
    rel = np.abs(S[fin] - want["S"][fin]) / np.maximum(np.abs(want["S"][fin]), 1e-30)
    assert rel.max() <= 1e-5  # north_star tolerance on aggregated costs
    assert np.mean(disp == want["disp"]) >= 0.999


def test_sgm_cnn_cli_and_batch(oracle, tmp_path):
    rep = tmp_path / "representation"
    rep.mkdir()
    want = {}
    for i in range(3):
        fl, fr = _descs(50 + i, 14, 70, 8)
        pre = "%06d_10" % i
        sv.save_representation(str(rep / (pre + "_left.bin")), fl)
        sv.save_representation(str(rep / (pre + "_right.bin")), fr)
        want[pre] = oracle.sgm_cost_f32(oracle.descriptor_cost(fl, fr, 40), 3, 60, 8)
    out = tmp_path / "depth"
    drivers.make_output_dirs(str(out))
    assert cli.main(["sgm_cnn", str(rep / "000001_10_left.bin"), str(rep / "000001_10_right.bin"), str(out),
                     "000001_10", "3", "60", "--disp-range", "40"]) == 0
    got = drivers.read_png16(str(out / "disparities" / "000001_10.png"))
    assert np.mean(got == want["000001_10"]) >= 0.999
    assert cli.main(["batch_cnn", str(tmp_path), "3", "60", "--disp-range", "40"]) == 0
    for pre, w in want.items():
        assert np.mean(drivers.read_png16(str(out / "disparities" / (pre + ".png"))) == w) >= 0.999
        assert os.path.exists(str(out / "norm_hist" / (pre + ".png")))


def test_sgm_single_cli_on_tsukuba(oracle, tsukuba, tmp_path):
    """sgm_single argv and outputs; D and wsz set to the golden configuration."""
    from test_oracle_golden import KNOWN
    out = tmp_path / "o"
    drivers.make_output_dirs(str(out))
    g = os.path.join(ROOT, "tests", "golden")
    assert cli.main(["sgm_single", os.path.join(g, "tsukuba_l.png"), os.path.join(g, "tsukuba_r.png"), str(out), "3", "60",
                     "--disp-range", "16", "--window-sz", "5"]) == 0
    disp = drivers.read_png16(str(out / "disparities" / "tsukuba_l.png"))
    s, nz, fnv = KNOWN[("tsukuba", 16, 5, 8)]
    assert int(disp.sum(dtype=np.uint64)) == s and drivers.fnv_u16(disp) == fnv
    prev = drivers.read_gray(str(out / "norm_hist" / "tsukuba_l.png"))
    assert np.array_equal(prev, drivers.normalize_minmax_u8(drivers.to_u8_div256(disp)))


def test_batch_census_driver_shards(oracle, tmp_path):
    """run_sgm_batch_census over 5 pairs as 1 rank and as 2 ranks: same files, bit-exact vs the oracle,
    and the XOR of the per-rank checksums equals the single-rank one."""
    from PIL import Image
    left, right = tmp_path / "image_0", tmp_path / "image_1"
    left.mkdir(), right.mkdir()
    want = {}
    for i in range(5):
        l, r = synth_pair(900 + i, 36, 110, 30)
        name = "%06d_10.png" % i
        Image.fromarray(l).save(str(left / name))
        Image.fromarray(r).save(str(right / name))
        want[name[:-4]] = oracle.sgm_census(l, r, 64, 3, 3, 60, 8)
    (left / "000000_11.png").write_bytes(b"not a _10 frame")
    out1 = tmp_path / "out1"
    f1, c1 = drivers.run_batch_census(str(left), str(right), str(out1), 3, 60, 64, 3, batch=2)
    assert f1 == 5
    for st, w in want.items():
        assert np.array_equal(drivers.read_png16(str(out1 / "disparities" / (st + ".png"))), w), st
    out2 = tmp_path / "out2"
    drivers.make_output_dirs(str(out2))
    fa, ca = drivers.run_batch_census(str(left), str(right), str(out2), 3, 60, 64, 3, rank=0, world=2, batch=4)
    fb, cb = drivers.run_batch_census(str(left), str(right), str(out2), 3, 60, 64, 3, rank=1, world=2, batch=4)
    assert (fa, fb) == (3, 2) and (ca ^ cb) == c1
    for st, w in want.items():
        assert np.array_equal(drivers.read_png16(str(out2 / "disparities" / (st + ".png"))), w), st


def test_grid_search_driver(oracle, tmp_path):
    """run_sgm_batch_grid_search over a 2 x 2 penalty grid and 3 numbered frames, as one rank and as two:
    every result folder holds disparity maps bit-exact vs the oracle for its own (P1, P2)."""
    from PIL import Image
    data = tmp_path / "training"
    (data / "image_0").mkdir(parents=True), (data / "image_1").mkdir()
    frames = []
    for i in range(3):
        l, r = synth_pair(950 + i, 30, 100, 24)
        Image.fromarray(l).save(str(data / "image_0" / ("%06d_10.png" % i)))
        Image.fromarray(r).save(str(data / "image_1" / ("%06d_10.png" % i)))
        frames.append((l, r))
    prefix = str(tmp_path / "experiment")
    res = drivers.run_grid_search(str(data), prefix, (3, 7), (20, 60), 0, 2, disp_range=32, window_sz=5, batch=2)
    assert sorted(res) == [(3, 20), (3, 60), (7, 20), (7, 60)] and all(v[0] == 3 for v in res.values())
    for (p1, p2) in res:
        folder = drivers.grid_search_folder(prefix, p1, p2)
        for i, (l, r) in enumerate(frames):
            got = drivers.read_png16(os.path.join(folder, "disparities", "%06d_10.png" % i))
            assert np.array_equal(got, oracle.sgm_census(l, r, 32, 5, p1, p2, 8)), (p1, p2, i)
    ra = drivers.run_grid_search(str(data), prefix, (3,), (60,), 0, 2, disp_range=32, window_sz=5, rank=0, world=2, write=False)
    rb = drivers.run_grid_search(str(data), prefix, (3,), (60,), 0, 2, disp_range=32, window_sz=5, rank=1, world=2, write=False)
    assert (ra[(3, 60)][0], rb[(3, 60)][0]) == (2, 1) and (ra[(3, 60)][1] ^ rb[(3, 60)][1]) == res[(3, 60)][1]


def test_create_rgbd_dataset_driver(oracle, tmp_path):
    """create_rgbd_dataset -train over two frames: depth bit-exact vs the oracle's CalculateDepthImage on the
    oracle's disparity, colour copy, recoloured ground truth, equalised preview."""
    from PIL import Image
    data, out = tmp_path / "data", tmp_path / "out"
    for d in ("left", "right", "calib", "groundtruth"):
        (data / d).mkdir(parents=True)
    out.mkdir()
    calib = "P0: 700 0 60 0 0 700 20 0 0 0 1 0\nP1: 700 0 60 -370 0 700 20 0 0 0 1 0\n" \
            "P2: 690 0 61 40 0 690 21 0 0 0 1 0\nP3: 690 0 61 -320 0 690 21 0 0 0 1 0\n"
    frames = {}
    for i in range(2):
        l, r = synth_pair(970 + i, 40, 150, 60)
        name = "um_%06d.png" % i
        Image.fromarray(np.stack([l, l // 2, 255 - l], -1)).save(str(data / "left" / name))  # colour left image
        Image.fromarray(r).save(str(data / "right" / name))
        (data / "calib" / (name[:-4] + ".txt")).write_text(calib)
        gt = np.zeros((40, 150, 3), np.uint8)
        gt[5:20, 10:90] = (255, 0, 0)
        Image.fromarray(gt).save(str(data / "groundtruth" / ("um_road_%06d.png" % i)))
        frames[name[:-4]] = (drivers.read_gray(str(data / "left" / name)), r)
    depths = drivers.create_rgbd_dataset("-train", str(data), str(out))
    assert len(depths) == 2
    for k, (prefix, (l, r)) in enumerate(sorted(frames.items())):
        disp = oracle.sgm_census(l, r, 120, 5, 5, 60, 8)
        want = oracle.depth_image(disp, 690.0, 360.0 / 690.0)
        got = drivers.read_png16(str(out / (prefix + "_depth.png")))
        assert np.array_equal(got, want) and np.array_equal(depths[k], want)
        assert (got > 0).any()
        gt = np.asarray(Image.open(str(out / (prefix + "_ground_truth.png"))))
        assert tuple(gt[10, 50]) == (0, 0, 255) and tuple(gt[30, 100]) == (0, 255, 0)
        vis = drivers.read_gray(str(out / (prefix + "_depth_visualization.png")))
        assert np.array_equal(vis, drivers.equalize_hist(drivers.to_u8_div256(disp)))
        assert os.path.exists(str(out / (prefix + "_colors.png")))


def test_float_volume_kitti_size(oracle):
    """BASELINE config 4: precomputed float data-cost volume, KITTI size 1241x376, D=128, 8 paths."""
    H, W, D = 376, 1241, 128
    rng = np.random.default_rng(7)
    C = rng.random((H, W, D), dtype=np.float32) * np.float32(4)
    ties = rng.random((H, W, D), dtype=np.float32) < 0.01
    C[ties] = np.round(C[ties] * 4) / 4  # 1 % exact ties
    C[:, np.arange(D)[None, :] > np.arange(W)[:, None]] = np.finfo(np.float32).max  # sgm.cc:55
    e = sv.SGM(sv.SGMParams(disp_range=D, penalty1=3, penalty2=60))
    disp = e.compute_cost(C)
    want = oracle.sgm_cost_f32(C, 3, 60, 8, taps=True)
    S = e.stages(H, W)["S"]
    fin = np.isfinite(want["S"])
    assert np.array_equal(np.isfinite(S), fin)
    rel = np.abs(S[fin] - want["S"][fin]) / np.maximum(np.abs(want["S"][fin]), 1e-30)
    assert rel.max() <= 1e-5
    assert np.mean(disp == want["disp"]) >= 0.999
    assert int(np.sum(disp != want["disp"])) <= 9 * max(1, want["n_undefined"])


def test_4k_stress_two_formulations_agree():
    """BASELINE config 5: synthetic 3840x2160, D=256, 8 paths.  The CPU oracle needs ~20 GB and
    minutes here, so the full-size check is a size-independent property: the two independent GPU
    formulations (census-recompute + byte volumes vs materialised C/S volumes, each bit-exact
    against the oracle at oracle-sized inputs) must agree bit for bit, and a rerun must too."""
    sys.path.insert(0, ROOT)
    from bench import synthetic_pair
    left, right = synthetic_pair(4242, 2160, 3840, 240)
    fast = sv.StereoSGM(sv.StereoSGMParams(disp_range=256, window_sz=5, penalty1=3, penalty2=60))
    a = fast.compute(left, right)
    b = fast.compute(left, right)
    assert np.array_equal(a, b)
    fast.close()
    mat = sv.StereoSGM(sv.StereoSGMParams(disp_range=256, window_sz=5, penalty1=3, penalty2=60, flags=sv.FLAG_MATERIALIZE))
    c = mat.compute(left, right)
    mat.close()
    assert np.array_equal(a, c)
    assert np.count_nonzero(a) > 0.5 * a.size  # most pixels pass the LR check on this textured pair


def test_large_pairs_run_in_cache_sized_launches():
    """Pairs whose census images would not stay in the Infinity Cache together are run in several kernel launches
    inside one call (sgmx.h, stage getters): 3000 x 2000 at D = 64 is 48 MB of census images per pair,
    so a call with 5 pairs runs 4 + 1.  Every pair must come out as when it is computed alone (the pairs of the
    reference's batch scripts are independent, scripts/run_sgm_batch_census.py:27-34), and the stage getters must
    see the last launch, which has one pair."""
    rng = np.random.default_rng(99)
    H, W, n = 2000, 3000, 5
    base = rng.integers(0, 256, size=(n, H, W + 40), dtype=np.uint8)
    lefts = np.ascontiguousarray(base[:, :, 40:])
    rights = np.ascontiguousarray(base[:, :, :W])
    e = sv.StereoSGM(sv.StereoSGMParams(disp_range=64, window_sz=5, penalty1=3, penalty2=60, max_batch=n))
    got = e.compute_batch(lefts, rights)

    def raw_of(eng, index):  # the disparity image before the median only (C and S would be a gigabyte here)
        raw = np.empty((H, W), np.uint16)
        eng.check(eng.lib.sgmx_get_stages_census(eng.h, index, None, None, raw.ctypes.data))
        return raw

    with pytest.raises(sv.SgmxError):
        raw_of(e, 1)
    raw_last = raw_of(e, 0)
    one = sv.StereoSGM(sv.StereoSGMParams(disp_range=64, window_sz=5, penalty1=3, penalty2=60))
    for k in (0, 3, 4):
        assert np.array_equal(got[k], one.compute(lefts[k], rights[k])), "pair %d" % k
    assert np.array_equal(raw_last, raw_of(one, 0))
    assert np.count_nonzero(got[4]) > 0.3 * got[4].size
    e.close()
    one.close()
