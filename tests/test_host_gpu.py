"""GPU tests of the C++ host layer: the CLI programs under stereo_vision_amd/bin (counterparts of
reconstruction/main/{sgm_single,sgm_cnn,dense_stereo,stereo_eval} and scripts/run_sgm_batch*.py)
run as the reference's binaries are run -- by argv, results read back from their output files --
and checked against the CPU oracle / the reference's known answers."""
import os
import re
import subprocess

import numpy as np
import pytest
from PIL import Image

from conftest import synth_pair

pytestmark = pytest.mark.gpu

sv = pytest.importorskip("stereo_vision_amd")
from stereo_vision_amd import drivers  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "stereo_vision_amd", "bin")
G = os.path.join(ROOT, "tests", "golden")


def run(prog, *args, **kw):
    r = subprocess.run([os.path.join(BIN, prog)] + [str(a) for a in args], capture_output=True, text=True, timeout=110, **kw)
    assert r.returncode == 0, (prog, r.returncode, r.stderr[-2000:])
    return r


def test_sgm_single_binary_on_the_reference_fixture(oracle, tmp_path):
    """sgm_single left right out P1 P2 on reconstruction/test's Tsukuba pair at the golden
    configuration: the 16-bit PNG carries the reference's known answer."""
    from test_oracle_golden import KNOWN
    out = tmp_path / "o"
    drivers.make_output_dirs(str(out))
    run("sgm_single", os.path.join(G, "tsukuba_l.png"), os.path.join(G, "tsukuba_r.png"), out, 3, 60,
        "--disp-range", 16, "--window-sz", 5)
    disp = drivers.read_png16(str(out / "disparities" / "tsukuba_l.png"))
    s, nz, fnv = KNOWN[("tsukuba", 16, 5, 8)]
    assert int(disp.sum(dtype=np.uint64)) == s and int(np.count_nonzero(disp)) == nz and drivers.fnv_u16(disp) == fnv
    prev = np.array(Image.open(out / "norm_hist" / "tsukuba_l.png"))
    assert np.array_equal(prev, drivers.normalize_minmax_u8(drivers.to_u8_div256(disp)))
    # the mains' own hard-coded configuration (D = 256, window 3) against the oracle, 4 paths as an option
    run("sgm_single", os.path.join(G, "tsukuba_l.png"), os.path.join(G, "tsukuba_r.png"), out, 5, 40, "--paths", 4)
    l, r = (np.array(Image.open(os.path.join(G, n))) for n in ("tsukuba_l.png", "tsukuba_r.png"))
    assert np.array_equal(drivers.read_png16(str(out / "disparities" / "tsukuba_l.png")),
                          oracle.sgm_census(l, r, 256, 3, 5, 40, 4))


def test_sgm_cnn_binary(oracle, tmp_path):
    rng = np.random.default_rng(4)
    H, W, K, D = 13, 90, 8, 48
    base = rng.standard_normal((H, W + 5, K)).astype(np.float32)
    fl = np.ascontiguousarray(base[:, 5:])
    fr = np.ascontiguousarray(base[:, :W] + 0.05 * rng.standard_normal((H, W, K)).astype(np.float32))
    sv.save_representation(str(tmp_path / "000007_10_left.bin"), fl)
    sv.save_representation(str(tmp_path / "000007_10_right.bin"), fr)
    out = tmp_path / "depth"
    drivers.make_output_dirs(str(out))
    run("sgm_cnn", tmp_path / "000007_10_left.bin", tmp_path / "000007_10_right.bin", out, "000007_10", 3, 60,
        "--disp-range", D)
    got = drivers.read_png16(str(out / "disparities" / "000007_10.png"))
    want = oracle.sgm_cost_f32(oracle.descriptor_cost(fl, fr, D), 3, 60, 8)
    assert got.shape == (H, W) and np.mean(got == want) >= 0.999
    assert (out / "norm_hist" / "000007_10.png").exists()


def test_dense_stereo_binary(oracle, tmp_path):
    """Image-list XML + config file in, img/disp_<n>.png and mat/disp_<n>.yml out
    (dense_stereo/main.cc:76-114), including the Gaussian pre-blur."""
    src = tmp_path / "seq"
    (src / "image_0").mkdir(parents=True), (src / "image_1").mkdir()
    names, pairs = [], []
    for i in range(2):
        l, r = synth_pair(40 + i, 40, 120, 30)
        Image.fromarray(l).save(src / "image_0" / ("%06d.png" % i))
        Image.fromarray(r).save(src / "image_1" / ("%06d.png" % i))
        names += ["/image_0/%06d.png" % i, "/image_1/%06d.png" % i]
        pairs.append((l, r))
    lst = tmp_path / "lst.xml"
    lst.write_text('<?xml version="1.0"?>\n<opencv_storage>\n<imagelist>\n' + "\n".join('"%s"' % n for n in names) +
                   "\n</imagelist>\n</opencv_storage>\n")
    cfg = tmp_path / "config.txt"
    outdir = str(tmp_path / "disp") + "/"
    cfg.write_text("source_folder = %s\ndisp_folder = %s\nimglist = %s\nstart_num = 0\n" % (src, outdir, lst))
    r = run("dense_stereo", "-c", cfg, "--disp-range", 64)
    assert "Saving: " + outdir + "img/disp_000000.png" in r.stdout
    import yaml
    for i, (l, rr) in enumerate(pairs):
        want = oracle.sgm_census(oracle.gaussian3(l, 0.7), oracle.gaussian3(rr, 0.7), 64, 5, 7, 60, 8)
        text = open(outdir + "mat/disp_%06d.yml" % i).read().split("\n", 1)[1].replace("!!opencv-matrix", "")
        m = yaml.safe_load(text)["disparity_matrix"]
        got = np.array(m["data"], np.uint16).reshape(m["rows"], m["cols"])
        assert np.array_equal(got, want)
        img = np.array(Image.open(outdir + "img/disp_%06d.png" % i))
        assert np.array_equal(img, drivers.equalize_hist(np.minimum(want, 255).astype(np.uint8)))
    # the Python counterpart writes the same previews
    py = drivers.dense_stereo(names, str(src), str(tmp_path / "disp_py") + "/", disp_range=64)
    assert all(np.array_equal(a, oracle.sgm_census(oracle.gaussian3(l, 0.7), oracle.gaussian3(rr, 0.7), 64, 5, 7, 60, 8))
               for a, (l, rr) in zip(py, pairs))
    assert np.array_equal(np.array(Image.open(str(tmp_path / "disp_py") + "/img/disp_000001.png")),
                          np.array(Image.open(outdir + "img/disp_000001.png")))


def test_dense_stereo_batch_pipeline_equals_one_at_a_time(oracle, tmp_path):
    """dense_stereo runs its list through the batch pipeline (launch groups, pre-blur fused on the device, decoders
    and writers beside the engine); --one-at-a-time keeps the loop of dense_stereo/main.cc:76-114.  Same files, same
    bytes, for 5 pairs in groups of 2 at the reference's own disparity range (270: the 12-pairs-per-lane layout)."""
    import filecmp
    src = tmp_path / "seq"
    (src / "image_0").mkdir(parents=True), (src / "image_1").mkdir()
    names = []
    for i in range(5):
        l, r = synth_pair(60 + i, 36, 300, 120)
        Image.fromarray(l).save(src / "image_0" / ("%06d.png" % i))
        Image.fromarray(r).save(src / "image_1" / ("%06d.png" % i))
        names += ["/image_0/%06d.png" % i, "/image_1/%06d.png" % i]
    lst = tmp_path / "lst.xml"
    lst.write_text('<?xml version="1.0"?>\n<opencv_storage>\n<imagelist>\n' + "\n".join('"%s"' % n for n in names) +
                   "\n</imagelist>\n</opencv_storage>\n")
    outs = []
    for k, extra in enumerate((["--batch", 2], ["--one-at-a-time"], ["--batch", 3, "--no-blur"], ["--one-at-a-time", "--no-blur"])):
        out = str(tmp_path / ("disp%d" % k)) + "/"
        r = run("dense_stereo", "-s", src, "-d", out, "-l", lst, "-c", _empty_cfg(tmp_path), *extra)
        assert r.stdout.count("Saving: ") == 5 and "pairs/s" in r.stderr
        outs.append(out)
    for a_, b_ in ((outs[0], outs[1]), (outs[2], outs[3])):
        for i in range(5):
            for f in ("img/disp_%06d.png" % i, "mat/disp_%06d.yml" % i):
                assert filecmp.cmp(a_ + f, b_ + f, shallow=False), f
    assert not filecmp.cmp(outs[0] + "mat/disp_000002.yml", outs[2] + "mat/disp_000002.yml", shallow=False)  # the blur matters


def _empty_cfg(tmp_path):
    c = tmp_path / "empty.txt"
    c.write_text("start_num = 0\n")
    return c


def test_preblur_fused_equals_explicit_blur(oracle):
    """sgmx_set_preblur3: the callers' GaussianBlur (dense_stereo/main.cc:83-85) in front of the census transform, on the
    device, for the one-pair, the batch and the device entry points == blurring first (oracle) and computing."""
    pairs = [synth_pair(90 + i, 33, 140, 40) for i in range(3)]
    want = [oracle.sgm_census(oracle.gaussian3(l, 0.7), oracle.gaussian3(r, 0.7), 64, 5, 3, 60, 8) for l, r in pairs]
    e = sv.StereoSGM(sv.StereoSGMParams(disp_range=64, window_sz=5, penalty1=3, penalty2=60, max_batch=2))
    e.set_preblur(0.7)
    assert np.array_equal(e.compute(*pairs[0]), want[0])
    got = e.compute_batch(np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs]))
    assert all(np.array_equal(got[i], want[i]) for i in range(3))
    e.set_preblur(0.7, enabled=False)
    assert np.array_equal(e.compute(*pairs[0]), oracle.sgm_census(pairs[0][0], pairs[0][1], 64, 5, 3, 60, 8))
    e.set_preblur(0.0)  # OpenCV's default sigma for a 3x3 kernel
    assert np.array_equal(e.compute(*pairs[1]), oracle.sgm_census(oracle.gaussian3(pairs[1][0], 0.0), oracle.gaussian3(pairs[1][1], 0.0), 64, 5, 3, 60, 8))
    e.close()


def test_stereo_eval_binary_prints_the_reference_timing_line(tmp_path):
    """stereo_eval: the 7 x 11 penalty grid of stereo_eval/main.cc:23-24, one timing line per point."""
    r = run("stereo_eval", os.path.join(G, "left_000087.png"), os.path.join(G, "right_000087.png"), tmp_path, "--no-write")
    times = [float(t) for t in re.findall(r"My SGM time spent = ([0-9.]+)s", r.stdout)]
    assert len(times) == 77 and all(t > 0 for t in times)


def test_sgm_batch_binary_shards_and_overlaps(oracle, tmp_path):
    """run_sgm_batch_census.py counterpart: file selection, per-pair outputs, slices per rank and
    worker processes per GPU (rehearsed on one GPU), checksum of checksums."""
    left, right = tmp_path / "image_0", tmp_path / "image_1"
    left.mkdir(), right.mkdir()
    want = {}
    for i in range(7):
        l, r = synth_pair(700 + i, 38 if i != 4 else 30, 100, 30)  # one pair of another size
        name = "%06d_10.png" % i
        Image.fromarray(l).save(left / name)
        Image.fromarray(r).save(right / name)
        want[name[:-4]] = oracle.sgm_census(l, r, 64, 3, 3, 60, 8)
    (left / "000000_11.png").write_bytes(b"skipped: not a _10 frame")
    x = 0
    for w in want.values():
        x ^= drivers.fnv_u16(w)

    def check(out, r):
        m = re.search(r"pairs: (\d+)  ranks: (\d+) .* checksum xor: ([0-9a-f]{16})", r.stdout)
        for st, w in want.items():
            assert np.array_equal(drivers.read_png16(str(out / "disparities" / (st + ".png"))), w), st
            assert (out / "norm_hist" / (st + ".png")).exists()
        assert (out / "interpolation").is_dir()
        return int(m.group(1)), int(m.group(2)), int(m.group(3), 16)

    o1 = tmp_path / "o1"
    assert check(o1, run("sgm_batch", "census", left, right, o1, 3, 60, "--disp-range", 64, "--batch", 3)) == (7, 1, x)
    o2 = tmp_path / "o2"
    r2 = run("sgm_batch", "census", left, right, o2, 3, 60, "--disp-range", 64, "--batch", 2, "--gpus", 2, "--share-gpu")
    assert check(o2, r2) == (7, 2, x)
    # external launcher style: one slice per process
    o3 = tmp_path / "o3"
    parts = [run("sgm_batch", "census", left, right, o3, 3, 60, "--disp-range", 64, "--rank", k, "--world", 3) for k in range(3)]
    got = [re.search(r"pairs: (\d+) .* checksum xor: ([0-9a-f]{16})", p.stdout).groups() for p in parts]
    assert [int(g[0]) for g in got] == [3, 3, 1]
    assert int(got[0][1], 16) ^ int(got[1][1], 16) ^ int(got[2][1], 16) == x
    check(o3, parts[0])


def test_sgm_batch_binary_cnn(oracle, tmp_path):
    rep = tmp_path / "representation"
    rep.mkdir()
    want = {}
    rng = np.random.default_rng(8)
    for i in range(3):
        base = rng.standard_normal((12, 64, 4)).astype(np.float32)
        fl = np.ascontiguousarray(base[:, 4:])
        fr = np.ascontiguousarray(base[:, :60] + 0.05 * rng.standard_normal((12, 60, 4)).astype(np.float32))
        pre = "%06d_10" % i
        sv.save_representation(str(rep / (pre + "_left.bin")), fl)
        sv.save_representation(str(rep / (pre + "_right.bin")), fr)
        want[pre] = oracle.sgm_cost_f32(oracle.descriptor_cost(fl, fr, 32), 3, 60, 8)
    run("sgm_batch", "cnn", tmp_path, 3, 60, "--disp-range", 32)
    for pre, w in want.items():
        assert np.mean(drivers.read_png16(str(tmp_path / "depth" / "disparities" / (pre + ".png"))) == w) >= 0.999


def test_sgm_binary_new_sgm_pipeline(oracle, tmp_path):
    """`sgm` (new_sgm/sgm_stereo_main.cc): the data-cost-file form and the descriptor form, output names
    and 8-bit preview as the reference's main derives them."""
    from test_sgmstereo_oracle import smooth_volume
    out = tmp_path / "out"
    drivers.make_output_dirs(str(out))
    C, _ = smooth_volume(3, 16, 100, 24)
    sv.save_representation(str(tmp_path / "000003_10_cost.bin"), C)
    run("sgm", tmp_path / "000003_10_cost.bin", out, 3, 40, 1)
    want = oracle.ns_compute(C, 3, 40, 1)
    got = drivers.read_png16(str(out / "disparities" / "000003_10.png"))
    assert np.array_equal(got, want)
    m8 = np.minimum(255, np.floor(want / 256.0 + 0.5)).astype(np.uint8)  # std::round: half away from zero
    assert np.array_equal(np.array(Image.open(out / "norm_hist" / "000003_10.png")), drivers.normalize_minmax_u8(m8))
    rng = np.random.default_rng(2)
    base = rng.standard_normal((8, 270, 4)).astype(np.float32)
    fl, fr = np.ascontiguousarray(base[:, 6:]), np.ascontiguousarray(base[:, :264])
    sv.save_representation(str(tmp_path / "000004_10_left.bin"), fl)
    sv.save_representation(str(tmp_path / "000004_10_right.bin"), fr)
    run("sgm", tmp_path / "000004_10_left.bin", tmp_path / "000004_10_right.bin", out, 3, 60, 1)
    want = oracle.ns_compute(oracle.ns_left_cost_desc(fl, fr, 256), 3, 60, 1)
    assert np.array_equal(drivers.read_png16(str(out / "disparities" / "000004_10.png")), want)
    # P1 >= P2 is std::invalid_argument in the reference (sgm_stereo.cc:21-24)
    r = subprocess.run([os.path.join(BIN, "sgm"), str(tmp_path / "000003_10_cost.bin"), str(out), "40", "3", "1"],
                       capture_output=True, text=True)
    assert r.returncode == 2 and "smoothness penalty" in r.stderr


def test_evaluate_stereo_binary(oracle, tmp_path):
    """evaluate_stereo result_dir: the stats files and the stats.txt line equal the Python counterpart's
    (which test_eval_gpu.py checks against the oracle)."""
    from stereo_vision_amd import evaluation
    from test_eval_gpu import make_case
    gt = tmp_path / "gt"
    for run_name in ("a", "b"):
        (tmp_path / run_name / "disparities").mkdir(parents=True)
    (gt / "disp_noc_0").mkdir(parents=True), (gt / "disp_occ_0").mkdir()
    for i in range(4):
        disp, noc, occ = make_case(60 + i, 44, 150)
        for d, arr in ((gt / "disp_noc_0", noc), (gt / "disp_occ_0", occ), (tmp_path / "a" / "disparities", disp),
                       (tmp_path / "b" / "disparities", disp)):
            drivers.write_png16(str(d / ("%06d_10.png" % i)), arr)
    run("evaluate_stereo", tmp_path / "a" / "disparities", "--gt-folder", gt, "--count", 4)
    e = evaluation.eval_stereo(str(tmp_path / "b" / "disparities"), str(gt), count=4)
    evaluation.append_stats_line(str(tmp_path / "b" / "disparities"), e)
    for name in ("stats_noc_out.txt", "stats_occ_out.txt", "stats_noc_avg.txt", "stats_occ_avg.txt"):
        assert (tmp_path / "a" / name).read_text() == (tmp_path / "b" / name).read_text(), name
    la, lb = (tmp_path / "a" / "stats.txt").read_text(), (tmp_path / "b" / "stats.txt").read_text()
    assert la.split("\t")[:5] == lb.split("\t")[:5] and la.endswith("a/disparities\n")
