"""Host logic of the callers (stereo_vision_amd/drivers.py): file selection rules of the
reference batch scripts, output naming, sharding, and the N > 1 statistics path over gloo with
world_size 2.  No GPU compute here."""
import os

import numpy as np
import pytest

from stereo_vision_amd import drivers


def test_shard_range_is_a_contiguous_partition():
    for n in (0, 1, 7, 8, 9, 4541):
        for world in (1, 2, 3, 4, 8):
            got = [drivers.shard_range(n, r, world) for r in range(world)]
            assert got[0][0] == 0 and got[-1][1] == n
            for (a, b), (c, d) in zip(got, got[1:]):
                assert b == c and a <= b and c <= d
            per = -(-n // world)
            assert all(hi - lo <= per for lo, hi in got)
    assert drivers.shard_range(4541, 7, 8) == (3976, 4541)  # KITTI odometry sequence 00, last of 8 GPUs
    with pytest.raises(ValueError):
        drivers.shard_range(10, 2, 2)


def test_census_file_selection(tmp_path):
    """run_sgm_batch_census.py:25-31: names containing "_10.png", right image of the same name."""
    left, right = tmp_path / "image_0", tmp_path / "image_1"
    left.mkdir(), right.mkdir()
    for name in ("000001_10.png", "000001_11.png", "000000_10.png", "000002_10.png.txt", "notes.txt"):
        (left / name).write_bytes(b"x")
    (left / "000009_10.png").mkdir()  # directories are skipped (isfile)
    pairs = drivers.list_census_pairs(str(left), str(right))
    assert [os.path.basename(l) for l, _ in pairs] == ["000000_10.png", "000001_10.png", "000002_10.png.txt"]
    assert all(os.path.dirname(r) == str(right) and os.path.basename(l) == os.path.basename(r) for l, r in pairs)


def test_cnn_file_selection(tmp_path):
    """run_sgm_batch_cnn.py:33-41: "left" in name, prefix = name[:9], right = prefix + "_right.bin"."""
    for name in ("000000_10_left.bin", "000000_10_right.bin", "000003_10_left.bin", "readme"):
        (tmp_path / name).write_bytes(b"x")
    got = drivers.list_cnn_pairs(str(tmp_path))
    assert [(os.path.basename(a), os.path.basename(b), p) for a, b, p in got] == [
        ("000000_10_left.bin", "000000_10_right.bin", "000000_10"),
        ("000003_10_left.bin", "000003_10_right.bin", "000003_10")]


def test_grid_search_naming():
    """run_sgm_batch_grid_search.py:41,49-53: "%06d" frame numbers, stop_num inclusive, one folder per (P1, P2)."""
    pairs = drivers.list_numbered_pairs("/data/training/", 0, 193)
    assert len(pairs) == 194
    assert pairs[0] == ("/data/training//image_0/000000_10.png", "/data/training//image_1/000000_10.png")
    assert pairs[-1][0].endswith("/image_0/000193_10.png")
    assert [os.path.basename(l) for l, _ in drivers.list_numbered_pairs("d", 3, 9, 3)] == [
        "000003_10.png", "000006_10.png", "000009_10.png"]
    assert drivers.grid_search_folder("/res/experiment", 4, 55) == "/res/experiment_4_55/data/"


def test_same_name_file_selection(tmp_path):
    """run_sgm_batch.py:51-60: every file of left/, partner of the same name in right/."""
    (tmp_path / "left").mkdir(), (tmp_path / "right").mkdir()
    for name in ("b.png", "a.png"):
        (tmp_path / "left" / name).write_bytes(b"x")
    (tmp_path / "left" / "sub").mkdir()
    got = drivers.list_same_name_pairs(str(tmp_path))
    assert [(os.path.basename(l), os.path.basename(r)) for l, r in got] == [("a.png", "a.png"), ("b.png", "b.png")]
    assert all("/left/" in l and "/right/" in r for l, r in got)


def test_calib_and_groundtruth_helpers(tmp_path):
    """ReadCalibKitti (core/format_helper.cc:19-32,70-92) and FixKittiGroundtruth (create_rgbd_dataset/main.cc:29-47)."""
    rows = ["P0: 721.5 0 609.5 0 0 721.5 172.8 0 0 0 1 0", "P1: 721.5 0 609.5 -387.5744 0 721.5 172.8 0 0 0 1 0",
            "P2: 700.0 0 600.0 44.8 0 710.0 170.0 0.2 0 0 1 0.003", "P3: 700.0 0 600.0 -330.2 0 710.0 170.0 0.2 0 0 1 0.003"]
    f = tmp_path / "um_000000.txt"
    f.write_text("\n".join(rows) + "\n")
    grey, color = drivers.read_calib_kitti(str(f))
    assert grey[:4] == [721.5, 721.5, 609.5, 172.8] and abs(grey[4] - 387.5744 / 721.5) < 1e-12
    assert color[:4] == [700.0, 710.0, 600.0, 170.0] and abs(color[4] - 375.0 / 700.0) < 1e-12
    rgb = np.zeros((2, 3, 3), np.uint8)
    rgb[0, 0] = (255, 0, 0)
    rgb[0, 1] = (255, 0, 255)
    rgb[1, 2] = (254, 0, 0)
    out = drivers.fix_kitti_groundtruth(rgb)
    assert tuple(out[0, 0]) == (0, 0, 255) and all(tuple(out[y, x]) == (0, 255, 0) for y, x in ((0, 1), (1, 2), (1, 1)))


def test_output_dirs_and_names(tmp_path, capsys):
    out = str(tmp_path / "out")
    assert drivers.make_output_dirs(out) is True
    for sub in ("disparities", "norm_hist", "interpolation"):
        assert os.path.isdir(os.path.join(out, sub))
    assert drivers.make_output_dirs(out) is False and "WARNING: path exists" in capsys.readouterr().out
    assert drivers.stem("/a/b.c/000012_10.png") == "000012_10" and drivers.stem("noext") == "noext"
    disp = (np.arange(12 * 20, dtype=np.uint32).reshape(12, 20) * 257 % 65536).astype(np.uint16)
    drivers.write_outputs(out, "p", disp)
    assert np.array_equal(drivers.read_png16(os.path.join(out, "disparities", "p.png")), disp)
    prev = drivers.read_gray(os.path.join(out, "norm_hist", "p.png"))
    assert prev.shape == disp.shape and prev.min() == 0 and prev.max() == 255


def test_preview_arithmetic():
    d = np.array([[0, 127, 128, 129, 384, 65535]], np.uint16)
    assert drivers.to_u8_div256(d).tolist() == [[0, 0, 0, 1, 2, 255]]  # cvRound: half to even
    assert drivers.normalize_minmax_u8(np.array([[10, 20, 30]], np.uint8)).tolist() == [[0, 128, 255]]
    assert drivers.normalize_minmax_u8(np.full((2, 2), 7, np.uint8)).max() == 0


def test_fnv_matches_the_oracle(oracle):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 65536, size=(9, 11)).astype(np.uint16)
    assert drivers.fnv_u16(img) == oracle.fnv_u16(img)


def test_descriptor_file_round_trip(tmp_path):
    import stereo_vision_amd as sv
    t = np.random.default_rng(0).random((5, 7, 3), dtype=np.float32)
    p = str(tmp_path / "000000_10_left.bin")
    sv.save_representation(p, t)
    assert os.path.getsize(p) == 4 + 24 + t.size * 4  # int32 dims, uint64 size[3], float32 data (sgm.h:389-409)
    assert np.array_equal(sv.load_representation(p), t)
    with open(p, "r+b") as f:
        f.write(b"\x02\x00\x00\x00")
    with pytest.raises(ValueError):
        sv.load_representation(p)  # the reference throws (sgm.h:391)


def _gloo_worker(rank, world, port, q):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    lo, hi = drivers.shard_range(4541, rank, world)
    st = drivers.gather_stats(hi - lo, 1.0 + rank, checksum_xor=0xF00DFACE00000000 | (rank + 1))
    q.put((rank, lo, hi, st))
    dist.destroy_process_group()


def test_two_rank_sharding_and_stats_over_gloo():
    """world_size 2 on CPU: the ranks' slices tile the sequence, frames add up, the time is the
    max over ranks and the 64-bit checksums XOR together."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, st0), (r1, lo1, hi1, st1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 2271, 2271, 4541)
    for st in (st0, st1):
        assert st["frames"] == 4541 and st["seconds"] == 2.0 and st["world"] == 2
        assert st["checksum_xor"] == (0xF00DFACE00000001 ^ 0xF00DFACE00000002)
