"""CPU checks of the C++ host layer above the C ABI (stereo_vision_amd/host, include/sgmx_recon.hpp):
its PNG codec against Pillow, the 8-bit preview arithmetic against the Python drivers, the image-list
and config readers on files in the reference's formats (config_files/kitti_00_lst.xml:1-7,
config_kitti_00.txt), argv conventions and exit codes of the reference's mains, and -- without a GPU
-- the loud failure instead of a CPU fallback."""
import os
import struct
import subprocess

import numpy as np
import pytest
from PIL import Image

from stereo_vision_amd import drivers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "stereo_vision_amd", "bin")


def run(prog, *args, **kw):
    return subprocess.run([os.path.join(BIN, prog)] + [str(a) for a in args], capture_output=True, text=True, **kw)


@pytest.fixture(scope="module", autouse=True)
def _built():
    if not os.path.exists(os.path.join(BIN, "host_tool")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "stereo_vision_amd", "host")])


# ---------------------------------------------------------------- PNG codec
@pytest.mark.parametrize("mode", ["L", "I;16", "RGB", "RGBA", "LA", "P", "1"])
def test_png_reader_against_pillow(tmp_path, mode):
    rng = np.random.default_rng(len(mode))
    H, W = 37, 53
    if mode == "L":
        img = Image.fromarray(rng.integers(0, 256, (H, W)).astype(np.uint8))
    elif mode == "I;16":
        img = Image.fromarray(rng.integers(0, 65536, (H, W)).astype(np.uint16))
    elif mode == "1":
        img = Image.fromarray((rng.integers(0, 2, (H, W)) * 255).astype(np.uint8)).convert("1")
    elif mode == "P":
        img = Image.fromarray(rng.integers(0, 256, (H, W, 3)).astype(np.uint8)).quantize(16)
    else:
        nc = {"RGB": 3, "RGBA": 4, "LA": 2}[mode]
        img = Image.fromarray(rng.integers(0, 256, (H, W, nc)).astype(np.uint8), mode)
    src, dst = tmp_path / "in.png", tmp_path / "out.png"
    img.save(src)
    r = run("host_tool", "gray8", src, dst)
    assert r.returncode == 0, r.stderr
    got = np.array(Image.open(dst))
    assert got.dtype == np.uint8 and got.shape == (H, W)
    a = np.array(Image.open(src))
    if mode in ("L",):
        want = a
    elif mode == "I;16":
        want = (a >> 8).astype(np.uint8)  # 16-bit samples keep their high byte
    elif mode == "1":
        want = a.astype(np.uint8) * 255
    elif mode == "LA":
        want = a[..., 0]
    else:
        rgb = np.array(Image.open(src).convert("RGB")).astype(np.uint32)
        want = ((rgb[..., 0] * 4899 + rgb[..., 1] * 9617 + rgb[..., 2] * 1868 + 8192) >> 14).astype(np.uint8)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("name", ["tsukuba_l.png", "left_000087.png"])
def test_png_reads_the_reference_fixtures(tmp_path, name):
    dst = tmp_path / "o.png"
    assert run("host_tool", "gray8", os.path.join(ROOT, "tests", "golden", name), dst).returncode == 0
    assert np.array_equal(np.array(Image.open(dst)), np.array(Image.open(os.path.join(ROOT, "tests", "golden", name))))


@pytest.mark.parametrize("level", [0, 6, 9])
def test_png16_round_trip(tmp_path, level):
    rng = np.random.default_rng(level)
    d = (rng.integers(0, 128 * 256, (40, 61)) * (rng.random((40, 61)) > 0.2)).astype(np.uint16)
    src, dst = tmp_path / "d.png", tmp_path / "e.png"
    drivers.write_png16(str(src), d)
    assert run("host_tool", "gray16", src, dst, "--level", level).returncode == 0
    assert np.array_equal(drivers.read_png16(str(dst)), d)
    r = run("host_tool", "fnv", dst)
    assert int(r.stdout.strip(), 16) == drivers.fnv_u16(d)


@pytest.mark.parametrize("case", ["fibonacci", "constant", "two_values", "noise", "one_pixel", "filters"])
def test_png_default_writer_huffman_block(tmp_path, case):
    """The default writer's own deflate block (literals only, dynamic Huffman, lengths limited to 15 bits): Pillow
    (zlib's inflate) must read back the pixels.  "fibonacci" makes the Sub-filtered bytes follow a Fibonacci
    histogram, whose Huffman tree is deeper than 15 and takes the length-limiting path; "filters" feeds the reader
    every PNG row filter (Pillow's adaptive choice on a smooth image) before the round trip."""
    rng = np.random.default_rng(5)
    if case == "fibonacci":
        fib = [1, 1]
        while len(fib) < 24:
            fib.append(fib[-1] + fib[-2])
        diffs = np.repeat(np.arange(24, dtype=np.int64), fib)
        rng.shuffle(diffs)
        W = 12000
        diffs = np.resize(diffs, (len(diffs) // W + 1) * W).reshape(-1, W)
        img = (np.cumsum(diffs, axis=1) % 256).astype(np.uint8)
    elif case == "constant":
        img = np.full((33, 70), 7, np.uint8)
    elif case == "two_values":
        img = (rng.integers(0, 2, (64, 64)) * 200).astype(np.uint8)
    elif case == "noise":
        img = rng.integers(0, 256, (376, 1241)).astype(np.uint8)
    elif case == "one_pixel":
        img = np.array([[9]], np.uint8)
    else:
        yy, xx = np.mgrid[0:120, 0:333]
        img = ((np.sin(xx / 17.0) + np.cos(yy / 11.0)) * 60 + 128 + rng.integers(0, 3, (120, 333))).astype(np.uint8)
    src, dst = tmp_path / "in.png", tmp_path / "out.png"
    Image.fromarray(img).save(src, optimize=(case == "filters"))
    r = run("host_tool", "gray8", src, dst)
    assert r.returncode == 0, r.stderr
    assert np.array_equal(np.array(Image.open(dst)), img)
    # and 16-bit: the same bytes as big-endian sample pairs
    d16 = (img.astype(np.uint16) << 8) | np.roll(img, 1, axis=1)
    drivers.write_png16(str(tmp_path / "d.png"), d16)
    assert run("host_tool", "gray16", tmp_path / "d.png", tmp_path / "e.png").returncode == 0
    assert np.array_equal(np.array(Image.open(tmp_path / "e.png")).astype(np.uint16), d16)
    assert np.array_equal(drivers.read_png16(str(tmp_path / "e.png")), d16)


def test_png_codec_inflate_and_deflate_against_zlib():
    """host_tool zfuzz: the reader's own inflate on zlib streams of every level / strategy / memory level (stored, fixed
    and dynamic blocks, long codes, overlapping matches), damaged and truncated streams, and the writer's own deflate
    block through both inflaters."""
    r = run("host_tool", "zfuzz", "1")
    assert r.returncode == 0, r.stdout + r.stderr
    assert "bad 0" in r.stdout and "cases 1750" in r.stdout


@pytest.mark.parametrize("threads", [1, 4, 16])
def test_worker_pool(threads):
    """The batch driver's pool (host_util.hpp): concurrent submitters, urgent jobs, every item once, exceptions
    delivered to the submitting thread."""
    r = run("host_tool", "pool", threads)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "caught 5 sum ok" in r.stdout


def test_png_rejects_garbage(tmp_path):
    bad = tmp_path / "bad.png"
    bad.write_bytes(b"\x89PNG\r\n\x1a\n" + b"\0" * 40)
    r = run("host_tool", "gray8", bad, tmp_path / "o.png")
    assert r.returncode == 2 and "bad.png" in r.stderr
    good = tmp_path / "g.png"
    Image.fromarray(np.zeros((4, 4), np.uint8)).save(good)
    data = bytearray(good.read_bytes())
    data[-20] ^= 0xFF  # corrupt the image data
    bad.write_bytes(bytes(data))
    assert run("host_tool", "gray8", bad, tmp_path / "o.png").returncode == 2
    assert run("host_tool", "gray8", tmp_path / "missing.png", tmp_path / "o.png").returncode == 2


# ---------------------------------------------------------------- previews
def test_preview_arithmetic_matches_python_drivers(tmp_path):
    rng = np.random.default_rng(1)
    d = rng.integers(0, 100 * 256, (33, 47)).astype(np.uint16)
    d[::3, ::2] = 0
    d[5, 5] = 128  # 0.5 after /256: rounds to even (0)
    d[6, 6] = 384  # 1.5 -> 2
    src = tmp_path / "d.png"
    drivers.write_png16(str(src), d)
    assert run("host_tool", "div256", src, tmp_path / "a.png").returncode == 0
    a = np.array(Image.open(tmp_path / "a.png"))
    assert np.array_equal(a, drivers.to_u8_div256(d))
    assert a[5, 5] == 0 and a[6, 6] == 2
    assert run("host_tool", "norm", tmp_path / "a.png", tmp_path / "n.png").returncode == 0
    assert np.array_equal(np.array(Image.open(tmp_path / "n.png")), drivers.normalize_minmax_u8(a))
    assert run("host_tool", "equalize", tmp_path / "a.png", tmp_path / "e.png").returncode == 0
    assert np.array_equal(np.array(Image.open(tmp_path / "e.png")), drivers.equalize_hist(a))
    assert run("host_tool", "sat8", src, tmp_path / "s.png").returncode == 0
    assert np.array_equal(np.array(Image.open(tmp_path / "s.png")), np.minimum(d, 255).astype(np.uint8))
    # constant image: normalize gives zeros, equalizeHist keeps the value
    c = tmp_path / "c.png"
    Image.fromarray(np.full((5, 7), 9, np.uint8)).save(c)
    run("host_tool", "norm", c, tmp_path / "cn.png")
    run("host_tool", "equalize", c, tmp_path / "ce.png")
    assert not np.array(Image.open(tmp_path / "cn.png")).any()
    assert (np.array(Image.open(tmp_path / "ce.png")) == 9).all()


def test_yml_matrix_writer(tmp_path):
    import yaml
    d = np.arange(7 * 40, dtype=np.uint16).reshape(7, 40) * 97
    drivers.write_png16(str(tmp_path / "d.png"), d)
    assert run("host_tool", "yml", tmp_path / "d.png", tmp_path / "d.yml").returncode == 0
    text = (tmp_path / "d.yml").read_text()
    assert text.startswith("%YAML:1.0\ndisparity_matrix: !!opencv-matrix\n")
    assert max(len(l) for l in text.splitlines()) <= 80
    body = text.split("\n", 1)[1].replace("!!opencv-matrix", "")
    m = yaml.safe_load(body)["disparity_matrix"]
    assert (m["rows"], m["cols"], m["dt"]) == (7, 40, "w")
    assert m["data"] == d.reshape(-1).tolist()


# ---------------------------------------------------------------- lists / configs / descriptors
def test_image_list_reader(tmp_path):
    names = ["/image_0/%06d.png" % i if k == 0 else "/image_1/%06d.png" % i for i in range(5) for k in (0, 1)]
    x = tmp_path / "lst.xml"
    x.write_text('<?xml version="1.0"?>\n<opencv_storage>\n<imagelist>\n' + "\n".join('"%s"' % n for n in names) +
                 "\n</imagelist>\n</opencv_storage>\n")
    r = run("host_tool", "list", x)
    assert r.returncode == 0 and r.stdout.split("\n")[:-1] == names
    # unquoted entries, embedded blank in quotes, entity
    x.write_text('<?xml version="1.0"?>\n<opencv_storage>\n<!-- c -->\n<images>\na.png b.png\n"c d.png" e&amp;f.png</images>\n'
                 "</opencv_storage>\n")
    assert run("host_tool", "list", x).stdout.split("\n")[:-1] == ["a.png", "b.png", "c d.png", "e&f.png"]
    # a map is not a sequence; a missing file fails
    x.write_text("<?xml version=\"1.0\"?>\n<opencv_storage>\n<m><a>1</a></m>\n</opencv_storage>\n")
    assert run("host_tool", "list", x).returncode == 3
    assert run("host_tool", "list", tmp_path / "none.xml").returncode == 3
    assert drivers.read_string_list(str(tmp_path / "none.xml")) is None


def test_python_and_cxx_list_readers_agree(tmp_path):
    x = tmp_path / "lst.xml"
    x.write_text('<?xml version="1.0"?>\n<opencv_storage>\n<imagelist>\n"/image_0/000000.png"\n"/image_1/000000.png"\n'
                 "plain.png\n</imagelist>\n</opencv_storage>\n")
    assert drivers.read_string_list(str(x)) == run("host_tool", "list", x).stdout.split("\n")[:-1]


def test_config_file_reader(tmp_path):
    c = tmp_path / "config.txt"
    c.write_text("camera_params  = /a/b.txt\nsource_folder  = /data/00/   # trailing comment\n\n# comment\n"
                 "imglist= /x/lst.xml\nstart_num      = 0\nnot a pair\n")
    r = run("host_tool", "config", c)
    kv = dict(l.split("=", 1) for l in r.stdout.split("\n")[:-1])
    assert kv == {"camera_params": "/a/b.txt", "source_folder": "/data/00/", "imglist": "/x/lst.xml", "start_num": "0"}
    assert drivers.read_config_file(str(c)) == kv


def test_descriptor_loader(tmp_path):
    from stereo_vision_amd.sgm import save_representation
    t = np.random.default_rng(0).random((5, 7, 3), dtype=np.float32)
    p = tmp_path / "a.bin"
    save_representation(str(p), t)
    r = run("host_tool", "desc", p)
    H, W, K, s = r.stdout.split()
    assert (int(H), int(W), int(K)) == (5, 7, 3) and abs(float(s) - float(t.astype(np.float64).sum())) < 1e-4
    p.write_bytes(struct.pack("<i", 2) + b"\0" * 64)  # dims != 3: the reference does `throw 1` (sgm.h:391)
    assert run("host_tool", "desc", p).returncode == 2
    save_representation(str(p), t)
    p.write_bytes(p.read_bytes()[:-8])  # truncated
    assert run("host_tool", "desc", p).returncode == 2


def test_shard_range_matches_python():
    for n, w in [(4541, 8), (10, 3), (2, 4), (0, 2)]:
        for r in range(w):
            lo, hi = map(int, run("host_tool", "shard", n, r, w).stdout.split())
            assert (lo, hi) == drivers.shard_range(n, r, w)


# ---------------------------------------------------------------- argv conventions, no-GPU behaviour
def test_usage_errors_exit_1_like_the_reference():
    for prog, n in [("sgm_single", 3), ("sgm_cnn", 5), ("stereo_eval", 1), ("sgm_batch", 0)]:
        r = run(prog, *(["x"] * n))
        assert r.returncode == 1 and "usage" in r.stderr
    r = run("sgm", "x", "y")  # new_sgm/sgm_stereo_main.cc:152-161 prints the usage and returns 0
    assert r.returncode == 0 and "usage: ./sgm data_cost_path out_folder P1 P2 consistency_threshold" in r.stderr
    r = run("sgm_single", "l", "r", "o", "three", "60")
    assert r.returncode == 2 and "integer" in r.stderr


def test_dense_stereo_config_handling(tmp_path):
    r = run("dense_stereo", "-c", tmp_path / "nope.txt")
    assert r.returncode == 0 and "can not open config file" in r.stdout  # dense_stereo/main.cc:155-160
    c = tmp_path / "config.txt"
    c.write_text("source_folder = /s/\ndisp_folder = /d/\n")
    r = run("dense_stereo", "-c", c)
    assert r.returncode == 255 and "no xml image list given" in r.stdout  # return -1
    c.write_text("source_folder = /s/\ndisp_folder = /d/\nimglist = %s\n" % (tmp_path / "missing.xml"))
    r = run("dense_stereo", "-c", c)
    assert r.returncode == 255 and "string list is empty" in r.stdout


def test_cli_fails_loudly_without_a_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    g = os.path.join(ROOT, "tests", "golden")
    r = run("sgm_single", os.path.join(g, "tsukuba_l.png"), os.path.join(g, "tsukuba_r.png"), tmp_path, 3, 60)
    assert r.returncode == 2 and "no CPU fallback" in r.stderr
    assert not list(tmp_path.iterdir())


def test_adapter_refuses_non_finite_and_fractional_penalties(tmp_path):
    """recon::SGMStereo::SetSmoothnessCostParameters (sgm_stereo.cc:16-28) in include/sgmx_recon.hpp: NaN, infinities,
    values beyond int and fractions throw std::invalid_argument before any cast (no GPU call is made)."""
    src = tmp_path / "p.cc"
    src.write_text('#include <cmath>\n#include <cstdio>\n#include "sgmx_recon.hpp"\n'
                   "int main() {\n"
                   "  const double bad[][2] = {{3, INFINITY}, {NAN, 40}, {3, NAN}, {2.5, 40}, {3, 1e12}, {-1, 4}, {40, 3}};\n"
                   "  int thrown = 0;\n"
                   "  for (auto &b : bad) { recon::SGMStereo t; try { t.SetSmoothnessCostParameters(b[0], b[1]); }\n"
                   "    catch (const std::invalid_argument &) { thrown++; } }\n"
                   "  recon::SGMStereo ok; ok.SetSmoothnessCostParameters(3, 40);\n"
                   '  std::printf("%d\\n", thrown); return thrown == 7 ? 0 : 1; }\n')
    exe = tmp_path / "p"
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", os.path.join(ROOT, "stereo_vision_amd"), "-lsgmx", "-Wl,-rpath," + os.path.join(ROOT, "stereo_vision_amd")])
    assert subprocess.run([str(exe)], capture_output=True, text=True).stdout.strip() == "7"


def test_adapter_header_compiles_standalone(tmp_path):
    """include/sgmx_recon.hpp is self-contained C++11 and keeps the reference's signatures."""
    src = tmp_path / "t.cc"
    src.write_text('#include "sgmx_recon.hpp"\n'
                   "void f(sgmx::Mat& l, sgmx::Mat& r, sgmx::Mat& d) {\n"
                   "  recon::StereoSGMParams p; p.disp_range = 128; p.window_sz = 5; p.penalty1 = 3; p.penalty2 = 60;\n"
                   "  recon::StereoSGM sgm(p); sgm.compute(l, r, d);\n"
                   "  recon::SGMParams q; q.disp_range = 255; q.penalty1 = 3; q.penalty2 = 60;\n"
                   '  recon::SGM s(q); sgmx::Mat m = s.Compute("a", "b"); (void)m;\n'
                   '  recon::SGMStereo t; t.SetSmoothnessCostParameters(3, 40); t.SetConsistencyThreshold(1);\n'
                   '  t.Compute("cost.bin", &d); t.Compute("l.bin", "r.bin", &d); }\n')
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)])
