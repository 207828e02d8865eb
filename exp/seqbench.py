"""Sequence-style run of the sgm_batch program on synthetic KITTI-size PNG pairs (config 3 plumbing):
python exp/seqbench.py NPAIRS -> writes the pairs under /tmp and runs sgm_batch census with and without output files."""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
from bench import synthetic_pair
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
root = "/tmp/seqbench"
for d in ("image_0", "image_1"):
    os.makedirs(os.path.join(root, d), exist_ok=True)
base = [synthetic_pair(100 + i, 376, 1241, 100) for i in range(8)]
t = time.perf_counter()
uniq = 64  # distinct files; the rest are hard links to them (same decode work per file)
for i in range(n):
    for d, k in (("image_0", 0), ("image_1", 1)):
        dst = os.path.join(root, d, "%06d_10.png" % i)
        if os.path.exists(dst):
            os.remove(dst)
        if i < uniq:
            Image.fromarray(np.roll(base[i % 8][k], i, axis=0)).save(dst, compress_level=1)
        else:
            os.link(os.path.join(root, d, "%06d_10.png" % (i % uniq)), dst)
print("wrote %d pairs in %.1f s" % (n, time.perf_counter() - t), flush=True)
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "stereo_vision_amd", "bin", "sgm_batch")
for extra in (["--no-write"], []):
    for threads in (8, 16):
        cmd = [exe, "census", root + "/image_0", root + "/image_1", root + "/out", "3", "60", "--disp-range", "128", "--window-sz", "5",
               "--batch", "32", "--threads", str(threads)] + extra
        t = time.perf_counter()
        out = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, SGMX_BATCH_TIMING="1"))
        dt = time.perf_counter() - t
        tail = [l for l in out.stdout.strip().splitlines() if "pairs" in l][-1:] or [out.stderr[-300:]]
        print("threads=%d %s: %.2f s wall, %.1f pairs/s | %s" % (threads, " ".join(extra) or "with PNG output", dt, n / dt, tail[0]), flush=True)
        for l in out.stderr.splitlines():
            if "timing" in l:
                print("   ", l, flush=True)
