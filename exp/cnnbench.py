"""Descriptor-file run of sgm_batch (run_sgm_batch_cnn.py counterpart) at KITTI size: python exp/cnnbench.py NPAIRS K
writes NPAIRS pairs of [376][1241][K] float32 descriptor files under /tmp and times `sgm_batch cnn` on them, and
`sgm_cnn` once per pair (the reference scripts' call pattern: a process per pair)."""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stereo_vision_amd as sv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
K = int(sys.argv[2]) if len(sys.argv) > 2 else 32
root = "/tmp/cnnbench"
rep = os.path.join(root, "representation")
os.makedirs(rep, exist_ok=True)
rng = np.random.default_rng(3)
base = rng.standard_normal((376, 1241 + 64, K)).astype(np.float32)
t = time.perf_counter()
for i in range(n):
    pre = "%06d_10" % i
    sv.save_representation(os.path.join(rep, pre + "_left.bin"), np.ascontiguousarray(base[:, 64:]))
    sv.save_representation(os.path.join(rep, pre + "_right.bin"), np.ascontiguousarray(base[:, 64 - (i % 8):1241 + 64 - (i % 8)]))
print("wrote %d pairs (%.0f MB per file) in %.1f s" % (n, 376 * 1241 * K * 4 / 1e6, time.perf_counter() - t), flush=True)
bindir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "stereo_vision_amd", "bin")
for extra in (["--no-write"], []):
    t = time.perf_counter()
    out = subprocess.run([os.path.join(bindir, "sgm_batch"), "cnn", root, "3", "60", "--disp-range", "128"] + extra, capture_output=True, text=True)
    dt = time.perf_counter() - t
    print("sgm_batch cnn %s: %.2f s wall, %.2f pairs/s | %s" % (" ".join(extra) or "with PNG output", dt, n / dt,
                                                             ([l for l in out.stdout.splitlines() if "pairs" in l] or [out.stderr[-300:]])[-1]), flush=True)
t = time.perf_counter()
m = min(n, 3)
for i in range(m):
    pre = "%06d_10" % i
    subprocess.run([os.path.join(bindir, "sgm_cnn"), os.path.join(rep, pre + "_left.bin"), os.path.join(rep, pre + "_right.bin"),
                    os.path.join(root, "single"), pre, "3", "60"], capture_output=True, text=True)
dt = time.perf_counter() - t
print("sgm_cnn, one process per pair (D = 255 as its main hard-codes): %.2f s per pair" % (dt / m), flush=True)
