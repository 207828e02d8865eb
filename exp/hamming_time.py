"""End-to-end times of the Hamming-library entry points (host buffers in and out) at na x nb signatures of ncodes bytes;
run it under rocprofv3 --kernel-trace --stats for the kernel times.  SGMX_LIB selects another build.
usage: exp/hamming_time.py [na] [nb] [ncodes] [ht]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stereo_vision_amd as sv
na, nb, nc, ht = [int(x) for x in (sys.argv[1:] + ["100000", "4096", "16", "44"][len(sys.argv) - 1:])]
rng = np.random.default_rng(5)
a = rng.integers(0, 256, size=(na, nc)).astype(np.uint8)
b = rng.integers(0, 256, size=(nb, nc)).astype(np.uint8)
e = sv.StereoSGM(sv.StereoSGMParams())
for name, fn in (("hamming_matrix", lambda: e.hamming_matrix(a, b)), ("match_hamming_count", lambda: e.match_hamming_count(a, b, ht)),
                 ("count + thres_prealloc", lambda: e.match_hamming_thres(a, b, ht)), ("crossmatch count + prealloc (na/4)", lambda: e.crossmatch_hamming(a[:na // 4], ht))):
    fn()
    t = time.perf_counter()
    for _ in range(3):
        r = fn()
    print("%-36s %9.2f ms per call" % (name, (time.perf_counter() - t) / 3 * 1e3), "" if not isinstance(r, tuple) else "(%d matches)" % len(r[1]), flush=True)
