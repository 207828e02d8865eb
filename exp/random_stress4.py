"""One-off: recon::SGMStereo pipeline (tests/test_sgmstereo_gpu.py's seeded random test) and the descriptor path on many
more configurations.  usage: exp/random_stress4.py [count] [seed]"""
import sys, os, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import numpy as np
import test_sgmstereo_gpu as N
import test_drivers_gpu as G
from oracle import oracle as o
o.build()
count = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 99
bad = 0
for case in N._random_ns_configs(count, seed):
    try:
        N.test_seeded_random_configurations(o, case)
    except AssertionError:
        bad += 1
        print("FAIL sgmstereo", case, traceback.format_exc().splitlines()[-1], flush=True)
print("sgmstereo:", count, "configurations,", bad, "failures", flush=True)
rng = np.random.default_rng(seed)
for k in range(count):
    case = (int(rng.integers(1, 14)), int(rng.integers(2, 320)), int(rng.choice([1, 3, 4, 8, 30, 32, 36, 64, 70])),
            int(rng.choice([2, 16, 33, 100, 128, 129, 200, 256, 300, 512])))
    try:
        G.test_descriptor_path(o, case)
    except AssertionError:
        bad += 1
        print("FAIL descriptor", case, traceback.format_exc().splitlines()[-1], flush=True)
print("descriptor path:", count, "configurations,", bad, "failures so far", flush=True)
sys.exit(1 if bad else 0)
