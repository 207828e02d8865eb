"""One-off: the seeded random tests of tests/test_parity_gpu.py (both census formulations; float volumes) on many more
configurations than the suite keeps.  usage: exp/random_stress2.py [count] [seed]"""
import sys, os, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import test_parity_gpu as T
from oracle import oracle as o
o.build()
count = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 31337
bad = 0
for name, gen, fn in (("census both formulations", T._random_configs, T.test_seeded_random_configurations),
                      ("float volumes", T._random_float_configs, T.test_seeded_random_float_volumes)):
    for k, case in enumerate(gen(count, seed)):
        try:
            fn(o, case)
        except AssertionError:
            bad += 1
            print("FAIL", name, case, traceback.format_exc().splitlines()[-1], flush=True)
    print(name, ":", count, "configurations,", bad, "failures so far", flush=True)
sys.exit(1 if bad else 0)
