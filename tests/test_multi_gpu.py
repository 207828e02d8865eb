"""The multi-GPU legs on the one-GPU box (stereo pairs are independent, so ranks share nothing but
a barrier and the max of their timings).  RCCL refuses two ranks on one device ("Duplicate GPU detected", measured on
the box), so what can run here is: the RCCL leg of bench.py with one rank (process-group init, barrier,
MAX all-reduce on a device tensor), the two-rank leg over gloo with both ranks on cuda:0, and sgm_batch's worker
processes per GPU at KITTI size.  Every launcher is a fresh child process that touches the GPU only in its ranks."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
from PIL import Image

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "stereo_vision_amd", "bin")


def _torchrun(nproc, port, *bench_args):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py")] + [str(a) for a in bench_args]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=110, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_bench_rccl_leg_with_one_rank():
    out = _torchrun(1, 29533, "--gpus", 1, "--backend", "nccl", "--dist-always", "--steps", 3, "--warmup", 1, "--batch", 4,
                    "--cpu-frames", 0, "--no-secondary")
    assert out["n_gpus"] == 1 and out["config"]["dist_backend"] == "nccl" and out["steps"] == 3
    assert out["value"] > 0 and 0 < out["roofline"]["frac"] <= 1


def test_bench_two_ranks_over_gloo_on_one_gpu():
    out = _torchrun(2, 29534, "--gpus", 2, "--backend", "gloo", "--share-gpu", "--steps", 3, "--warmup", 1, "--batch", 4,
                    "--cpu-frames", 0)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["dist_backend"] == "gloo"
    # whole-job figure: both ranks' pairs over the slower rank's time
    assert abs(out["value"] - 2 * 4 * 3 / (out["ms_per_step"] * 3e-3)) < 1e-6 * out["value"] + 0.1
    # several ranks: `steps` stays what was asked for, the timed region is timed_blocks whole blocks of it (2 s floor),
    # and every rank's own rate is in the line
    assert out["steps"] == 3 and out["timed_blocks"] >= 1 and len(out["per_rank_fps"]) == 2
    assert all(f >= out["value"] / 2 * 0.999 for f in out["per_rank_fps"])  # no rank is slower than the job's per-rank share


def test_sgm_batch_two_workers_equal_one_at_kitti_size(tmp_path):
    """`sgm_batch census --gpus 2 --share-gpu` (one worker process per "GPU", forked before any HIP call) against the
    single-rank run on 6 KITTI-size pairs: same pair count, same XOR of the per-pair FNV checksums
    (scripts/run_sgm_batch_census.py:27-34: pairs are independent)."""
    sys.path.insert(0, ROOT)
    from bench import synthetic_pair
    left, right = tmp_path / "image_0", tmp_path / "image_1"
    left.mkdir(), right.mkdir()
    for i in range(6):
        l, r = synthetic_pair(5000 + i, 376, 1241, 100)
        Image.fromarray(l).save(left / ("%06d_10.png" % i))
        Image.fromarray(r).save(right / ("%06d_10.png" % i))

    def go(out, *extra):
        r = subprocess.run([os.path.join(BIN, "sgm_batch"), "census", str(left), str(right), str(out), "3", "60", "--disp-range",
                            "128", "--window-sz", "5", "--no-write"] + list(extra), capture_output=True, text=True, timeout=110)
        assert r.returncode == 0, r.stderr[-2000:]
        m = re.search(r"pairs: (\d+)  ranks: (\d+) .* checksum xor: ([0-9a-f]{16})", r.stdout)
        return int(m.group(1)), int(m.group(2)), m.group(3)

    one = go(tmp_path / "o1", "--batch", "4")
    two = go(tmp_path / "o2", "--batch", "2", "--gpus", "2", "--share-gpu")
    assert one[:2] == (6, 1) and two[:2] == (6, 2) and one[2] == two[2]
