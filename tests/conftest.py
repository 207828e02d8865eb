import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_gray(name):
    from PIL import Image
    return np.array(Image.open(os.path.join(GOLDEN, name)).convert("L"))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o


@pytest.fixture(scope="session")
def tsukuba():
    return load_gray("tsukuba_l.png"), load_gray("tsukuba_r.png")


@pytest.fixture(scope="session")
def kitti():
    return load_gray("left_000087.png"), load_gray("right_000087.png")


def synth_pair(seed, H, W, max_disp, noise=2.0):
    """Seeded synthetic rectified pair: box-blurred uniform noise as
    texture; the right view is the left shifted per row-band by a piecewise-constant
    disparity, plus Gaussian noise."""
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, size=(H + 4, W + max_disp + 4)).astype(np.float32)
    k = np.ones(3, np.float32) / 3
    base = np.apply_along_axis(lambda r: np.convolve(r, k, mode="same"), 1, base)
    base = np.apply_along_axis(lambda c: np.convolve(c, k, mode="same"), 0, base)
    left = base[2:H + 2, 2:W + 2]
    right = np.empty_like(left)
    nb = max(1, H // 16)
    for b in range(0, H, nb):
        d = int(rng.integers(0, max(1, max_disp)))
        right[b:b + nb] = base[2 + b:2 + min(H, b + nb), 2 + d:2 + d + W]
    right = right + rng.normal(0, noise, size=right.shape)
    # apply_along_axis leaves Fortran-ordered arrays behind; device uploads need dense C order
    return (np.ascontiguousarray(np.clip(left, 0, 255).astype(np.uint8)),
            np.ascontiguousarray(np.clip(right, 0, 255).astype(np.uint8)))
