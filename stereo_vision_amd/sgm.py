"""Host-side mirror of the reference's SGM interface on top of the C ABI.

    reference (C++)                                   here
    recon::StereoSGMParams  (stereo_sgm.h:15-21)      StereoSGMParams
    recon::StereoSGM(params).compute(L, R, disp)      StereoSGM(params).compute(L, R) -> disp
    recon::SGMParams        (sgm.h:18-23)             SGMParams
    recon::SGM(params).Compute(left_path, right_path) SGM(params).Compute(left_path, right_path)
                                                      SGM(params).compute_cost(volume)

Same field names and argument meaning; errors are exceptions (SgmxError) where the
reference asserts / throws / has undefined behaviour.  All arithmetic happens in
libsgmx.so on the GPU; numpy is only used to hold host buffers.
"""
import ctypes as C
import math
import struct
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import SgmxError, SgmxParams


@dataclass
class StereoSGMParams:
    disp_range: int = 128
    window_sz: int = 5
    penalty1: int = 3
    penalty2: int = 60
    # extensions: not in the reference struct
    num_paths: int = 8
    device: int = 0
    max_batch: int = 1
    flags: int = 0  # _lib.FLAG_MATERIALIZE keeps the C and S volumes in HBM like the reference


SGMParams = StereoSGMParams  # sgm.h:18-23 has the same four ints


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Engine:
    """Owns one sgmx context."""

    def __init__(self, params):
        self.lib = _lib.load()
        self.params = params
        p = SgmxParams(params.disp_range, params.window_sz, params.penalty1, params.penalty2, params.num_paths,
                       params.device, params.max_batch, params.flags)
        h = C.c_void_p()
        rc = self.lib.sgmx_create(C.byref(p), C.byref(h))
        if rc != 0:
            raise SgmxError(rc, self.lib.sgmx_last_error(None).decode())
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.sgmx_destroy(self.h)
            self.h = None

    __del__ = close

    def check(self, rc):
        if rc != 0:
            raise SgmxError(rc, self.lib.sgmx_last_error(self.h).decode())

    @property
    def margin(self):
        return (self.params.window_sz - 1) // 2

    # ---- measurement ----
    def set_profiling(self, on):
        """False / 0: off; True / 1: two HIP events around every kernel launch; 2: the direction sweeps of the volume-free
        census path only, with shared boundary events (the light mode bench.py times its headline under)."""
        self.check(self.lib.sgmx_set_profiling(self.h, int(on)))

    def kernel_stats(self):
        out = {}
        for k, name in enumerate(_lib.KERNEL_CLASS_NAMES):
            ms, n = C.c_double(), C.c_long()
            self.check(self.lib.sgmx_kernel_stats(self.h, k, C.byref(ms), C.byref(n)))
            out[name] = (ms.value, n.value)
        return out

    def direction_stats(self):
        """[(total ms, launches)] of the direction sweeps since the last call, in schedule order."""
        out = []
        for k in range(self.params.num_paths):
            ms, n = C.c_double(), C.c_long()
            self.check(self.lib.sgmx_direction_stats(self.h, k, C.byref(ms), C.byref(n)))
            out.append((ms.value, n.value))
        return out

    def last_launch_group(self):
        """(first_pair, count): the pairs of the last compute call that the stage taps can reach -- `index` of
        stages() / data_cost_image() counts from first_pair.  A call runs its pairs in launch groups of at most
        max_batch pairs (the volume-free census path: also at most 32 pairs and 208 MB of census images), and the
        scratch holds the last group only."""
        first, count = C.c_int(0), C.c_int(0)
        self.check(self.lib.sgmx_last_launch_group(self.h, C.byref(first), C.byref(count)))
        return first.value, count.value

    def algorithmic_bytes(self, W, H):
        agg = C.c_double()
        tot = self.lib.sgmx_algorithmic_bytes_census(self.h, W, H, C.byref(agg))
        return tot, agg.value


def _gaussian_blur3(engine, img, sigma):
    img = np.asarray(img)
    if img.dtype != np.uint8 or img.ndim != 2:
        raise ValueError("gaussian_blur3 wants a single-channel uint8 image")
    if img.strides[1] != 1:
        img = np.ascontiguousarray(img)
    H, W = img.shape
    out = np.empty((H, W), np.uint8)
    engine.check(engine.lib.sgmx_gaussian_blur3_u8(engine.h, _ptr(img), W, H, img.strides[0], float(sigma), _ptr(out), W))
    return out


class StereoSGM(Engine):
    """recon::StereoSGM (stereo_sgm.h:53-57), COST_CENSUS build."""

    def set_preblur(self, sigma=0.7, enabled=True):
        """cv::GaussianBlur(img, img, Size(3, 3), sigma) on both images of every pair, fused in front of the census
        transform on the device (the callers' pre-filter: dense_stereo/main.cc:83-85, stereo_eval/main.cc:36-38)."""
        self.check(self.lib.sgmx_set_preblur3(self.h, int(enabled), float(sigma)))

    def compute(self, left_img, right_img):
        """left/right: (H, W) uint8 (any row pitch) -> (H, W) uint16, 256 * disparity."""
        left_img, right_img = np.asarray(left_img), np.asarray(right_img)
        if left_img.dtype != np.uint8 or right_img.dtype != np.uint8 or left_img.ndim != 2:
            raise ValueError("StereoSGM.compute wants single-channel uint8 images")
        if left_img.shape != right_img.shape:
            raise ValueError("left and right image sizes differ")
        if left_img.strides[1] != 1:
            left_img = np.ascontiguousarray(left_img)
        if right_img.strides[1] != 1:
            right_img = np.ascontiguousarray(right_img)
        H, W = left_img.shape
        disp = np.empty((H, W), np.uint16)
        self.check(self.lib.sgmx_compute_census_u8(self.h, _ptr(left_img), _ptr(right_img), W, H,
                                                   left_img.strides[0], right_img.strides[0], _ptr(disp), 2 * W))
        return disp

    def gaussian_blur3(self, img, sigma=0.7):
        """cv::GaussianBlur(img, img, Size(3,3), sigma), the pre-filter of dense_stereo/main.cc:83-85."""
        return _gaussian_blur3(self, img, sigma)

    def depth_image(self, disp, focal, baseline):
        """CalculateDepthImage (create_rgbd_dataset/main.cc:12-27): 16-bit disparity -> depth in cm."""
        disp = np.ascontiguousarray(disp, np.uint16)
        H, W = disp.shape
        out = np.empty((H, W), np.uint16)
        self.check(self.lib.sgmx_depth_from_disparity_u16(self.h, _ptr(disp), W, H, float(focal), float(baseline), _ptr(out)))
        return out

    def compute_batch(self, lefts, rights):
        """(n, H, W) uint8 x2 -> (n, H, W) uint16; what the batch scripts do pair by pair."""
        lefts = np.ascontiguousarray(lefts, np.uint8)
        rights = np.ascontiguousarray(rights, np.uint8)
        if lefts.shape != rights.shape or lefts.ndim != 3:
            raise ValueError("want two (n, H, W) uint8 stacks of equal shape")
        n, H, W = lefts.shape
        disp = np.empty((n, H, W), np.uint16)
        self.check(self.lib.sgmx_compute_census_u8_batch(self.h, n, _ptr(lefts), _ptr(rights), W, H, _ptr(disp)))
        return disp

    def compute_device(self, d_left, d_right, d_disp, n, W, H, stream=0):
        """Device-resident batch: raw device pointers (ints), enqueue only."""
        self.check(self.lib.sgmx_compute_census_u8_dev(self.h, n, d_left, d_right, W, H, d_disp, stream))

    def compute_device_noalloc(self, d_left, d_right, d_disp, n, W, H, stream=0):
        """The same through the entry point that allocates nothing (static per-GPU scratch, size limits:
        include/sgmx.h)."""
        self.check(self.lib.sgmx_compute_census_u8_dev_noalloc(self.h, n, d_left, d_right, W, H, d_disp, stream))

    # ---- stage taps ----
    def census(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        H, W = img.shape
        m = self.margin
        out = np.empty((H - 2 * m, W - 2 * m), np.uint32)
        self.check(self.lib.sgmx_census_u32(self.h, _ptr(img), W, H, W, _ptr(out)))
        return out

    def hamming_volume(self, census_left, census_right):
        cl = np.ascontiguousarray(census_left, np.uint32)
        cr = np.ascontiguousarray(census_right, np.uint32)
        h, w = cl.shape
        out = np.empty((h, w, self.params.disp_range), np.uint8)
        self.check(self.lib.sgmx_hamming_volume_u8(self.h, _ptr(cl), _ptr(cr), w, h, _ptr(out)))
        return out

    def hamming_matrix(self, a, b):
        """compute_hamming (core/hamming.h:67-68): a (na, ncodes), b (nb, ncodes) -> (nb, na) uint16."""
        a = np.ascontiguousarray(a, np.uint8)
        b = np.ascontiguousarray(b, np.uint8)
        out = np.empty((b.shape[0], a.shape[0]), np.uint16)
        self.check(self.lib.sgmx_hamming_matrix_u16(self.h, _ptr(out), _ptr(a), _ptr(b), a.shape[0], b.shape[0],
                                                    a.shape[1]))
        return out

    def match_hamming_count(self, bs1, bs2, ht):
        """match_hamming_count (core/hamming.h:75-76): number of pairs (i, j) with distance <= ht."""
        a = np.ascontiguousarray(bs1, np.uint8)
        b = np.ascontiguousarray(bs2, np.uint8)
        n = C.c_size_t(0)
        self.check(self.lib.sgmx_match_hamming_count(self.h, _ptr(a), _ptr(b), a.shape[0], b.shape[0], int(ht),
                                                     a.shape[1], C.byref(n)))
        return n.value

    def crossmatch_hamming_count(self, dbs, ht):
        """crossmatch_hamming_count (core/hamming.h:78-79): pairs j > i of one set."""
        a = np.ascontiguousarray(dbs, np.uint8)
        n = C.c_size_t(0)
        self.check(self.lib.sgmx_crossmatch_hamming_count(self.h, _ptr(a), a.shape[0], int(ht), a.shape[1], C.byref(n)))
        return n.value

    def match_hamming_thres(self, bs1, bs2, ht):
        """match_hamming_count + match_hamming_thres_prealloc (core/hamming.h:75-76, 90-92), the
        reference's count / allocate / fill usage: returns idx (n, 2) int32 rows (i, j) in the
        reference's loop order and the distances (n,) uint16."""
        a = np.ascontiguousarray(bs1, np.uint8)
        b = np.ascontiguousarray(bs2, np.uint8)
        n = self.match_hamming_count(a, b, ht)
        idx = np.empty((n, 2), np.int32)
        hams = np.empty((n,), np.uint16)
        got = C.c_size_t(0)
        self.check(self.lib.sgmx_match_hamming_thres_prealloc(self.h, _ptr(a), _ptr(b), a.shape[0], b.shape[0], int(ht),
                                                              a.shape[1], _ptr(idx), _ptr(hams), n, C.byref(got)))
        assert got.value == n
        return idx, hams

    def crossmatch_hamming(self, dbs, ht):
        """crossmatch_hamming_count + crossmatch_hamming_prealloc (core/hamming.h:78-79, 103-104)."""
        a = np.ascontiguousarray(dbs, np.uint8)
        n = self.crossmatch_hamming_count(a, ht)
        idx = np.empty((n, 2), np.int32)
        hams = np.empty((n,), np.uint16)
        got = C.c_size_t(0)
        self.check(self.lib.sgmx_crossmatch_hamming_prealloc(self.h, _ptr(a), a.shape[0], int(ht), a.shape[1], _ptr(idx),
                                                             _ptr(hams), n, C.byref(got)))
        assert got.value == n
        return idx, hams

    def _take_records(self, hm, n):
        """Copy n hammatch_t records (core/hamming.h:52-56) out of a malloc'ed array and release it."""
        rec = np.dtype([("qid", np.int32), ("bid", np.int32), ("score", np.uint16), ("_pad", np.uint16)])
        try:
            out = np.ctypeslib.as_array(C.cast(hm, C.POINTER(C.c_uint8)), (n * 12,)).view(rec).copy() if n else np.empty(0, rec)
        finally:
            self.lib.sgmx_free(hm)
        return np.stack([out["qid"], out["bid"]], 1).astype(np.int32), out["score"].copy()

    def match_hamming_thres_alloc(self, bs1, bs2, ht, bufsize=16):
        """match_hamming_thres (core/hamming.h:84-86): the library allocates the hammatch_t array."""
        a = np.ascontiguousarray(bs1, np.uint8)
        b = np.ascontiguousarray(bs2, np.uint8)
        hm, n = C.c_void_p(), C.c_size_t(0)
        rc = self.lib.sgmx_match_hamming_thres(self.h, _ptr(a), _ptr(b), a.shape[0], b.shape[0], int(ht), a.shape[1],
                                               int(bufsize), C.byref(hm), C.byref(n))
        if rc:
            self.lib.sgmx_free(hm)
            self.check(rc)
        return self._take_records(hm, n.value)

    def crossmatch_hamming_alloc(self, dbs, ht, bufsize=16):
        """crossmatch_hamming (core/hamming.h:95-96)."""
        a = np.ascontiguousarray(dbs, np.uint8)
        hm, n = C.c_void_p(), C.c_size_t(0)
        rc = self.lib.sgmx_crossmatch_hamming(self.h, _ptr(a), a.shape[0], int(ht), a.shape[1], int(bufsize), C.byref(hm),
                                              C.byref(n))
        if rc:
            self.lib.sgmx_free(hm)
            self.check(rc)
        return self._take_records(hm, n.value)

    def hamming_matrix_thread(self, a, b):
        """compute_hamming_thread (core/hamming.h:108-109): the same matrix as compute_hamming."""
        a = np.ascontiguousarray(a, np.uint8)
        b = np.ascontiguousarray(b, np.uint8)
        out = np.empty((b.shape[0], a.shape[0]), np.uint16)
        self.check(self.lib.sgmx_compute_hamming_thread(self.h, _ptr(out), _ptr(a), _ptr(b), a.shape[0], b.shape[0], a.shape[1]))
        return out

    def match_hamming_thres_nt(self, bs1, bs2, ht, nt=4):
        """match_hamming_thres_nt (core/hamming.h:111-112): matches block pair by block pair, block-local indices."""
        a = np.ascontiguousarray(bs1, np.uint8)
        b = np.ascontiguousarray(bs2, np.uint8)
        keys, ham, n = C.c_void_p(), C.c_void_p(), C.c_size_t(0)
        self.check(self.lib.sgmx_match_hamming_thres_nt(self.h, _ptr(a), _ptr(b), a.shape[0], b.shape[0], int(ht), a.shape[1],
                                                        int(nt), C.byref(keys), C.byref(ham), C.byref(n)))
        try:
            k = np.ctypeslib.as_array(C.cast(keys, C.POINTER(C.c_int32)), (max(n.value, 1), 2))[:n.value].copy()
            h = np.ctypeslib.as_array(C.cast(ham, C.POINTER(C.c_uint16)), (max(n.value, 1),))[:n.value].copy()
        finally:
            self.lib.sgmx_free(keys)
            self.lib.sgmx_free(ham)
        return k, h

    def stages(self, H, W, index=0):
        """Stages of pair `index` of the LAST LAUNCH GROUP of the last call (not of the call: see
        last_launch_group(), which says which pair of the call that is)."""
        m = self.margin
        h, w, D = H - 2 * m, W - 2 * m, self.params.disp_range
        Cv = np.empty((h, w, D), np.uint8)
        S = np.empty((h, w, D), np.uint16)
        raw = np.empty((H, W), np.uint16)
        self.check(self.lib.sgmx_get_stages_census(self.h, index, _ptr(Cv), _ptr(S), _ptr(raw)))
        return dict(C=Cv, S=S, raw=raw)

    def data_cost_image(self, H, W, index=0):
        out = np.empty((H, W), np.uint8)
        self.check(self.lib.sgmx_get_data_cost_wta_u8(self.h, index, _ptr(out)))
        return out


def save_representation(path, tensor):
    """Writer for the descriptor format of sgm.h:378-410 (the reference only reads it)."""
    t = np.ascontiguousarray(tensor, np.float32)
    with open(path, "wb") as f:
        f.write(struct.pack("<i", 3))
        f.write(struct.pack("<3Q", *t.shape))
        t.tofile(f)


def load_representation(path):
    """SGM::LoadRepresentationFromFile (sgm.h:378-410): int32 dims (= 3), uint64 size[3]
    = {H, W, K}, float32[H][W][K]."""
    with open(path, "rb") as f:
        (dims,) = struct.unpack("<i", f.read(4))
        if dims != 3:
            raise ValueError("descriptor file: dims != 3")  # the reference does `throw 1` (sgm.h:391)
        size = struct.unpack("<3Q", f.read(24))
        data = np.fromfile(f, np.float32, int(np.prod(size)))
    if data.size != int(np.prod(size)):
        raise ValueError("descriptor file truncated")
    return data.reshape(size)


class SGM(Engine):
    """recon::SGM (sgm.h:61-65), COST_CNN build: float data cost."""

    def compute_cost(self, cost):
        """cost: (H, W, D) float32 data-cost volume (what sgm.cc:76-84 builds) -> (H, W) uint16."""
        cost = np.ascontiguousarray(cost, np.float32)
        H, W, D = cost.shape
        disp = np.empty((H, W), np.uint16)
        self.check(self.lib.sgmx_compute_cost_f32(self.h, _ptr(cost), W, H, D, _ptr(disp)))
        return disp

    def compute_desc(self, desc_left, desc_right):
        """(H, W, K) float32 descriptor tensors -> (H, W) uint16: L2 data cost (sgm.cc:76-84) on the
        GPU, then the float path."""
        fl = np.ascontiguousarray(desc_left, np.float32)
        fr = np.ascontiguousarray(desc_right, np.float32)
        if fl.shape != fr.shape or fl.ndim != 3:
            raise ValueError("want two (H, W, K) float32 tensors of equal shape")
        H, W, K = fl.shape
        disp = np.empty((H, W), np.uint16)
        self.check(self.lib.sgmx_compute_desc_f32(self.h, _ptr(fl), _ptr(fr), W, H, K, _ptr(disp)))
        return disp

    def Compute(self, left_desc_path, right_desc_path):
        """cv::Mat SGM::Compute(left_desc_path, right_desc_path) (sgm.h:65, sgm.cc:14-174)."""
        return self.compute_desc(load_representation(left_desc_path), load_representation(right_desc_path))

    def compute_cost_device(self, d_cost, d_disp, n, W, H, stream=0):
        self.check(self.lib.sgmx_compute_cost_f32_dev(self.h, n, d_cost, W, H, self.params.disp_range, d_disp, stream))

    def stages(self, H, W, index=0):
        S = np.empty((H, W, self.params.disp_range), np.float32)
        raw = np.empty((H, W), np.uint16)
        self.check(self.lib.sgmx_get_stages_f32(self.h, index, _ptr(S), _ptr(raw)))
        return dict(S=S, raw=raw)


def load_data_cost(path):
    """SGMStereo::LoadDataCostFromFile (new_sgm/sgm_stereo.cc:602-627): int32 dims (= 3), uint64
    size[3] = {H, W, D}, float32[H][W][D] -- the same container as the descriptor files."""
    return load_representation(path)


class SGMStereo:
    """recon::SGMStereo (reconstruction/base/new_sgm/sgm_stereo.h:11-81): the reference's second float
    pipeline -- eight textbook paths on the left and on the derived right volume, speckle filter,
    correct / mismatch / occlusion states, interpolation, 5x5 median.  Same method names and defaults
    (kDisparityRange 256, kP1 3, kP2 40, kConsistencyThreshold 1; sgm_stereo.h:17-22).  The disparity
    range of a data-cost volume comes with the volume (sgm_stereo.cc:613-615), so the sgmx context is
    made on first use and remade when the range or the penalties change."""

    def __init__(self, device=0, max_batch=1):
        self.disp_range, self.P1, self.P2, self.consistency_threshold = 256, 3, 40, 1
        self.device, self.max_batch = device, max_batch
        self._eng = None

    def SetSmoothnessCostParameters(self, P1, P2):
        """sgm_stereo.cc:16-28 (std::invalid_argument there)."""
        if P1 < 0 or P2 < 0:
            raise ValueError("smoothness penalty value is less than zero")
        if P1 >= P2:
            raise ValueError("small value of smoothness penalty must be smaller than large penalty value")
        # The reference keeps the penalties as float (sgm_stereo.h:63-64); this engine's parameter block is the
        # four ints of StereoSGMParams, so fractional penalties are refused rather than silently truncated
        # (NaN passes both comparisons above, infinities the first: refused here, before any int()).
        if not (math.isfinite(P1) and math.isfinite(P2)) or not (float(P1).is_integer() and float(P2).is_integer()) \
                or P2 > 2 ** 31 - 1:
            raise ValueError("smoothness penalties must be whole numbers (sgmx_params holds ints)")
        self.P1, self.P2 = int(P1), int(P2)

    def SetConsistencyThreshold(self, consistency_threshold):
        """sgm_stereo.cc:30-36."""
        if consistency_threshold < 0:
            raise ValueError("threshold for LR consistency must be positive")
        self.consistency_threshold = int(consistency_threshold)

    def _engine(self, D):
        key = (D, self.P1, self.P2)
        if self._eng is None or self._eng[0] != key:
            if self._eng is not None:
                self._eng[1].close()
            self._eng = (key, Engine(StereoSGMParams(disp_range=D, window_sz=1, penalty1=self.P1, penalty2=self.P2,
                                                     device=self.device, max_batch=self.max_batch)))
        return self._eng[1]

    def compute_cost(self, cost):
        """(H, W, D) float32 left data-cost volume -> (H, W) uint16 (ComputeSGM, sgm_stereo.cc:58-101)."""
        cost = np.ascontiguousarray(cost, np.float32)
        H, W, D = cost.shape
        e = self._engine(D)
        disp = np.empty((H, W), np.uint16)
        e.check(e.lib.sgmx_sgmstereo_cost_f32(e.h, _ptr(cost), W, H, D, self.consistency_threshold, _ptr(disp)))
        return disp

    def compute_cost_device(self, d_cost, d_disp, n, W, H, D, stream=0):
        e = self._engine(D)
        e.check(e.lib.sgmx_sgmstereo_cost_f32_dev(e.h, n, d_cost, W, H, D, self.consistency_threshold, d_disp, stream))

    def compute_desc(self, desc_left, desc_right):
        """(H, W, K) float32 descriptor tensors -> (H, W) uint16 (Compute(left, right), sgm_stereo.cc:38-50)."""
        fl = np.ascontiguousarray(desc_left, np.float32)
        fr = np.ascontiguousarray(desc_right, np.float32)
        if fl.shape != fr.shape or fl.ndim != 3:
            raise ValueError("sizes of left and right images are different")  # sgm_stereo.cc:112-114
        H, W, K = fl.shape
        e = self._engine(self.disp_range)
        disp = np.empty((H, W), np.uint16)
        e.check(e.lib.sgmx_sgmstereo_desc_f32(e.h, _ptr(fl), _ptr(fr), W, H, K, self.consistency_threshold, _ptr(disp)))
        return disp

    def Compute(self, *paths):
        """Compute(left_descriptors_path, right_descriptors_path) or Compute(data_cost_path)
        (sgm_stereo.h:26-29); returns the CV_16U disparity the reference writes through its pointer."""
        if len(paths) == 2:
            return self.compute_desc(load_representation(paths[0]), load_representation(paths[1]))
        if len(paths) == 1:
            return self.compute_cost(load_data_cost(paths[0]))
        raise TypeError("Compute(left_desc_path, right_desc_path) or Compute(data_cost_path)")

    def stages(self, H, W, index=0):
        e = self._eng[1]
        D = self._eng[0][0]
        S = np.empty((H, W, D), np.float32)
        ld = np.empty((H, W), np.int32)
        rd = np.empty((H, W), np.int32)
        st = np.empty((H, W), np.uint8)
        raw = np.empty((H, W), np.uint16)
        e.check(e.lib.sgmx_get_stages_sgmstereo(e.h, index, _ptr(S), _ptr(ld), _ptr(rd), _ptr(st), _ptr(raw)))
        return dict(S=S, left=ld, right=rd, states=st, raw=raw)

    @property
    def engine(self):
        return self._eng[1] if self._eng else None
