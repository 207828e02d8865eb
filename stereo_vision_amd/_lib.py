"""ctypes binding of libsgmx.so (include/sgmx.h).  The engine is HIP-only: if the
library is missing this module raises, it never falls back to anything else."""
import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SGMX_LIB") or os.path.join(_HERE, "libsgmx.so")  # SGMX_LIB: experiment builds

SGMX_OK = 0
FLAG_MATERIALIZE = 1  # SGMX_FLAG_MATERIALIZE
ERR_NAMES = {1: "INVALID_ARG", 2: "RANGE", 3: "NO_DEVICE", 4: "HIP", 5: "OOM", 6: "STATE"}
K_CENSUS, K_HAMMING, K_AGGREGATE, K_WTA, K_MEDIAN, K_HAMLIB, K_COUNT = range(7)
KERNEL_CLASS_NAMES = ["census", "hamming_volume", "aggregate", "wta", "median", "hamming_library"]


class SgmxParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("disp_range", "window_sz", "penalty1", "penalty2", "num_paths", "device",
                                       "max_batch", "flags")]


# every symbol include/sgmx.h declares: name -> (restype, argtypes)
_vp, _i, _sz = C.c_void_p, C.c_int, C.c_size_t
SYMBOLS = {
    "sgmx_version": (C.c_char_p, []),
    "sgmx_create": (_i, [C.POINTER(SgmxParams), C.POINTER(_vp)]),
    "sgmx_destroy": (None, [_vp]),
    "sgmx_last_error": (C.c_char_p, [_vp]),
    "sgmx_compute_census_u8": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _i]),
    "sgmx_compute_census_u8_batch": (_i, [_vp, _i, _vp, _vp, _i, _i, _vp]),
    "sgmx_compute_census_u8_dev": (_i, [_vp, _i, _vp, _vp, _i, _i, _vp, _vp]),
    "sgmx_compute_census_u8_dev_noalloc": (_i, [_vp, _i, _vp, _vp, _i, _i, _vp, _vp]),
    "sgmx_noalloc_scratch_bytes": (C.c_size_t, []),
    "sgmx_compute_cost_f32": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "sgmx_compute_cost_f32_dev": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _vp]),
    "sgmx_compute_desc_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "sgmx_sgmstereo_cost_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "sgmx_sgmstereo_cost_f32_dev": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp]),
    "sgmx_sgmstereo_desc_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "sgmx_get_stages_sgmstereo": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "sgmx_kitti_eval_u16": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "sgmx_depth_from_disparity_u16": (_i, [_vp, _vp, _i, _i, C.c_double, C.c_double, _vp]),
    "sgmx_gaussian_blur3_u8": (_i, [_vp, _vp, _i, _i, _i, C.c_double, _vp, _i]),
    "sgmx_census_u32": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "sgmx_hamming_volume_u8": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "sgmx_hamming_matrix_u16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i]),
    "sgmx_host_alloc": (_i, [_vp, C.c_size_t, C.POINTER(C.c_void_p)]),
    "sgmx_host_free": (_i, [_vp, _vp]),
    "sgmx_match_hamming_count": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, C.POINTER(C.c_size_t)]),
    "sgmx_crossmatch_hamming_count": (_i, [_vp, _vp, C.c_long, _i, _i, C.POINTER(C.c_size_t)]),
    "sgmx_match_hamming_thres_prealloc": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "sgmx_crossmatch_hamming_prealloc": (_i, [_vp, _vp, C.c_long, _i, _i, _vp, _vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "sgmx_match_hamming_thres": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, C.c_size_t, C.POINTER(_vp), C.POINTER(C.c_size_t)]),
    "sgmx_crossmatch_hamming": (_i, [_vp, _vp, C.c_long, _i, _i, C.c_long, C.POINTER(_vp), C.POINTER(C.c_size_t)]),
    "sgmx_compute_hamming_thread": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i]),
    "sgmx_match_hamming_thres_nt": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(C.c_size_t)]),
    "sgmx_free": (None, [_vp]),
    "sgmx_get_stages_census": (_i, [_vp, _i, _vp, _vp, _vp]),
    "sgmx_get_stages_f32": (_i, [_vp, _i, _vp, _vp]),
    "sgmx_get_data_cost_wta_u8": (_i, [_vp, _i, _vp]),
    "sgmx_set_preblur3": (_i, [_vp, _i, C.c_double]),
    "sgmx_last_launch_group": (_i, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sgmx_set_profiling": (_i, [_vp, _i]),
    "sgmx_kernel_stats": (_i, [_vp, _i, C.POINTER(C.c_double), C.POINTER(C.c_long)]),
    "sgmx_direction_stats": (_i, [_vp, _i, C.POINTER(C.c_double), C.POINTER(C.c_long)]),
    "sgmx_algorithmic_bytes_census": (C.c_double, [_vp, _i, _i, C.POINTER(C.c_double)]),
}

_lib = None


def load():
    """Load libsgmx.so and bind every declared symbol; raises if the library or a
    symbol is missing (no fallback)."""
    global _lib
    if _lib is None:
        # PyTorch wheels bundle their own HIP/HSA runtime; two runtimes in one process cannot
        # both open the GPU.  Importing torch first makes libsgmx.so bind to the runtime torch
        # already loaded (same SONAME), so device pointers and streams can be shared.
        if "torch" not in sys.modules and os.environ.get("SGMX_NO_TORCH_PRELOAD") != "1":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libsgmx.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C stereo_vision_amd/csrc` (the engine has no CPU fallback)")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if the export is missing
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


class SgmxError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("sgmx error %d (%s): %s" % (code, ERR_NAMES.get(code, "?"), text))
        self.code = code
