#!/usr/bin/env python3
"""bench.py -- throughput of the census/Hamming SGM hot path on MI355X.

One "step" = one pass of the whole hot path (census -> Hamming volume -> 8 ordered
direction sweeps -> WTA/LR/subpixel -> median) over one batch of B synthetic KITTI-size
stereo pairs per GPU, inputs and outputs resident in HBM.  Workload = BASELINE.json
configs[1]: KITTI 1241x376, D=128, 8 paths, wsz=5, P1=3, P2=60.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: stereo pairs are independent, so every rank runs its own B pairs per step with
no data-path collective (weak scaling); RCCL only carries the barrier and the max-over-ranks
of the elapsed time.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
N_SIMD = 1024          # 256 CUs x 4 SIMDs
# Issue cost of the sweeps' instruction mix, measured by exp/valu_rate.hip (profiles/r02_valu_issue.txt): every
# opcode of the step but v_xor_b32 costs 4.25 cycles per wave64 instruction per SIMD at 4 waves per SIMD
# (v_pk_*_u16, v_bcnt, DPP, v_lshl_add, v_perm, v_alignbit, v_pk_maximum3_f16), v_xor_b32 2.5.
VALU_CYCLES_SLOW, VALU_CYCLES_FAST = 4.25, 2.5


def kernel_src_sha256():
    """Hash of the sources libsgmx.so is built from: profiles/*_hbm_traffic.json carries the hash it was measured
    at, and `roofline.traffic` is only reported when it is the hash of this tree."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "stereo_vision_amd", "csrc", "*.h*"))) + [os.path.join(ROOT, "include", "sgmx.h")]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def find_traffic(cfg):
    """Newest profiles/r*_hbm_traffic.json measured on this configuration AND on these kernel sources."""
    import glob
    sha = kernel_src_sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")), reverse=True):
        try:
            t = json.load(open(f))
        except ValueError:
            continue
        if t.get("config") == cfg and t.get("kernel_src_sha256") == sha:
            return t, os.path.basename(f)
    return None, None


def box3(a):
    """3x3 box blur with edge replication (vectorised; float32)."""
    p = np.pad(a, 1, mode="edge")
    return (p[:-2, :-2] + p[:-2, 1:-1] + p[:-2, 2:] + p[1:-1, :-2] + p[1:-1, 1:-1] + p[1:-1, 2:] +
            p[2:, :-2] + p[2:, 1:-1] + p[2:, 2:]) / np.float32(9)


def synthetic_pair(seed, H, W, max_disp):
    """SURVEY.md 8d synthetic input: smooth random texture; the right view is the left
    shifted by a piecewise-constant disparity field (row bands x column blocks) + N(0,2)."""
    rng = np.random.default_rng(seed)
    base = box3(rng.integers(0, 256, size=(H, W + max_disp)).astype(np.float32))
    left = base[:, :W]
    right = np.empty_like(left)
    band = 24
    for y0 in range(0, H, band):
        d = int(rng.integers(0, max_disp))
        right[y0:y0 + band] = base[y0:y0 + band, d:d + W]
    right = right + rng.normal(0, 2.0, size=right.shape).astype(np.float32)
    to8 = lambda a: np.ascontiguousarray(np.clip(a, 0, 255).astype(np.uint8))
    return to8(left), to8(right)


def secondary_figures(eng, sv, args, Lh, Rh, B, W, H):
    """Figures beside the headline, same workload, clearly labelled (never `value`):
    host_batch_pairs_per_s -- page-locked host buffers -> sgmx_compute_census_u8_batch -> host (PCIe both ways);
    single_pair_ms -- one pair per sgmx_compute_census_u8 call, the reference's own call pattern
    (reconstruction/main/sgm_single/main.cc:43-44), pageable host buffers."""
    import ctypes as C
    lib = eng.lib
    sec = {}
    n, nfull = B, W * H
    ptrs = []
    try:
        for nbytes in (n * nfull, n * nfull, n * nfull * 2):
            p = C.c_void_p()
            eng.check(lib.sgmx_host_alloc(eng.h, nbytes, C.byref(p)))
            ptrs.append(p)
        hl = np.ctypeslib.as_array(C.cast(ptrs[0], C.POINTER(C.c_uint8)), (n, H, W))
        hr = np.ctypeslib.as_array(C.cast(ptrs[1], C.POINTER(C.c_uint8)), (n, H, W))
        hl[:] = Lh
        hr[:] = Rh
        reps = 3
        eng.check(lib.sgmx_compute_census_u8_batch(eng.h, n, ptrs[0], ptrs[1], W, H, ptrs[2]))  # warm-up
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.check(lib.sgmx_compute_census_u8_batch(eng.h, n, ptrs[0], ptrs[1], W, H, ptrs[2]))
        sec["host_batch_pairs_per_s"] = round(reps * n / (time.perf_counter() - t0), 1)
        sec["host_batch_note"] = "%d pairs per call from page-locked host memory, results back to the host" % n
    finally:
        for p in ptrs:
            lib.sgmx_host_free(eng.h, p)
    one = sv.StereoSGM(sv.StereoSGMParams(disp_range=args.disp, window_sz=args.wsz, penalty1=args.p1, penalty2=args.p2,
                                          num_paths=args.paths, device=eng.params.device, max_batch=1))
    l1, r1 = np.ascontiguousarray(Lh[0]), np.ascontiguousarray(Rh[0])
    o1 = np.empty((H, W), np.uint16)

    def call():  # the C entry point itself, caller-owned buffers reused from call to call
        one.check(lib.sgmx_compute_census_u8(one.h, l1.ctypes.data, r1.ctypes.data, W, H, W, W, o1.ctypes.data, 2 * W))
    for _ in range(25):  # (a lightly loaded GPU and the host take a few dozen calls to settle: 1.3 ms at first, then 1.2)
        call()
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    sec["single_pair_ms"] = round((time.perf_counter() - t0) / reps * 1e3, 4)
    sec["single_pair_note"] = "sgmx_compute_census_u8, one pair per call, pageable host buffers in and out"
    one.close()
    return sec


def other_configs(sv, torch, device):
    """The other BASELINE.json configurations, timed in the same process after the headline (never `value`): inputs
    resident in HBM, HIP-event kernel times from the engine's own profiling hooks.
      float_volume  -- configs[3]: precomputed float data-cost volumes 1241x376x128 through recon::SGM's aggregation
                       (sgm.cc:103-118,177-254), 8 volumes per step; bytes = SURVEY.md 8d's 96 B/voxel model
      uhd_4k        -- configs[4]: 3840x2160, D=256, 8 paths, 8 pairs per call (the call runs them as launches of 3+3+2)
      tsukuba_4path -- configs[0]'s geometry on the GPU: 384x288, D=16, 4 paths, 64 pairs per step"""
    out = {}

    def guarded(name):  # a leg that fails (e.g. no room for the 4K scratch) leaves its error instead of a figure
        def deco(fn):
            try:
                fn()
            except Exception as exc:  # noqa: BLE001
                out[name] = {"error": "%s: %s" % (type(exc).__name__, exc)}
            torch.cuda.empty_cache()
        return deco

    def timed(fn, steps, warm):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / steps

    def kernel_ms(eng, steps):
        return {k: round(v[0] / steps, 4) for k, v in eng.kernel_stats().items() if v[1]}

    # ---- float data-cost volumes
    @guarded("float_volume")
    def _float():
        H, W, D, B, N = 376, 1241, 128, 8, 8
        g = torch.Generator(device="cuda").manual_seed(7)
        Cv = (torch.rand((B, H, W, D), device="cuda", generator=g) * 4).contiguous()
        do = torch.zeros((B, H, W), dtype=torch.int16, device="cuda")
        e = sv.SGM(sv.SGMParams(disp_range=D, penalty1=3, penalty2=60, device=device, max_batch=B))
        fn = lambda: e.compute_cost_device(Cv.data_ptr(), do.data_ptr(), B, W, H, 0)
        timed(fn, 1, 2)
        e.set_profiling(True)
        steps = 10
        dt = timed(fn, steps, 0)
        dirs = e.direction_stats()
        km = kernel_ms(e, steps)
        e.set_profiling(False)
        V = H * W * D
        sweep_ms = sum(d[0] for d in dirs) / max(1, sum(d[1] for d in dirs))
        sweep_bytes = V * B * (12 * N - 4) / N  # C read once per sweep; S written by the first sweep, read-modify-written by the others
        out["float_volume"] = {
            "workload": "float data-cost volumes %dx%dx%d, %d paths, P1=3, P2=60, %d volumes per step" % (W, H, D, N, B),
            "volumes_per_s": round(B / dt, 1), "ms_per_step": round(dt * 1e3, 3), "dtype": "f32",
            "model_bytes_per_voxel": 12 * N, "step_GBs": round(12 * N * V * B / dt / 1e9, 1),
            "step_frac_of_peak": round(12 * N * V * B / dt / 1e9 / HBM_PEAK_GBS, 4),
            "sweep": {"kernel": "aggregate_f32_kernel", "avg_launch_ms": round(sweep_ms, 4), "alg_bytes_per_launch": int(sweep_bytes),
                      "achieved": round(sweep_bytes / (sweep_ms * 1e-3) / 1e9, 1), "unit": "GB/s",
                      "frac": round(sweep_bytes / (sweep_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                      "per_direction_ms": [round(d[0] / max(1, d[1]), 4) for d in dirs]},
            "kernel_ms_per_step": km}
        e.close()
        del Cv, do, e
        torch.cuda.empty_cache()

    # ---- 3840x2160, D = 256
    @guarded("uhd_4k")
    def _uhd():
        H, W, D, B = 2160, 3840, 256, 8
        pairs = [synthetic_pair(7 + i, H, W, 240) for i in range(2)]
        L = torch.from_numpy(np.stack([pairs[i % 2][0] for i in range(B)])).cuda()
        R = torch.from_numpy(np.stack([pairs[i % 2][1] for i in range(B)])).cuda()
        do = torch.zeros((B, H, W), dtype=torch.int16, device="cuda")
        e = sv.StereoSGM(sv.StereoSGMParams(disp_range=D, window_sz=5, penalty1=3, penalty2=60, device=device, max_batch=B))
        fn = lambda: e.compute_device(L.data_ptr(), R.data_ptr(), do.data_ptr(), B, W, H, 0)
        timed(fn, 1, 1)
        e.set_profiling(True)
        steps = 4
        dt = timed(fn, steps, 0)
        dirs = e.direction_stats()
        km = kernel_ms(e, steps)
        e.set_profiling(False)
        h_, w_, Dp = H - 4, W - 4, 256
        pair_bytes = h_ * w_ * Dp + h_ * w_ * 4 + h_ * (w_ + Dp) * 4 + 4 * (2 * w_ + 2 * h_) * Dp * 2  # as `roofline` below
        sweeps_ms = sum(d[0] for d in dirs) / steps  # all sweeps of one step (8 pairs)
        out["uhd_4k"] = {
            "workload": "census/Hamming SGM %dx%d, D=%d, 8 paths, wsz=5, P1=3, P2=60, %d pairs per call" % (W, H, D, B),
            "frames_per_s": round(B / dt, 2), "ms_per_step": round(dt * 1e3, 2), "mde_per_s": round(B / dt * W * H * D / 1e6, 1),
            "dtype": "u16",
            "sweep": {"kernel": "aggregate3_kernel", "per_direction_ms_per_%d_pairs" % B: [round(d[0] / steps, 3) for d in dirs],
                      "launches_per_direction_per_step": dirs[0][1] // steps,
                      "alg_bytes_per_direction": int(B * pair_bytes),
                      "achieved": round(8 * B * pair_bytes / (sweeps_ms * 1e-3) / 1e9, 1), "unit": "GB/s",
                      "frac": round(8 * B * pair_bytes / (sweeps_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            "kernel_ms_per_step": km}
        e.close()
        del L, R, do, e
        torch.cuda.empty_cache()

    # ---- Tsukuba geometry, 4 paths
    @guarded("tsukuba_4path")
    def _tsukuba():
        H, W, D, B = 288, 384, 16, 64
        pairs = [synthetic_pair(3 + i, H, W, 12) for i in range(4)]
        L = torch.from_numpy(np.stack([pairs[i % 4][0] for i in range(B)])).cuda()
        R = torch.from_numpy(np.stack([pairs[i % 4][1] for i in range(B)])).cuda()
        do = torch.zeros((B, H, W), dtype=torch.int16, device="cuda")
        e = sv.StereoSGM(sv.StereoSGMParams(disp_range=D, window_sz=5, penalty1=3, penalty2=60, num_paths=4, device=device, max_batch=B))
        fn = lambda: e.compute_device(L.data_ptr(), R.data_ptr(), do.data_ptr(), B, W, H, 0)
        timed(fn, 1, 3)
        e.set_profiling(True)
        steps = 50
        dt = timed(fn, steps, 0)
        km = kernel_ms(e, steps)
        e.set_profiling(False)
        out["tsukuba_4path"] = {
            "workload": "census/Hamming SGM %dx%d, D=%d, 4 paths, wsz=5, P1=3, P2=60, %d pairs per step" % (W, H, D, B),
            "frames_per_s": round(B / dt, 1), "ms_per_step": round(dt * 1e3, 4), "mde_per_s": round(B / dt * W * H * D / 1e6, 1),
            "dtype": "u16", "kernel_ms_per_step": km}
        e.close()
    return out


def hamming_library_leg(sv, device, na=100000, nb=4096, ncodes=16, ht=44):
    """core/hamming.h's set functions at na x nb signatures of ncodes bytes through the C ABI (host buffers in, host
    buffers out: these entry points have no device-resident form, core/hamming.h:62-112 has none either), next to the
    reference's own core/hamming.c (oracle/_ref/libhamming_ref.so, built from the reference tree where it lies) on the
    box's host cores -- the one place where the CPU baseline IS the reference's compiled code (kind "reference").
    Matrix: 2 na nb bytes out; write-bound on the GPU, PCIe-bound end to end."""
    from oracle import oracle as o
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, size=(na, ncodes)).astype(np.uint8)
    b = rng.integers(0, 256, size=(nb, ncodes)).astype(np.uint8)
    e = sv.StereoSGM(sv.StereoSGMParams(device=device))
    out = {"workload": "%d x %d signatures of %d bytes, threshold %d" % (na, nb, ncodes, ht)}

    def gpu(fn, reps=3):
        fn()  # (sizes the context's buffers)
        e.kernel_stats()
        e.set_profiling(True)
        t = time.perf_counter()
        for _ in range(reps):
            r = fn()
        dt = (time.perf_counter() - t) / reps
        ms, n = e.kernel_stats()["hamming_library"]
        e.set_profiling(False)
        return r, dt, ms / reps

    dis, dt, kms = gpu(lambda: e.hamming_matrix(a, b))
    mbytes = 2.0 * na * nb
    out["matrix"] = {"entry": "sgmx_hamming_matrix_u16 (compute_hamming, core/hamming.c:215-236)",
                     "kernel_ms": round(kms, 4), "bytes_out": int(mbytes), "kernel_GBs": round(mbytes / (kms * 1e-3) / 1e9, 1),
                     "kernel_frac_of_peak": round(mbytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "write_bound_ms_at_6300_GBs": round(mbytes / 6.3e12 * 1e3, 4),
                     "end_to_end_ms": round(dt * 1e3, 2), "end_to_end_note": "pageable host buffers; the %d MB matrix crosses PCIe" % (mbytes / 1e6)}
    (idx, hams), dt, kms = gpu(lambda: e.match_hamming_thres(a, b, ht))
    out["match"] = {"entry": "sgmx_match_hamming_count + sgmx_match_hamming_thres_prealloc (core/hamming.c:300-322, 725-765)",
                    "matches": int(len(hams)), "kernel_ms_count_plus_emit": round(kms, 4), "end_to_end_ms": round(dt * 1e3, 2),
                    "pairs_per_s_end_to_end": round(na * nb / dt, 0)}
    e.close()
    ncore = min(len(os.sched_getaffinity(0)), int(os.environ.get("OMP_NUM_THREADS", "0")) or 10 ** 6)
    t = time.perf_counter()
    rdis = o.ref_hamming_matrix_thread(a, b)
    t_thread = time.perf_counter() - t
    if rdis is None:
        out["cpu_reference"] = None  # oracle/_ref not built (no reference tree where this tree was built)
        return out
    t = time.perf_counter()
    ridx, rhams = o.ref_match_hamming(a, b, ht)
    t_match = time.perf_counter() - t
    out["cpu_reference"] = {
        "kind": "reference", "library": "oracle/_ref/libhamming_ref.so = the reference's core/hamming.c, gcc -O2 -msse4.2 -fopenmp",
        "compute_hamming_thread_ms": round(t_thread * 1e3, 1), "compute_hamming_thread_cores": ncore,
        "match_hamming_count_plus_prealloc_ms": round(t_match * 1e3, 1), "match_cores": 1,
        "sample": "the whole workload, once",
        "gpu_matrix_equal": bool(np.array_equal(dis, rdis)),
        "gpu_matches_equal": bool(np.array_equal(idx, ridx) and np.array_equal(hams, rhams))}
    return out


def sequence_files_leg(npairs=1056, threads=16):
    """BASELINE.json configs[2] file to file on this rank's GPU: `sgm_batch census` (the counterpart of
    scripts/run_sgm_batch_census.py:14-34: PNG pairs in, disparities/ and norm_hist/ PNGs out) on npairs synthetic KITTI-size
    pairs (64 distinct files, the rest hard links: the same decode work per file), groups of 32, `threads` host threads.
    A child process; bounded by the host's PNG work, not by the GPU."""
    import re
    import shutil
    import subprocess
    import tempfile
    from PIL import Image
    exe = os.path.join(ROOT, "stereo_vision_amd", "bin", "sgm_batch")
    root = tempfile.mkdtemp(prefix="sgmx_seq_")
    try:
        for d in ("image_0", "image_1"):
            os.makedirs(os.path.join(root, d))
        base = [synthetic_pair(100 + i, 376, 1241, 100) for i in range(8)]
        uniq = 64
        for i in range(npairs):
            for d, k in (("image_0", 0), ("image_1", 1)):
                dst = os.path.join(root, d, "%06d_10.png" % i)
                if i < uniq:
                    Image.fromarray(np.roll(base[i % 8][k], i, axis=0)).save(dst, compress_level=1)
                else:
                    os.link(os.path.join(root, d, "%06d_10.png" % (i % uniq)), dst)
        out = {"workload": "%d KITTI-size PNG pairs, D=128, wsz=5, 8 paths, P1=3, P2=60; sgm_batch census --batch 32 --threads %d" % (npairs, threads)}
        for name, extra in (("pairs_per_s_with_png_outputs", []), ("pairs_per_s_decode_only", ["--no-write"])):
            cmd = [exe, "census", root + "/image_0", root + "/image_1", root + "/out", "3", "60", "--disp-range", "128", "--window-sz", "5",
                   "--batch", "32", "--threads", str(threads)] + extra
            t = time.perf_counter()
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
            dt = time.perf_counter() - t
            m = re.search(r"pairs: (\d+)  ranks: (\d+)  wall: ([0-9.]+) s  ([0-9.]+) pairs/s  checksum xor: ([0-9a-f]{16})", r.stdout)
            if r.returncode != 0 or not m:
                raise RuntimeError("sgm_batch failed: " + r.stderr[-300:])
            out[name] = float(m.group(4))
            out[name.replace("pairs_per_s", "process_wall_s")] = round(dt, 3)
            out["checksum_xor"] = m.group(5)
        out["host_threads"] = threads
        out["host_cores_available"] = len(os.sched_getaffinity(0))
        return out
    finally:
        shutil.rmtree(root, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="stereo pairs per GPU per step (one launch group)")
    ap.add_argument("--width", type=int, default=1241)
    ap.add_argument("--height", type=int, default=376)
    ap.add_argument("--disp", type=int, default=128)
    ap.add_argument("--wsz", type=int, default=5)
    ap.add_argument("--p1", type=int, default=3)
    ap.add_argument("--p2", type=int, default=60)
    ap.add_argument("--paths", type=int, default=8)
    ap.add_argument("--cpu-frames", type=int, default=8, help="frames of the same workload timed on the CPU oracle (0 = skip)")
    ap.add_argument("--no-events", action="store_true", help="do not bracket kernels with HIP events in the timed region")
    ap.add_argument("--all-events", action="store_true", help="bracket every kernel launch of the timed region (two events each) "
                    "instead of the sweeps only")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse "
                    "the multi-rank path on a box with fewer GPUs than ranks)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses cuda:0 (RCCL refuses two "
                    "ranks on one device: use it with --backend gloo)")
    ap.add_argument("--dist-always", action="store_true", help="initialise the process group, barrier and all-reduce "
                    "even with one rank (exercises the RCCL leg on a one-GPU box)")
    ap.add_argument("--min-seconds", type=float, default=None, help="floor on the timed region: the block of --steps steps "
                    "is repeated (`timed_blocks` whole blocks, same count on every rank) until it lasts this long.  Default: "
                    "0 with one rank (exactly --steps steps), 2 s with several (a short region there is mostly barrier jitter)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary figures (single pair, host batch, "
                    "the other BASELINE configurations)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the float-volume / 4K / Tsukuba legs of `secondary`")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the SGM engine has no CPU path")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.dist_always
    if use_dist:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    import stereo_vision_amd as sv

    W, H, D, B = args.width, args.height, args.disp, args.batch
    eng = sv.StereoSGM(sv.StereoSGMParams(disp_range=D, window_sz=args.wsz, penalty1=args.p1, penalty2=args.p2,
                                          num_paths=args.paths, device=local_rank, max_batch=B))
    # B distinct seeded pairs per rank; a handful of distinct textures is enough to keep the
    # data realistic, so generate min(B, 8) and tile them
    ndist = min(B, 8)
    pairs = [synthetic_pair(1234 + rank * 1000 + i, H, W, 100) for i in range(ndist)]
    Lh = np.stack([pairs[i % ndist][0] for i in range(B)])
    Rh = np.stack([pairs[i % ndist][1] for i in range(B)])
    dL = torch.from_numpy(Lh).cuda()
    dR = torch.from_numpy(Rh).cuda()
    dO = torch.zeros((B, H, W), dtype=torch.int16, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        eng.compute_device(dL.data_ptr(), dR.data_ptr(), dO.data_ptr(), B, W, H, stream)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    eng.kernel_stats()  # clear
    # HIP events over the timed region, on the launch stream: around the dominant kernel (the direction sweeps) only, with
    # shared boundary events (mode 2) -- bracketing all 13 kernels of a step with two events each costs 1.4 % of the step,
    # this a third of that.  The other kernel classes are timed in a few extra steps after the timed region (mode 1).
    light = args.p2 <= 255 and not args.all_events
    eng.set_profiling(0 if args.no_events else (2 if light else 1))
    # The timed region is `blocks` whole blocks of --steps steps.  One rank: one block, exactly --steps steps.  Several
    # ranks: as many blocks as last min_seconds (2 s by default), the same count on every rank, estimated from the
    # slowest rank's probe step -- --steps steps of 6 ms are mostly barrier jitter at N = 8.
    dev_t = "cuda" if (use_dist and args.backend == "nccl") else "cpu"
    min_seconds = args.min_seconds if args.min_seconds is not None else (2.0 if world > 1 else 0.0)
    blocks = 1
    if min_seconds > 0:
        tp = time.perf_counter()
        step()
        fence()
        est = torch.tensor([time.perf_counter() - tp], dtype=torch.float64, device=dev_t)
        if use_dist:
            dist.all_reduce(est, op=dist.ReduceOp.MAX)
        blocks = max(1, -(-int(min_seconds / max(float(est.item()), 1e-6) + 1) // args.steps))
        eng.kernel_stats()
    steps = args.steps * blocks
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0  # this rank's own steps, before it waits for the others
    fence()
    dt = time.perf_counter() - t0
    dirs = eng.direction_stats()
    stats = eng.kernel_stats()
    eng.set_profiling(False)
    other_steps = 0
    if light and not args.no_events:  # census / WTA / median times: a few steps outside the timed region
        other_steps = 3
        eng.set_profiling(1)
        for _ in range(other_steps):
            step()
        eng.direction_stats()
        extra = eng.kernel_stats()
        eng.set_profiling(False)
    per_rank_fps = [round(B * steps / dt_own, 2)]
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev_t)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        own = torch.tensor([dt_own], dtype=torch.float64, device=dev_t)
        allr = [torch.zeros_like(own) for _ in range(world)]
        dist.all_gather(allr, own)
        per_rank_fps = [round(B * steps / float(x.item()), 2) for x in allr]

    frames = world * B * steps
    fps = frames / dt
    m = (args.wsz - 1) // 2
    V = (W - 2 * m) * (H - 2 * m) * D
    alg_total, alg_agg = eng.algorithmic_bytes(W, H)

    out = {
        "metric": "frames/s (KITTI 1241x376, D=128, 8-path census SGM)",
        "value": round(fps, 2),
        "unit": "frames/s",
        "mde_per_s": round(fps * W * H * D / 1e6, 1),
        "n_gpus": world,
        "steps": args.steps,
        "timed_blocks": blocks,  # the timed region = timed_blocks x steps steps (1 unless --min-seconds / several ranks)
        "warmup": args.warmup,
        "ms_per_step": round(dt / steps * 1e3, 4),
        "per_rank_fps": per_rank_fps,  # each rank's own steps / its own time, before the closing barrier
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u16",
        "data": "synthetic",
        "config": {"workload": "KITTI-size %dx%d census/Hamming SGM, D=%d, %d paths, wsz=%d, P1=%d, P2=%d" %
                               (W, H, D, args.paths, args.wsz, args.p1, args.p2),
                   "pairs_per_gpu_per_step": B, "sharding": "independent stereo pairs per rank, no data-path collective",
                   "dist_backend": (args.backend if use_dist else None)},
    }

    if rank == 0:
        # Dominant kernel = the direction sweeps (8 launches per step).  `achieved` = the ALGORITHMIC bytes of
        # one sweep launch in THIS engine's formulation / its average duration, measured live with
        # HIP events on the launch stream: per pair one E byte per padded voxel written (h*w*Dp) + the two census
        # images read once (h*w*4 + h*(w+Dp)*4) + the two border rings read and written (4 * (2w+2h)*Dp*2).
        # The materialised formulation (--p2 > 255) moves SURVEY.md 8d's bytes: V*(5N-2)/N per launch and pair.
        # SURVEY.md 8d's yardstick (materialised C + S volumes, 41 B/voxel) is kept as `model_bytes_equiv_GBs`:
        # this engine never moves those bytes, so that figure is NOT a fraction of the HBM peak.
        fast = args.p2 <= 255
        agg_ms, agg_n = stats["aggregate"]
        roof = None
        if agg_n:
            h_, w_ = H - 2 * m, W - 2 * m
            Dp = 32 * (1 if D <= 32 else 2 if D <= 64 else 4 if D <= 128 else 8 if D <= 256 else 12 if D <= 384 else 16)
            # pairs per kernel launch: B, unless the engine runs the call as several launches (more than 32 pairs or
            # more than 208 MB of census images per launch: DESIGN.md 4)
            ppl = B * args.paths * steps / agg_n
            model_bytes = ppl * alg_agg / args.paths
            if fast:
                launch_bytes = ppl * (h_ * w_ * Dp + h_ * w_ * 4 + h_ * (w_ + Dp) * 4 + 4 * (2 * w_ + 2 * h_) * Dp * 2)
            else:
                launch_bytes = model_bytes
            avg_s = agg_ms / agg_n * 1e-3
            ach = launch_bytes / avg_s / 1e9
            cfg = {"batch": B, "width": W, "height": H, "disp": D, "paths": args.paths, "wsz": args.wsz}
            t, tname = find_traffic(cfg) if fast else (None, None)
            traffic = step_bytes = valu = None
            if t:
                k = t["kernels"]
                traffic = k["aggregate3_kernel"]["hbm_bytes_per_launch"]
                # HBM bytes of one whole step (PMC passes under profiles/): census + N sweeps + WTA + median
                step_bytes = (k["census4_kernel"]["hbm_bytes_per_launch"] + args.paths * traffic +
                              k["wta3_kernel"]["hbm_bytes_per_launch"] + k["median3_kernel"]["hbm_bytes_per_launch"])
                sq = t.get("sq", {}).get("aggregate3_kernel")
                if sq:
                    # VALU issue: instructions per launch x measured cycles per instruction / (SIMDs x clock x time)
                    nv, nx = sq["SQ_INSTS_VALU"], sq.get("xor_share", 0.0) * sq["SQ_INSTS_VALU"]
                    cyc = (nv - nx) * VALU_CYCLES_SLOW + nx * VALU_CYCLES_FAST
                    valu = {"insts_per_launch": nv, "cycles_per_inst_slow_fast": [VALU_CYCLES_SLOW, VALU_CYCLES_FAST],
                            "fast_share": sq.get("xor_share", 0.0), "clock_GHz": sq.get("clock_GHz"),
                            "frac": round(cyc / (N_SIMD * sq.get("clock_GHz", 2.2) * 1e9 * avg_s), 4)}
            roof = {"bound": "hbm", "kernel": "aggregate3_kernel" if fast else "aggregate_u16_kernel",
                    "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                    "traffic": traffic, "traffic_source": tname, "avg_launch_ms": round(agg_ms / agg_n, 4),
                    "alg_bytes_per_launch": int(launch_bytes), "pairs_per_launch": round(ppl, 3),
                    "hbm_GBs_from_traffic": round(traffic / avg_s / 1e9, 1) if traffic else None,
                    "valu_issue": valu,
                    "model_bytes_per_launch": model_bytes,
                    "model_bytes_equiv_GBs": round(model_bytes / avg_s / 1e9, 1),
                    "step_hbm_bytes_from_pmc": step_bytes,
                    "step_hbm_GBs": round(step_bytes / (dt / steps) / 1e9, 1) if step_bytes else None,
                    "step_hbm_frac_of_peak": round(step_bytes / (dt / steps) / 1e9 / HBM_PEAK_GBS, 4) if step_bytes else None,
                    "kernel_ms_per_step": ({k: round(v[0] / steps, 4) for k, v in stats.items()} if not other_steps else
                                           dict({k: round(v[0] / other_steps, 4) for k, v in extra.items()},
                                                aggregate=round(stats["aggregate"][0] / steps, 4))),
                    "kernel_ms_per_step_note": ("aggregate: HIP events over the timed region (sweeps only, shared boundary events); "
                                                "the other classes: %d extra steps after it with every launch bracketed" % other_steps
                                                if other_steps else "HIP events around every launch of the timed region"),
                    "per_direction_ms": [round(d[0] / max(1, d[1]), 4) for d in dirs]}
        out["roofline"] = roof

        cpu = None
        if world == 1 and args.cpu_frames > 0:
            from oracle import oracle as o
            o.build()
            nfr = min(args.cpu_frames, ndist)
            got = dO[:nfr].cpu().numpy().view(np.uint16)
            t1 = time.perf_counter()
            ok = True
            for i in range(nfr):
                want = o.sgm_census(pairs[i][0], pairs[i][1], D, args.wsz, args.p1, args.p2, args.paths)
                ok = ok and bool(np.array_equal(want, got[i]))
            cdt = time.perf_counter() - t1
            cpu = {"value": round(nfr / cdt, 4), "unit": "frames/s", "cores": 1, "kind": "port",
                   "sample": "%d of the benchmark's own %dx%d D=%d frames through oracle/sgm_oracle.c (single thread, "
                             "like the reference's aggregation)" % (nfr, W, H, D),
                   "gpu_output_bit_exact_on_sample": ok}
            # Best case for a CPU node: independent frames on all the cores this process may use,
            # one single-threaded oracle call per core (the reference scripts run pairs one after the other).
            from concurrent.futures import ThreadPoolExecutor
            ncore = min(len(os.sched_getaffinity(0)), int(os.environ.get("OMP_NUM_THREADS", "0")) or 10 ** 6, 32)
            if ncore > 1:
                t2 = time.perf_counter()
                with ThreadPoolExecutor(ncore) as ex:  # ctypes calls release the GIL
                    list(ex.map(lambda i: o.sgm_census(pairs[i % ndist][0], pairs[i % ndist][1], D, args.wsz, args.p1,
                                                       args.p2, args.paths), range(ncore)))
                pdt = time.perf_counter() - t2
                cpu["independent_frames_on_all_cores"] = {"value": round(ncore / pdt, 4), "unit": "frames/s", "cores": ncore,
                                                          "sample": "%d frames, one per core" % ncore}
        out["cpu_baseline"] = cpu
        if world == 1 and not args.no_secondary:
            # the legs beside the headline never take the line down with them: a failure is recorded in its place
            def leg(name, fn):
                try:
                    return fn()
                except Exception as exc:  # noqa: BLE001
                    return {name + "_error": "%s: %s" % (type(exc).__name__, exc)}
            out["secondary"] = leg("secondary", lambda: secondary_figures(eng, sv, args, Lh, Rh, B, W, H))
            if not args.no_other_configs:
                eng.close()
                del dL, dR, dO
                torch.cuda.empty_cache()
                out["secondary"].update(leg("other_configs", lambda: other_configs(sv, torch, local_rank)))
                out["secondary"]["hamming_library"] = leg("hamming_library", lambda: hamming_library_leg(sv, local_rank))
                out["secondary"]["sequence_files"] = leg("sequence_files", sequence_files_leg)
        print(json.dumps(out), flush=True)

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
