"""Callers of the SGM engine: counterparts of the reference's CLI mains and batch scripts.

    reference                                              here
    reconstruction/main/sgm_single/main.cc:11-79            sgm_single(left, right, out_folder, P1, P2)
    reconstruction/main/sgm_cnn/main.cc:11-74               sgm_cnn(left_desc, right_desc, out_dir, prefix, P1, P2)
    scripts/run_sgm_batch_census.py:14-34                   list_census_pairs, make_output_dirs, run_batch_census
    scripts/run_sgm_batch_cnn.py:22-42                      list_cnn_pairs, run_batch_cnn
    scripts/run_sgm_batch_grid_search.py:14-54              list_numbered_pairs, grid_search_folder, run_grid_search
    scripts/run_sgm_batch.py:51-66                          list_same_name_pairs (left/ and right/ hold equal names)
    reconstruction/main/create_rgbd_dataset/main.cc:49-158  read_calib_kitti, fix_kitti_groundtruth, create_rgbd_dataset

Same file-selection rules, argv meaning and output files.  The reference runs one process per
pair, strictly sequentially; here a rank takes a contiguous slice of the pair list (one process
per GPU, no data-path collective -- stereo pairs are independent) and feeds the engine in batches.
Image decoding / PNG encoding and the 8-bit preview arithmetic are host-side caller code in the
reference too (OpenCV); all SGM arithmetic happens in libsgmx.so on the GPU.
"""
import os
from os.path import isfile, join

import numpy as np

from .sgm import SGM, SGMParams, StereoSGM, StereoSGMParams


# ---------------------------------------------------------------- sharding / statistics
def shard_range(n_items, rank, world):
    """Contiguous slice [lo, hi) of n_items for `rank` of `world`:
    ceil(n / world) items per rank, the last ranks may get fewer or none."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    per = -(-n_items // world)
    lo = min(n_items, rank * per)
    return lo, min(n_items, lo + per)


FNV_OFFSET = 1469598103934665603
FNV_PRIME = 1099511628211


def fnv_u16(img):
    """Checksum of a disparity map, one update per uint16 pixel."""
    h = FNV_OFFSET
    for v in np.asarray(img, np.uint16).reshape(-1).tolist():
        h = ((h ^ v) * FNV_PRIME) & 0xFFFFFFFFFFFFFFFF
    return h


def gather_stats(frames, seconds, checksum_xor=0):
    """Whole-job statistics over all ranks: total frames, the slowest rank's time and the XOR of
    the per-rank checksums.  Uses torch.distributed when a process group exists (RCCL on GPUs,
    gloo in the CPU tests); a single process returns its own numbers."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return dict(frames=int(frames), seconds=float(seconds), checksum_xor=int(checksum_xor), world=1)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([float(frames)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    s = torch.tensor([float(seconds)], dtype=torch.float64, device=dev)
    dist.all_reduce(s, op=dist.ReduceOp.MAX)
    # 64-bit checksums travel as two 32-bit halves (int64 tensors cannot hold an unsigned 64-bit value)
    c = torch.tensor([checksum_xor >> 32, checksum_xor & 0xFFFFFFFF], dtype=torch.int64, device=dev)
    parts = [torch.zeros_like(c) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, c)
    x = 0
    for p in parts:
        hi, lo = (int(v) for v in p.cpu().tolist())
        x ^= (hi << 32) | lo
    return dict(frames=int(t.item()), seconds=float(s.item()), checksum_xor=x, world=dist.get_world_size())


# ---------------------------------------------------------------- file selection
def list_census_pairs(left_dir, right_dir, token="_10.png"):
    """scripts/run_sgm_batch_census.py:25-31: every file of left_dir whose name contains
    "_10.png"; the right image has the same name in right_dir.  Sorted for determinism."""
    names = sorted(f for f in os.listdir(left_dir) if isfile(join(left_dir, f)) and token in f)
    return [(join(left_dir, f), join(right_dir, f)) for f in names]


def list_cnn_pairs(data_folder):
    """scripts/run_sgm_batch_cnn.py:33-41: every file whose name contains "left";
    prefix = name[:9]; right = prefix + "_right.bin".  Returns (left, right, prefix)."""
    names = sorted(f for f in os.listdir(data_folder) if isfile(join(data_folder, f)) and "left" in f)
    return [(join(data_folder, f), join(data_folder, f[:9] + "_right.bin"), f[:9]) for f in names]


def list_numbered_pairs(data_folder, start_num=0, stop_num=193, frame_step=1, left_prefix="/image_0/",
                        right_prefix="/image_1/", left_suffix="_10.png", right_suffix="_10.png", digits=6):
    """scripts/run_sgm_batch_grid_search.py:14-31,49-53: frames start_num .. stop_num (inclusive) named
    "%06d" + suffix under data_folder + prefix; plain string concatenation like the script."""
    out = []
    for i in range(start_num, stop_num + 1, frame_step):
        num_str = "%0*d" % (digits, i)
        out.append((data_folder + left_prefix + num_str + left_suffix, data_folder + right_prefix + num_str + right_suffix))
    return out


def grid_search_folder(out_folder_prefix, penalty1, penalty2):
    """scripts/run_sgm_batch_grid_search.py:41: one result folder per grid point."""
    return out_folder_prefix + "_" + str(penalty1) + "_" + str(penalty2) + "/data/"


def list_same_name_pairs(data_folder):
    """scripts/run_sgm_batch.py:51-60: every file of data_folder/left/, its partner has the same name in
    data_folder/right/.  Sorted for determinism."""
    left_dir, right_dir = data_folder + "/left/", data_folder + "/right/"
    names = sorted(f for f in os.listdir(left_dir) if isfile(join(left_dir, f)))
    return [(left_dir + f, right_dir + f) for f in names]


def make_output_dirs(out_folder):
    """scripts/run_sgm_batch_census.py:14-23 / run_sgm_batch_cnn.py:25-31."""
    existed = os.path.exists(out_folder)
    for sub in ("", "disparities", "norm_hist", "interpolation"):
        os.makedirs(join(out_folder, sub), exist_ok=True)
    if existed:
        print("WARNING: path exists - ", out_folder)
    return not existed


def read_string_list(filename):
    """core::FormatHelper::readStringList (core/format_helper.cc:232-245): an OpenCV FileStorage XML
    file whose first top-level node is a sequence of strings (config_files/kitti_00_lst.xml:1-7).
    Returns the list, or None where the reference returns false."""
    import re
    from xml.sax.saxutils import unescape
    try:
        text = open(filename).read()
    except OSError:
        return None
    m = re.search(r"<opencv_storage>(.*)</opencv_storage>", text, re.S)
    if not m:
        return None
    body = re.sub(r"<!--.*?-->", "", m.group(1), flags=re.S)
    m = re.search(r"<(\w+)[^>]*>(.*?)</\1>", body, re.S)
    if not m or "<" in m.group(2):
        return None
    ent = {"&quot;": '"', "&apos;": "'"}
    return [unescape(a if a else b, ent) for a, b in re.findall(r'"([^"]*)"|(\S+)', m.group(2))]


def read_config_file(filename):
    """boost::program_options config file as dense_stereo/main.cc:160-166 reads it (unknown keys
    allowed): `key = value`, `#` comments.  Returns a dict, or None if the file cannot be opened."""
    try:
        lines = open(filename).read().splitlines()
    except OSError:
        return None
    kv = {}
    for line in lines:
        line = line.split("#", 1)[0]
        if "=" in line:
            k, v = line.split("=", 1)
            if k.strip():
                kv[k.strip()] = v.strip()
    return kv


# ---------------------------------------------------------------- image I/O and previews
def read_gray(path):
    from PIL import Image
    return np.ascontiguousarray(np.array(Image.open(path).convert("L"), dtype=np.uint8))


def write_png16(path, disp):
    """16-bit PNG, value = 256 * disparity, 0 = invalid (what cv::imwrite does with a CV_16U Mat;
    the KITTI devkit format, reconstruction/evaluation/io_disp.h:220-238)."""
    from PIL import Image
    Image.fromarray(np.ascontiguousarray(disp, np.uint16)).save(path)


def read_png16(path):
    from PIL import Image
    return np.array(Image.open(path)).astype(np.uint16)


def to_u8_div256(disp):
    """img_disp.convertTo(img8, CV_8U, 1.0/256.0) (sgm_single/main.cc:45): float32 product, round half
    to even, saturate."""
    return np.clip(np.rint(disp.astype(np.float32) * np.float32(1.0 / 256.0)), 0, 255).astype(np.uint8)


def normalize_minmax_u8(img8):
    """cv::normalize(src, dst, 0, 255, cv::NORM_MINMAX) on CV_8U (sgm_single/main.cc:56): scale and
    shift in double, applied in float32 (two rounded operations), round half to even."""
    lo, hi = int(img8.min()), int(img8.max())
    if hi == lo:
        return np.zeros_like(img8)
    scale = 255.0 / (hi - lo)
    t = img8.astype(np.float32) * np.float32(scale)
    return np.clip(np.rint(t + np.float32(-lo * scale)), 0, 255).astype(np.uint8)


def equalize_hist(img8):
    """cv::equalizeHist on CV_8U (dense_stereo/main.cc:108, stereo_eval/main.cc:55)."""
    hist = np.bincount(img8.reshape(-1), minlength=256)
    i = int(np.flatnonzero(hist)[0])
    if hist[i] == img8.size:
        return np.full_like(img8, i)
    scale = np.float32(255.0) / np.float32(img8.size - hist[i])
    lut = np.zeros(256, np.uint8)
    csum = np.cumsum(hist[i + 1:]).astype(np.float32)
    lut[i + 1:] = np.clip(np.rint(csum * scale), 0, 255).astype(np.uint8)
    return lut[img8]


def stem(path):
    """sgm_single/main.cc:47-51: strip the directory and the last extension."""
    name = path[path.rfind("/") + 1:]
    dot = name.rfind(".")
    return name[:dot] if dot != -1 else name


def write_outputs(out_folder, prefix, disp):
    from PIL import Image
    write_png16(join(out_folder, "disparities", prefix + ".png"), disp)
    Image.fromarray(normalize_minmax_u8(to_u8_div256(disp))).save(join(out_folder, "norm_hist", prefix + ".png"))


# ---------------------------------------------------------------- single-pair mains
def sgm_single(left_img_fname, right_img_fname, out_folder, P1, P2, disp_range=256, window_sz=3, engine=None,
               device=0):
    """sgm_single left right out_folder P1 P2 (sgm_single/main.cc:11-79).  The reference hard-codes
    disp_range = 256 and window_sz = 3 (:29-32); they are options here."""
    eng = engine or StereoSGM(StereoSGMParams(disp_range=disp_range, window_sz=window_sz, penalty1=P1, penalty2=P2,
                                              device=device))
    disp = eng.compute(read_gray(left_img_fname), read_gray(right_img_fname))
    write_outputs(out_folder, stem(left_img_fname), disp)
    return disp


def sgm_cnn(left_desc_path, right_desc_path, out_dir, prefix, P1, P2, disp_range=255, engine=None, device=0):
    """sgm_cnn left right out_folder prefix P1 P2 (sgm_cnn/main.cc:11-74; disp_range = 255 at :24)."""
    eng = engine or SGM(SGMParams(disp_range=disp_range, penalty1=P1, penalty2=P2, device=device))
    disp = eng.Compute(left_desc_path, right_desc_path)
    write_outputs(out_dir, prefix, disp)
    return disp


def dense_stereo(imagelist, source_folder, output_folder, disp_range=270, window_sz=5, P1=7, P2=60, sigma=0.7, device=0,
                 write_yml=False, batch=8):
    """runDenseStereo (dense_stereo/main.cc:42-118): left/right names alternate in `imagelist`; both
    images are Gaussian-blurred (:83-85), run through StereoSGM with the mains' hard-coded parameters
    (:87-97) and saved as output_folder/img/disp_<n>.png (8-bit, saturated, equalised).  Returns the
    list of disparity maps.  The reference handles one pair after the other; here `batch` pairs go through one
    compute_batch call with the pre-filter fused in front of the census transform on the device (set_preblur),
    like the dense_stereo program (stereo_vision_amd/host/dense_stereo.cc) -- same results."""
    from PIL import Image
    batch = max(1, int(batch))
    eng = StereoSGM(StereoSGMParams(disp_range=disp_range, window_sz=window_sz, penalty1=P1, penalty2=P2, device=device,
                                    max_batch=batch))
    if sigma is not None:
        eng.set_preblur(sigma)
    os.makedirs(output_folder + "img", exist_ok=True)
    os.makedirs(output_folder + "mat", exist_ok=True)
    starts = list(range(0, len(imagelist) - 1, 2))
    out = []
    for g in range(0, len(starts), batch):
        chunk = starts[g:g + batch]
        pairs = [tuple(read_gray(source_folder + imagelist[k]) for k in (i, i + 1)) for i in chunk]
        if len({p[0].shape for p in pairs} | {p[1].shape for p in pairs}) == 1:
            disps = list(eng.compute_batch(np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs])))
        else:  # sizes differ inside the chunk: one call per pair
            disps = [eng.compute(l, r) for l, r in pairs]
        for i, disp in zip(chunk, disps):
            nm = imagelist[i]
            tag = nm[9:len(nm) - 4] if len(nm) >= 14 else stem(nm)  # substr(9, size - 13), :106
            Image.fromarray(equalize_hist(np.minimum(disp, 255).astype(np.uint8))).save(output_folder + "img/disp_" + tag + ".png")
            out.append(disp)
    eng.close()
    return out


# ---------------------------------------------------------------- RGB-D data set main
def read_calib_kitti(path):
    """core::FormatHelper::ReadCalibKitti (core/format_helper.cc:19-32,70-92): four "Pn:" rows of 12 numbers;
    returns (grey_cam, color_cam), each [fx, fy, cx, cy, baseline] with baseline = |t_left - t_right| / fx."""
    words = open(path).read().split()
    P = []
    for i in range(4):
        P.append(np.array([float(v) for v in words[13 * i + 1:13 * i + 13]], np.float64).reshape(3, 4))

    def params(pl, pr):
        diff = pl[:, 3] - pr[:, 3]
        return [pl[0, 0], pl[1, 1], pl[0, 2], pl[1, 2], float(np.sqrt((diff * diff).sum()) / pl[0, 0])]
    return params(P[0], P[1]), params(P[2], P[3])


def fix_kitti_groundtruth(rgb):
    """FixKittiGroundtruth (create_rgbd_dataset/main.cc:29-47) on an (H, W, 3) RGB image: pure red -> blue,
    everything else -> green."""
    red = (rgb[..., 0] == 255) & (rgb[..., 1] == 0) & (rgb[..., 2] == 0)
    out = np.zeros_like(rgb)
    out[..., 1] = 255
    out[red] = (0, 0, 255)
    return out


def create_rgbd_dataset(data_type, data_path, out_folder, P1=5, P2=60, disp_range=120, window_sz=5, device=0):
    """create_rgbd_dataset -train/-test data_folder out_folder (main.cc:49-158): every file of data_path/left/
    through the census SGM (disp_range 120, window 5, P1 5, P2 60; :73-83,134-135), depth in cm from the colour
    camera's calibration (:12-27,92-95), and the four outputs per frame (:104-115).  Returns the depth images."""
    from PIL import Image
    if data_type not in ("-train", "-test"):
        raise ValueError("data_type must be -train or -test")
    have_gt = data_type == "-train"
    left_dir = data_path + "/left/"
    files = sorted(f for f in os.listdir(left_dir) if not f.startswith(".") and not os.path.isdir(join(left_dir, f)))
    eng = StereoSGM(StereoSGMParams(disp_range=disp_range, window_sz=window_sz, penalty1=P1, penalty2=P2, device=device))
    out = []
    for name in files:
        print(name)
        _, color_cam = read_calib_kitti(data_path + "/calib/" + name[:-4] + ".txt")
        left, right = read_gray(left_dir + name), read_gray(data_path + "/right/" + name)
        disp = eng.compute(left, right)
        depth = eng.depth_image(disp, color_cam[0], color_cam[4])
        prefix = name[:-4]
        Image.open(left_dir + name).convert("RGB").save(out_folder + "/" + prefix + "_colors.png")
        write_png16(out_folder + "/" + prefix + "_depth.png", depth)
        Image.fromarray(equalize_hist(to_u8_div256(disp))).save(out_folder + "/" + prefix + "_depth_visualization.png")
        if have_gt:
            pos = name.find("_")
            gt_name = name[:pos] + "_road_" + name[pos + 1:]
            gt = fix_kitti_groundtruth(np.asarray(Image.open(data_path + "/groundtruth/" + gt_name).convert("RGB")))
        else:
            gt = np.zeros(left.shape + (3,), np.uint8)
        Image.fromarray(gt).save(out_folder + "/" + prefix + "_ground_truth.png")
        out.append(depth)
    return out


# ---------------------------------------------------------------- batch drivers
def run_batch_census(left_dir, right_dir, out_folder, P1=3, P2=60, disp_range=256, window_sz=3, rank=0, world=1,
                     device=0, batch=8, write=True):
    """scripts/run_sgm_batch_census.py over this rank's slice of the pair list.  Pairs of equal
    size are sent to the engine `batch` at a time.  Returns (frames, xor of the output checksums)."""
    pairs = list_census_pairs(left_dir, right_dir)
    lo, hi = shard_range(len(pairs), rank, world)
    if rank == 0 and write:
        make_output_dirs(out_folder)
    eng = StereoSGM(StereoSGMParams(disp_range=disp_range, window_sz=window_sz, penalty1=P1, penalty2=P2,
                                    device=device, max_batch=batch))
    frames, cks = 0, 0
    i = lo
    while i < hi:
        group = [(read_gray(l), read_gray(r), l) for l, r in pairs[i:min(hi, i + batch)]]
        n = 1
        while n < len(group) and group[n][0].shape == group[0][0].shape:
            n += 1
        group = group[:n]
        disp = eng.compute_batch(np.stack([g[0] for g in group]), np.stack([g[1] for g in group]))
        for g, d in zip(group, disp):
            print(os.path.basename(g[2]))
            if write:
                write_outputs(out_folder, stem(g[2]), d)
            cks ^= fnv_u16(d)
        frames += n
        i += n
    return frames, cks


def _run_pairs(eng, pairs, out_folder, batch, write):
    frames, cks = 0, 0
    i = 0
    while i < len(pairs):
        group = [(read_gray(l), read_gray(r), l) for l, r in pairs[i:i + batch]]
        n = 1
        while n < len(group) and group[n][0].shape == group[0][0].shape:
            n += 1
        group = group[:n]
        disp = eng.compute_batch(np.stack([g[0] for g in group]), np.stack([g[1] for g in group]))
        for g, d in zip(group, disp):
            print(g[2])
            if write:
                write_outputs(out_folder, stem(g[2]), d)
            cks ^= fnv_u16(d)
        frames += n
        i += n
    return frames, cks


def run_grid_search(data_folder, out_folder_prefix, P1=(4, 5, 6), P2=(40, 45, 50, 55, 60), start_num=0, stop_num=193,
                    frame_step=1, disp_range=256, window_sz=3, rank=0, world=1, device=0, batch=8, write=True, **naming):
    """scripts/run_sgm_batch_grid_search.py:36-54: for every (P1, P2) of the grid, every numbered frame through
    the census SGM binary (argv left right out_folder P1 P2, the sgm_single form) into its own result folder.
    Each rank takes a contiguous slice of the frames of every grid point.  Returns {(P1, P2): (frames, checksum)}."""
    pairs = list_numbered_pairs(data_folder, start_num, stop_num, frame_step, **naming)
    lo, hi = shard_range(len(pairs), rank, world)
    res = {}
    for penalty1 in P1:
        for penalty2 in P2:
            out_folder = grid_search_folder(out_folder_prefix, penalty1, penalty2)
            print(out_folder)
            if rank == 0 and write:
                make_output_dirs(out_folder)
            eng = StereoSGM(StereoSGMParams(disp_range=disp_range, window_sz=window_sz, penalty1=penalty1,
                                            penalty2=penalty2, device=device, max_batch=batch))
            res[(penalty1, penalty2)] = _run_pairs(eng, pairs[lo:hi], out_folder, batch, write)
            eng.close()
    return res


def run_batch_cnn(data_folder, out_folder, P1=3, P2=60, disp_range=255, rank=0, world=1, device=0, write=True):
    """scripts/run_sgm_batch_cnn.py over this rank's slice of the descriptor list."""
    items = list_cnn_pairs(data_folder)
    lo, hi = shard_range(len(items), rank, world)
    if rank == 0 and write:
        make_output_dirs(out_folder)
    eng = SGM(SGMParams(disp_range=disp_range, penalty1=P1, penalty2=P2, device=device))
    frames, cks = 0, 0
    for left, right, prefix in items[lo:hi]:
        print(os.path.basename(left))
        disp = eng.Compute(left, right)
        if write:
            write_outputs(out_folder, prefix, disp)
        cks ^= fnv_u16(disp)
        frames += 1
    return frames, cks
