"""KITTI stereo evaluation of a result directory: counterpart of reconstruction/evaluation/evaluate_stereo.cc.

    reference                                               here
    eval(result_dir, kitti_errors)   evaluate_stereo.cc:121-307   eval_stereo(result_dir, gt_folder)
    statMean / statMin / statMax     utils.h:22-41                 stat_mean / stat_min / stat_max
    main: one line appended to stats.txt  evaluate_stereo.cc:309-327   append_stats_line

The per-image work (background interpolation, outlier and average errors against the non-occluded and
the full ground truth) runs on the GPU through sgmx_kitti_eval_u16; what is left here is file handling
and the reference's float32 statistics over the images.
"""
import os

import numpy as np

from .drivers import read_png16
from .sgm import Engine, StereoSGMParams, _ptr


def evaluate_images(disp, gt_noc, gt_occ, engine=None, want_ipol=False, device=0):
    """(n, H, W) uint16 stacks in the KITTI format -> (n, 26) float32:
    noc_out[11], occ_out[11], noc_avg[2], occ_avg[2] per image (evaluate_stereo.cc:158-175)."""
    disp, gt_noc, gt_occ = (np.ascontiguousarray(a, np.uint16) for a in (disp, gt_noc, gt_occ))
    if disp.ndim == 2:
        disp, gt_noc, gt_occ = disp[None], gt_noc[None], gt_occ[None]
    if disp.shape != gt_noc.shape or disp.shape != gt_occ.shape:
        raise ValueError("Wrong file size!")  # evaluate_stereo.cc:24-27
    n, H, W = disp.shape
    eng = engine or Engine(StereoSGMParams(disp_range=16, window_sz=1, device=device))
    out = np.empty((n, 26), np.float32)
    ipol = np.empty_like(disp) if want_ipol else None
    eng.check(eng.lib.sgmx_kitti_eval_u16(eng.h, n, _ptr(disp), _ptr(gt_noc), _ptr(gt_occ), W, H, _ptr(out), _ptr(ipol)))
    return (out, ipol) if want_ipol else out


def stat_mean(errors, idx):
    """utils.h:22-27: float accumulation in image order, divided by the float count."""
    m = np.float32(0)
    for row in errors:
        m = np.float32(m + np.float32(row[idx]))
    return np.float32(m / np.float32(len(errors)))


def stat_min(errors, idx):
    """utils.h:29-34 (starts from 1)."""
    return np.float32(min([np.float32(1)] + [np.float32(r[idx]) for r in errors]))


def stat_max(errors, idx):
    """utils.h:36-41 (starts from 0)."""
    return np.float32(max([np.float32(0)] + [np.float32(r[idx]) for r in errors]))


def eval_stereo(result_dir, gt_folder, count=200, engine=None, write_interpolation=False):
    """eval() of evaluate_stereo.cc:121-307 for the files %06d_10.png, i < count, of result_dir against
    gt_folder/disp_noc_0 and gt_folder/disp_occ_0.  Writes result_dir/../stats_{noc,occ}_{out,avg}.txt
    (mean, min, max rows) and returns the five numbers of struct Errors, or None where the reference
    returns false (a file that cannot be read, wrong size, ground truth without valid pixels)."""
    disp, noc, occ = [], [], []
    try:
        for i in range(count):
            prefix = "%06d_10" % i
            n = read_png16(os.path.join(gt_folder, "disp_noc_0", prefix + ".png"))
            o = read_png16(os.path.join(gt_folder, "disp_occ_0", prefix + ".png"))
            image_file = os.path.join(result_dir, prefix + ".png")
            print(image_file)
            d = read_png16(image_file)
            if d.shape != n.shape:
                print("ERROR: Input must be png, 1 channel, 16 bit, %d x %d px" % (n.shape[1], n.shape[0]))
                return None
            disp.append(d), noc.append(n), occ.append(o)
    except (OSError, ValueError):
        return None
    rows = []
    ipols = []
    # images of one size go to the GPU together
    i = 0
    try:
        while i < len(disp):
            j = i
            while j < len(disp) and disp[j].shape == disp[i].shape:
                j += 1
            r = evaluate_images(np.stack(disp[i:j]), np.stack(noc[i:j]), np.stack(occ[i:j]), engine, write_interpolation)
            if write_interpolation:
                ipols += list(r[1])
                r = r[0]
            rows += list(r)
            i = j
    except Exception:  # SgmxError: ground truth without a valid pixel -> the reference throws and eval returns false
        return None
    if write_interpolation:
        from .drivers import write_png16
        os.makedirs(os.path.join(result_dir, "..", "interpolation"), exist_ok=True)
        for k, ip in enumerate(ipols):
            write_png16(os.path.join(result_dir, "..", "interpolation", "%06d_10.png" % k), ip)
    parts = {"noc_out": [r[0:11] for r in rows], "occ_out": [r[11:22] for r in rows],
             "noc_avg": [r[22:24] for r in rows], "occ_avg": [r[24:26] for r in rows]}
    for name, errs in parts.items():
        with open(os.path.join(result_dir, "..", "stats_%s.txt" % name), "w") as f:
            for stat in (stat_mean, stat_min, stat_max):
                f.write("".join("%f " % stat(errs, k) for k in range(len(errs[0]))) + "\n")
    return dict(out_noc=100.0 * float(stat_mean(parts["noc_out"], 4)), out_all=100.0 * float(stat_mean(parts["occ_out"], 4)),
                avg_noc=float(stat_mean(parts["noc_avg"], 0)), avg_all=float(stat_mean(parts["occ_avg"], 0)),
                density=100.0 * float(stat_mean(parts["noc_out"], 10)))


def append_stats_line(result_dir, e):
    """main() of evaluate_stereo.cc:320-326: one tab-separated line appended to result_dir/../stats.txt."""
    with open(os.path.join(result_dir, "..", "stats.txt"), "a") as f:
        f.write("%.2f\t%.2f\t%.2f\t%.2f\t%.2f\t\t%s\n" % (e["out_noc"], e["out_all"], e["avg_noc"], e["avg_all"], e["density"],
                                                          result_dir))
