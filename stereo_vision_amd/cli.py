"""Command-line counterparts of the reference binaries and batch scripts.

    python -m stereo_vision_amd.cli sgm_single LEFT RIGHT OUT_FOLDER P1 P2      (sgm_single/main.cc:66-79)
    python -m stereo_vision_amd.cli sgm_cnn LEFT.bin RIGHT.bin OUT PREFIX P1 P2 (sgm_cnn/main.cc:60-74)
    python -m stereo_vision_amd.cli batch_census LEFT_DIR RIGHT_DIR OUT [P1 P2] (scripts/run_sgm_batch_census.py)
    python -m stereo_vision_amd.cli batch_cnn DATA_DIR [P1 P2]                  (scripts/run_sgm_batch_cnn.py)
    python -m stereo_vision_amd.cli grid_search DATA_DIR OUT_PREFIX [--p1 ..] [--p2 ..] (scripts/run_sgm_batch_grid_search.py)
    python -m stereo_vision_amd.cli create_rgbd_dataset -train|-test DATA_DIR OUT   (create_rgbd_dataset/main.cc:119-158)

Same positional arguments as the reference; what it hard-codes (disp_range, window_sz) are options.
The batch commands shard the pair list over the ranks of a torchrun launch (one process per GPU).
"""
import argparse
import os
import sys
import time


def _dist():
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("WORLD_SIZE", 1), ("LOCAL_RANK", 0)))
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    return rank, world, local


def main(argv=None):
    ap = argparse.ArgumentParser(prog="stereo_vision_amd.cli")
    sub = ap.add_subparsers(dest="cmd", required=True)
    a = sub.add_parser("sgm_single")
    for n in ("left", "right", "out_folder"):
        a.add_argument(n)
    a.add_argument("P1", type=int)
    a.add_argument("P2", type=int)
    a.add_argument("--disp-range", type=int, default=256)
    a.add_argument("--window-sz", type=int, default=3)
    b = sub.add_parser("sgm_cnn")
    for n in ("left", "right", "out_folder", "prefix"):
        b.add_argument(n)
    b.add_argument("P1", type=int)
    b.add_argument("P2", type=int)
    b.add_argument("--disp-range", type=int, default=255)
    c = sub.add_parser("batch_census")
    for n in ("left_dir", "right_dir", "out_folder"):
        c.add_argument(n)
    c.add_argument("P1", type=int, nargs="?", default=3)
    c.add_argument("P2", type=int, nargs="?", default=60)
    c.add_argument("--disp-range", type=int, default=256)
    c.add_argument("--window-sz", type=int, default=3)
    c.add_argument("--batch", type=int, default=8)
    d = sub.add_parser("batch_cnn")
    d.add_argument("data_dir")
    d.add_argument("P1", type=int, nargs="?", default=3)
    d.add_argument("P2", type=int, nargs="?", default=60)
    d.add_argument("--disp-range", type=int, default=255)
    g = sub.add_parser("grid_search")
    g.add_argument("data_dir")
    g.add_argument("out_prefix")
    g.add_argument("--p1", type=int, nargs="+", default=[4, 5, 6])
    g.add_argument("--p2", type=int, nargs="+", default=[40, 45, 50, 55, 60])
    g.add_argument("--start", type=int, default=0)
    g.add_argument("--stop", type=int, default=193)
    g.add_argument("--step", type=int, default=1)
    g.add_argument("--disp-range", type=int, default=256)
    g.add_argument("--window-sz", type=int, default=3)
    g.add_argument("--batch", type=int, default=8)
    r = sub.add_parser("create_rgbd_dataset", prefix_chars="+")
    for n in ("data_type", "data_dir", "out_folder"):
        r.add_argument(n)
    args = ap.parse_args(argv)

    from . import drivers
    if args.cmd == "sgm_single":
        drivers.sgm_single(args.left, args.right, args.out_folder, args.P1, args.P2, args.disp_range, args.window_sz)
        return 0
    if args.cmd == "sgm_cnn":
        drivers.sgm_cnn(args.left, args.right, args.out_folder, args.prefix, args.P1, args.P2, args.disp_range)
        return 0
    if args.cmd == "create_rgbd_dataset":
        if args.data_type not in ("-train", "-test"):
            return 1  # main.cc:131-136
        drivers.create_rgbd_dataset(args.data_type, args.data_dir, args.out_folder)
        return 0
    rank, world, local = _dist()
    t0 = time.perf_counter()
    if args.cmd == "grid_search":
        res = drivers.run_grid_search(args.data_dir, args.out_prefix, args.p1, args.p2, args.start, args.stop, args.step,
                                      args.disp_range, args.window_sz, rank, world, local, args.batch)
        frames = sum(v[0] for v in res.values())
        cks = 0
        for v in res.values():
            cks ^= v[1]
    elif args.cmd == "batch_census":
        frames, cks = drivers.run_batch_census(args.left_dir, args.right_dir, args.out_folder, args.P1, args.P2,
                                               args.disp_range, args.window_sz, rank, world, local, args.batch)
    else:  # run_sgm_batch_cnn.py:22-23: representation/ in, depth/ out
        frames, cks = drivers.run_batch_cnn(os.path.join(args.data_dir, "representation"),
                                            os.path.join(args.data_dir, "depth"), args.P1, args.P2, args.disp_range,
                                            rank, world, local)
    st = drivers.gather_stats(frames, time.perf_counter() - t0, cks)
    if rank == 0:
        print("pairs: %d  ranks: %d  wall: %.3f s  %.2f pairs/s  checksum xor: %016x" %
              (st["frames"], st["world"], st["seconds"], st["frames"] / max(st["seconds"], 1e-9), st["checksum_xor"]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
