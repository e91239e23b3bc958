"""Exhaustive (or windowed) matching of a .sift directory on several GPUs: one vsm_match_pairs process
per GPU, each on a contiguous slice of the pair list (no communication between them), then the
per-slice match-txt parts are concatenated in slice order, which is the global pair order.

usage: match_multi_gpu.py --sift-dir DIR --num N --out matching.txt [--devices 0,1,2,3]
                          [--verify PIXELS --inlier-out verified.txt] [other vsm_match_pairs options ...]
"""
import argparse
import os
import shutil
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
CLI = os.path.join(os.path.dirname(HERE), "visual_slam_amd", "vsm_match_pairs")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sift-dir", required=True)
    ap.add_argument("--num", type=int, required=True)
    ap.add_argument("--out", default="matching.txt")
    ap.add_argument("--devices", default="0", help="comma-separated GPU ids, one process each")
    ap.add_argument("--inlier-out", default="", help="with --verify: match txt of the verified matches")
    args, rest = ap.parse_known_args()
    if args.inlier_out:
        rest += ["--inlier-out", args.inlier_out]
    devices = [int(d) for d in args.devices.split(",")]
    w = len(devices)
    t0 = time.perf_counter()
    procs = []
    for r, dev in enumerate(devices):
        cmd = [CLI, "--sift-dir", args.sift_dir, "--num", str(args.num), "--out", args.out, "--device", str(dev),
               "--shard", "%d/%d" % (r, w)] + rest
        procs.append(subprocess.Popen(cmd))
    rc = 0
    for p in procs:
        rc = rc or p.wait()
    if rc:
        sys.exit(rc)
    for path in (args.out, args.inlier_out):
        if w > 1 and path and path != "/dev/null":
            with open(path, "wb") as out:
                for r in range(w):
                    part = "%s.part%d" % (path, r)
                    with open(part, "rb") as f:
                        shutil.copyfileobj(f, out, 16 << 20)
                    os.remove(part)
    print("%d processes, %.3f s" % (w, time.perf_counter() - t0))


if __name__ == "__main__":
    main()
