"""Times vocabulary training and the shortlist on a synthetic collection (GPU box)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import visual_slam_amd as vsm  # noqa: E402
from visual_slam_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--groups", type=int, default=100)
    ap.add_argument("--per-group", type=int, default=10)
    ap.add_argument("--features", type=int, default=8192)
    ap.add_argument("--words", type=int, default=4096)
    ap.add_argument("--iterations", type=int, default=10)
    ap.add_argument("--topk", type=int, default=20)
    args = ap.parse_args()
    images, group = synth.image_groups(args.groups, args.per_group, args.features, seed=1, shared=0.5)
    m = vsm.Matcher(0)
    ids = [m.upload_u8(img) for img in images]
    for name, words in (("drawn", None), ("trained", args.iterations)):
        t0 = time.perf_counter()
        vocab = m.train_vocabulary(ids, args.words, 0 if words is None else words)
        t1 = time.perf_counter()
        vid = m.upload_u8(vocab)
        pairs = m.shortlist_pairs(ids, vid, args.topk)
        t2 = time.perf_counter()
        same = group[pairs[:, 0]] == group[pairs[:, 1]]
        want = args.groups * args.per_group * (args.per_group - 1) // 2
        print("%s words: vocabulary %.1f ms, shortlist %.1f ms, %d pairs, %d of %d same-scene pairs listed, %.1f %% of the list same-scene"
              % (name, (t1 - t0) * 1e3, (t2 - t1) * 1e3, len(pairs), int(same.sum()), want, 100.0 * same.mean()))
        m.release(vid)
    m.close()


if __name__ == "__main__":
    main()
