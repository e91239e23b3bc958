"""End to end on the GPU: .sift files in -> vsm_match_pairs (C++ driver over the C ABI) ->
matching.txt / .mat out, compared with what the CPU oracle produces for the same files."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as orc
import visual_slam_amd as vsm
from visual_slam_amd import host, synth

pytestmark = pytest.mark.gpu
CLI = os.path.join(os.path.dirname(vsm.lib_path()), "vsm_match_pairs")


def _make_images(tmp_path, sizes, seed):
    rng = np.random.default_rng(seed)
    pool = synth.sift_like(600, rng)
    imgs = []
    for i, n in enumerate(sizes):
        des = synth.sift_like(n, rng)
        k = min(n, 600) // 2
        if k:
            pick = rng.choice(600, k, replace=False)
            des[rng.choice(n, k, replace=False)] = np.clip(
                pool[pick].astype(np.int32) + rng.integers(-5, 6, (k, 128)), 0, 255).astype(np.uint8)
        loc = rng.random((max(n, 1), 5)).astype(np.float32)[:n]
        if n:
            host.write_sift(str(tmp_path / ("%d.sift" % i)), loc, des)
        imgs.append(des)
    return imgs


@pytest.mark.parametrize("mode", ["exhaustive", "window"])
def test_cli_matches_oracle(tmp_path, mode):
    sizes = [700, 513, 0, 900, 64, 1100] if mode == "exhaustive" else [300] * 18
    imgs = _make_images(tmp_path, sizes, seed=21)
    out = str(tmp_path / "matching.txt")
    cmd = [CLI, "--sift-dir", str(tmp_path), "--num", str(len(sizes)), "--out", out, "--pairs", mode,
           "--batch", "7", "--mat"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    if mode == "exhaustive":
        pairs = [(i, j) for i in range(len(sizes)) for j in range(i + 1, len(sizes))]
    else:
        pairs = [tuple(p) for p in orc.image_pairs(len(sizes), 3, [7, 15])]
    expect = []
    for a, b in pairs:
        m0, _ = orc.match_dense_u8(imgs[a], imgs[b], orc.params((4, 5), True))
        expect.append(orc.matchtxt_record("%d.jpg" % a, "%d.jpg" % b, m0))
        i0 = np.nonzero(m0 >= 0)[0]
        ga, gb = host.get_pmatch(str(tmp_path / str(a)), str(tmp_path / str(b)), len(imgs[b]))
        np.testing.assert_array_equal(ga, i0)
        np.testing.assert_array_equal(gb, m0[i0])
    assert open(out).read() == "".join(expect)
    assert sum(len(e.splitlines()[1].split()) for e in expect) > 100


@pytest.mark.parametrize("pair_args", [[], ["--pairs", "vocab", "--topk", "2", "--vocab-size", "300", "--kmeans", "2"]])
def test_sharded_runs_concatenate_to_the_unsharded_file(tmp_path, pair_args):
    """Three processes on contiguous slices of the pair list (here all on GPU 0) = one process on all of it;
    with a vocabulary pair list every process derives the same list before it takes its slice."""
    import sys
    sizes = [400, 300, 513, 256, 700, 120, 333]
    _make_images(tmp_path, sizes, seed=22)
    whole = str(tmp_path / "whole.txt")
    res = subprocess.run([CLI, "--sift-dir", str(tmp_path), "--num", str(len(sizes)), "--out", whole, "--batch", "5"] + pair_args,
                         capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    sharded = str(tmp_path / "sharded.txt")
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "match_multi_gpu.py")
    res = subprocess.run([sys.executable, tool, "--sift-dir", str(tmp_path), "--num", str(len(sizes)), "--out", sharded,
                          "--devices", "0,0,0", "--batch", "4"] + pair_args, capture_output=True, text=True, timeout=180)
    assert res.returncode == 0, res.stderr
    assert open(sharded, "rb").read() == open(whole, "rb").read()
    assert os.path.getsize(whole) > 1000


def _read_records(path):
    lines = open(path).read().split("\n")
    recs = []
    for k in range(0, len(lines) - 1, 3):
        name0, name1, n = lines[k].split()
        recs.append((name0, name1, np.array(lines[k + 1].split(), np.int32), np.array(lines[k + 2].split(), np.int32)))
        assert len(recs[-1][2]) == int(n)
    return recs


def test_cli_verifies_matches_on_the_device(tmp_path):
    """--verify: the inlier lists and matrices the driver writes (match txt and .mat inlier records) are
    the ones the oracle derives from the putative lists of the same run."""
    nimg, n, batch = 4, 500, 4
    kps, dess = synth.multi_view(nimg, n, seed=31)
    for i in range(nimg):
        loc = np.zeros((n, 5), np.float32)
        loc[:, :2] = kps[i]
        loc[:, 3] = 1.0
        host.write_sift(str(tmp_path / ("%d.sift" % i)), loc, dess[i])
    out, inl = str(tmp_path / "matching.txt"), str(tmp_path / "inliers.txt")
    cmd = [CLI, "--sift-dir", str(tmp_path), "--num", str(nimg), "--out", out, "--batch", str(batch), "--mat",
           "--verify", "1.5", "--samples", "300", "--seed", "5", "--inlier-out", inl, "--min-inliers", "12"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    put, ver = _read_records(out), _read_records(inl)
    assert len(put) == len(ver) == nimg * (nimg - 1) // 2
    kept = 0
    for k, ((a0, a1, p0, p1), (b0, b1, v0, v1)) in enumerate(zip(put, ver)):
        assert (a0, a1) == (b0, b1)
        ia, ib = int(a0.split(".")[0]), int(a1.split(".")[0])
        F, o0, o1 = orc.geom_pair(kps[ia], kps[ib], p0, p1, max_sampson=1.5, hypotheses=300, seed=5, pair_pos=k,
                                  min_inliers=12)
        np.testing.assert_array_equal(v0, o0)
        np.testing.assert_array_equal(v1, o1)
        nm, Fm, r0, r1 = host.get_imatch(str(tmp_path / str(ia)), str(tmp_path / str(ib)), n)
        assert nm == len(p0)
        np.testing.assert_array_equal(Fm, F)
        np.testing.assert_array_equal(r0, o0)
        np.testing.assert_array_equal(r1, o1)
        g0, g1 = host.get_pmatch(str(tmp_path / str(ia)), str(tmp_path / str(ib)), n)   # putative lists intact
        np.testing.assert_array_equal(g0, p0)
        kept += len(o0)
        assert len(p0) > 150 and len(o0) > 0.9 * len(p0)      # nearly every match shows a common scene point
    assert "geometric verification: %d pairs" % len(put) in res.stderr and kept > 1000


def test_cli_writes_the_relative_pose(tmp_path):
    """--camera: NE, R, T of every verified pair in the .mat records, equal to the oracle's pose from the F and
    the inlier lists of the same run; the partner's file holds the exchanged view."""
    nimg, n = 3, 500
    kps, dess = synth.multi_view(nimg, n, seed=35)
    for i in range(nimg):
        loc = np.zeros((n, 5), np.float32)
        loc[:, :2] = kps[i]
        loc[:, 3] = 1.0
        host.write_sift(str(tmp_path / ("%d.sift" % i)), loc, dess[i])
    out, inl = str(tmp_path / "matching.txt"), str(tmp_path / "inliers.txt")
    cmd = [CLI, "--sift-dir", str(tmp_path), "--num", str(nimg), "--out", out, "--mat", "--verify", "1.5",
           "--samples", "300", "--seed", "5", "--inlier-out", inl, "--camera", "1200,1200,800,600"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    assert "relative pose: 3 pairs" in res.stderr
    cam = np.asarray([1200, 1200, 800, 600] * 2, np.float32)
    for (a0, a1, v0, v1) in _read_records(inl):
        ia, ib = int(a0.split(".")[0]), int(a1.split(".")[0])
        nm, F, r0, r1 = host.get_imatch(str(tmp_path / str(ia)), str(tmp_path / str(ib)), n)
        oR, oT, one = orc.pose_pair(kps[ia], kps[ib], v0, v1, F, cam)
        nm2, ne, R, T, f12 = host.get_pose(str(tmp_path / str(ia)), str(tmp_path / str(ib)), n)
        assert nm2 == nm and ne == one and one > 100 and f12 == (1200.0, 1200.0)
        np.testing.assert_array_equal(R, oR)
        np.testing.assert_array_equal(T, oT)
        _, ne_b, Rb, Tb, _ = host.get_pose(str(tmp_path / str(ib)), str(tmp_path / str(ia)), n)
        assert ne_b == one
        np.testing.assert_array_equal(Rb, oR.T)
        np.testing.assert_allclose(Tb, -(oR.T.astype(np.float64) @ oT), atol=1e-6)


def test_sharded_verification_equals_the_unsharded_run(tmp_path):
    """The RANSAC samples are keyed by a pair's position in the whole pair list, so cutting the list into
    shards and submissions in another way gives the same inlier file byte for byte."""
    import sys
    nimg = 5
    kps, dess = synth.multi_view(nimg, 400, seed=33)
    for i in range(nimg):
        loc = np.zeros((400, 5), np.float32)
        loc[:, :2] = kps[i]
        host.write_sift(str(tmp_path / ("%d.sift" % i)), loc, dess[i])
    whole_in = str(tmp_path / "whole_in.txt")
    res = subprocess.run([CLI, "--sift-dir", str(tmp_path), "--num", str(nimg), "--out", str(tmp_path / "whole.txt"),
                          "--batch", "4", "--verify", "2.0", "--seed", "3", "--inlier-out", whole_in],
                         capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "match_multi_gpu.py")
    shard_in = str(tmp_path / "shard_in.txt")
    res = subprocess.run([sys.executable, tool, "--sift-dir", str(tmp_path), "--num", str(nimg), "--out", str(tmp_path / "shard.txt"),
                          "--devices", "0,0,0", "--batch", "3", "--verify", "2.0", "--seed", "3", "--inlier-out", shard_in],
                         capture_output=True, text=True, timeout=180)
    assert res.returncode == 0, res.stderr
    assert open(shard_in, "rb").read() == open(whole_in, "rb").read()
    assert open(str(tmp_path / "shard.txt"), "rb").read() == open(str(tmp_path / "whole.txt"), "rb").read()
    assert os.path.getsize(whole_in) > 5000


@pytest.mark.parametrize("own_words", [True, False, "kmeans", "kmeans-sampled"])
def test_cli_pairs_from_a_vocabulary(tmp_path, own_words):
    """--pairs vocab: the pair list is the shortlist the oracle computes from the same words (drawn from
    the images by the driver's rule, or read from a .sift file), and every listed pair is matched."""
    images, group = synth.image_groups(4, 3, 500, seed=31, shared=0.6)
    rng = np.random.default_rng(1)
    for i, des in enumerate(images):
        host.write_sift(str(tmp_path / ("%d.sift" % i)), rng.random((len(des), 5)).astype(np.float32), des)
    num, topk, nwords = len(images), 2, 700
    out = str(tmp_path / "matching.txt")
    cmd = [CLI, "--sift-dir", str(tmp_path), "--num", str(num), "--out", out, "--pairs", "vocab", "--topk", str(topk)]
    if own_words == "kmeans-sampled":
        cmd += ["--vocab-size", str(nwords), "--kmeans", "2", "--kmeans-stride", "3"]
        vocab = orc.train_vocabulary([np.concatenate(images)[::3]], nwords, 2)
    elif own_words == "kmeans":
        cmd += ["--vocab-size", str(nwords), "--kmeans", "3"]
        vocab = orc.train_vocabulary(images, nwords, 3)
    elif own_words:
        cmd += ["--vocab-size", str(nwords)]
        words = []
        for i, des in enumerate(images):
            q = min(len(des), nwords // num + (1 if i < nwords % num else 0))
            words += [des[j * len(des) // q] for j in range(q)]
        vocab = np.asarray(words, np.uint8)
    else:
        vocab = synth.sample_vocabulary(images, nwords, seed=32)
        host.write_sift(str(tmp_path / "words.sift"), np.zeros((nwords, 5), np.float32), vocab)
        cmd += ["--vocab", str(tmp_path / "words.sift")]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    pairs, _ = orc.shortlist([orc.knn2_u8(img, vocab)[0] for img in images], len(vocab), topk)
    assert (group[pairs[:, 0]] == group[pairs[:, 1]]).sum() == 4 * 3
    assert ("%d pairs" % len(pairs)) in res.stderr
    expect = []
    for a, b in pairs:
        m0, _ = orc.match_dense_u8(images[a], images[b], orc.params((4, 5), True))
        expect.append(orc.matchtxt_record("%d.jpg" % a, "%d.jpg" % b, m0))
    assert open(out).read() == "".join(expect)


def test_cli_explicit_pair_list(tmp_path):
    """--pairs-file: pairs in file order, repeats and (j, i) after (i, j) allowed; bad lists are refused."""
    sizes = [300, 200, 0, 450, 128]
    imgs = _make_images(tmp_path, sizes, seed=23)
    wanted = [(3, 0), (0, 3), (1, 4), (1, 4), (2, 1), (4, 4)]
    lst = tmp_path / "pairs.txt"
    lst.write_text("".join("%d %d\n" % p for p in wanted))
    out = str(tmp_path / "m.txt")
    res = subprocess.run([CLI, "--sift-dir", str(tmp_path), "--num", "5", "--out", out, "--pairs-file", str(lst), "--batch", "4"],
                         capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    expect = []
    for a, b in wanted:
        m0, _ = orc.match_dense_u8(imgs[a], imgs[b], orc.params((4, 5), True))
        expect.append(orc.matchtxt_record("%d.jpg" % a, "%d.jpg" % b, m0))
    assert open(out).read() == "".join(expect)
    for bad in ("0 9\n", "0 1\n2\n", "a b\n"):
        lst.write_text(bad)
        res = subprocess.run([CLI, "--sift-dir", str(tmp_path), "--num", "5", "--out", out, "--pairs-file", str(lst)],
                             capture_output=True, text=True, timeout=120)
        assert res.returncode == 1, bad


def test_several_contexts_in_one_process_write_the_single_context_files(tmp_path):
    """--devices: a thread and a context per entry take the submissions as they come, one writer thread emits them in
    pair order: match txt, inlier txt and every .mat record (putative, inlier and pose) are byte for byte those of
    the run with one context.  Here both contexts sit on GPU 0."""
    nimg, n = 6, 400
    kps, dess = synth.multi_view(nimg, n, seed=41)
    for i in range(nimg):
        loc = np.zeros((n, 5), np.float32)
        loc[:, :2] = kps[i]
        loc[:, 3] = 1.0
        host.write_sift(str(tmp_path / ("%d.sift" % i)), loc, dess[i])

    def run(devices, batch):
        for i in range(nimg):
            mat = tmp_path / ("%d.mat" % i)
            if mat.exists():
                mat.unlink()
        out, inl = str(tmp_path / "matching.txt"), str(tmp_path / "inliers.txt")
        cmd = [CLI, "--sift-dir", str(tmp_path), "--num", str(nimg), "--out", out, "--batch", str(batch), "--mat",
               "--verify", "1.5", "--samples", "200", "--seed", "9", "--inlier-out", inl, "--camera", "1200,1200,800,600",
               "--devices", devices]
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=180)
        assert res.returncode == 0, res.stderr
        files = {"matching": open(out, "rb").read(), "inliers": open(inl, "rb").read()}
        for i in range(nimg):
            files["%d.mat" % i] = open(str(tmp_path / ("%d.mat" % i)), "rb").read()
        return files, res.stderr

    one, log1 = run("0", 4)
    two, log2 = run("0,0", 2)          # eight submissions dealt to two contexts
    three, _ = run("0,0,0", 1)
    assert "on 1 context;" in log1 and "on 2 contexts;" in log2
    assert set(one) == set(two) == set(three)
    for name in one:
        assert one[name] == two[name], name
        assert one[name] == three[name], name
    assert len(one["matching"]) > 5000 and len(one["0.mat"]) > 5000


def test_sharded_processes_write_the_same_mat_files_under_locks(tmp_path):
    """--mat with --shard: three processes (all on GPU 0 here) match their slices of the pair list at the same time and
    write putative and inlier records into the same .mat files, each update under a lock on the file: afterwards every
    pair's records read back as the oracle's lists, from both images' files."""
    import sys
    nimg, n = 5, 300
    kps, dess = synth.multi_view(nimg, n, seed=51)
    for i in range(nimg):
        loc = np.zeros((n, 5), np.float32)
        loc[:, :2] = kps[i]
        host.write_sift(str(tmp_path / ("%d.sift" % i)), loc, dess[i])
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "match_multi_gpu.py")
    inl = str(tmp_path / "inliers.txt")
    res = subprocess.run([sys.executable, tool, "--sift-dir", str(tmp_path), "--num", str(nimg), "--out", str(tmp_path / "m.txt"),
                          "--devices", "0,0,0", "--batch", "2", "--mat", "--verify", "2.0", "--seed", "3", "--inlier-out", inl],
                         capture_output=True, text=True, timeout=240)
    assert res.returncode == 0, res.stderr
    ver = {(r[0], r[1]): r for r in _read_records(inl)}
    for a in range(nimg):
        for b in range(a + 1, nimg):
            m0, _ = orc.match_dense_u8(dess[a], dess[b], orc.params((4, 5), True))
            i0 = np.nonzero(m0 >= 0)[0]
            ga, gb = host.get_pmatch(str(tmp_path / str(a)), str(tmp_path / str(b)), n)
            np.testing.assert_array_equal(ga, i0)
            np.testing.assert_array_equal(gb, m0[i0])
            hb, ha = host.get_pmatch(str(tmp_path / str(b)), str(tmp_path / str(a)), n)
            np.testing.assert_array_equal(ha, i0)
            _, _, v0, v1 = ver[("%d.jpg" % a, "%d.jpg" % b)]
            nm, Fm, r0, r1 = host.get_imatch(str(tmp_path / str(a)), str(tmp_path / str(b)), n)
            assert nm == len(i0) and len(v0) > 50
            np.testing.assert_array_equal(r0, v0)
            np.testing.assert_array_equal(r1, v1)
