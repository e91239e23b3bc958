"""CPU: the shortlist oracle against a numpy restatement of its definition, and that it does what it is
for (images of one scene choose each other).  No output of the reference exists for this step (it has no
content-based pair selection): parity unpinned."""
import numpy as np

import oracle_lib as orc
from visual_slam_amd import synth


def _numpy_shortlist(words, nwords, topk):
    nimg = len(words)
    h = np.zeros((nimg, nwords), np.int64)
    for i, w in enumerate(words):
        np.add.at(h[i], w, 1)
    df = (h > 0).sum(0)
    weight = np.zeros(nwords, np.int64)
    for w in range(nwords):
        if df[w]:
            weight[w] = 1 + int(np.floor(np.log2(nimg // df[w])))
    sc = np.zeros((nimg, nimg), np.int64)
    for i in range(nimg):
        sc[i] = (np.minimum(h[i][None, :], h) * weight[None, :]).sum(1)
    chosen = set()
    for i in range(nimg):
        order = sorted((j for j in range(nimg) if j != i and sc[i, j] > 0), key=lambda j: (-sc[i, j], j))
        for j in order[:topk]:
            chosen.add((min(i, j), max(i, j)))
    return np.array(sorted(chosen), np.int32).reshape(-1, 2), sc.astype(np.int32)


def test_oracle_equals_numpy_restatement():
    rng = np.random.default_rng(5)
    for nimg, nwords, topk in [(1, 7, 3), (2, 5, 1), (9, 16, 2), (17, 40, 5), (30, 64, 40), (12, 3, 4)]:
        words = [rng.integers(0, nwords, int(rng.integers(0, 200))).astype(np.int32) for _ in range(nimg)]
        pairs, sc = orc.shortlist(words, nwords, topk)
        epairs, esc = _numpy_shortlist(words, nwords, topk)
        np.testing.assert_array_equal(sc, esc)
        np.testing.assert_array_equal(pairs, epairs)


def test_images_of_one_scene_choose_each_other():
    images, group = synth.image_groups(6, 4, 300, seed=3, shared=0.6)
    vocab = synth.sample_vocabulary(images, 1024, seed=4)
    words = [orc.knn2_u8(img, vocab)[0] for img in images]
    pairs, _ = orc.shortlist(words, len(vocab), 3)
    same = group[pairs[:, 0]] == group[pairs[:, 1]]
    assert same.sum() == 6 * 6          # every pair inside a group of four is there
    assert same.mean() > 0.9            # and little else


def test_trained_vocabulary_lowers_the_quantisation_error():
    images, _ = synth.image_groups(3, 3, 200, seed=6, shared=0.6)
    rows = np.concatenate(images)

    def cost(words):
        return int(orc.knn2_u8(rows, words)[1].astype(np.int64).sum())

    w0 = orc.train_vocabulary(images, 64, 0)
    w5 = orc.train_vocabulary(images, 64, 5)
    np.testing.assert_array_equal(w0, rows[[k * len(rows) // 64 for k in range(64)]])
    assert w5.shape == (64, 128) and cost(w5) < 0.8 * cost(w0)
    assert len(orc.train_vocabulary(images[:1], 1000, 2)) == 200     # no more words than rows
