"""GPU parity: the image-pair shortlist through the C ABI against the CPU oracle.  Bar: pair list and
score matrix identical (integer arithmetic on word assignments that are themselves bit-exact nearest
neighbours).  The reference has no content-based pair selection: parity unpinned."""
import numpy as np
import pytest

import oracle_lib as orc
import visual_slam_amd as vsm
from visual_slam_amd import synth

pytestmark = pytest.mark.gpu


def _words(images, vocab):
    if images[0].dtype == np.uint8:
        return [orc.knn2_u8(img, vocab)[0] if len(img) else np.zeros(0, np.int32) for img in images]
    return [orc.knn2_f32(img, vocab, orc.METRIC_L2)[0] if len(img) else np.zeros(0, np.int32) for img in images]


def _check(matcher, images, vocab, topk):
    up = matcher.upload_u8 if vocab.dtype == np.uint8 else matcher.upload_f32
    ids = [up(img) for img in images]
    vid = up(vocab)
    try:
        pairs, scores = matcher.shortlist_pairs(ids, vid, topk, want_scores=True)
    finally:
        for s in ids + [vid]:
            matcher.release(s)
    epairs, escores = orc.shortlist(_words(images, vocab), len(vocab), topk)
    np.testing.assert_array_equal(scores, escores)
    np.testing.assert_array_equal(pairs, epairs)
    return pairs


def test_groups_of_sift_images(matcher):
    images, group = synth.image_groups(8, 5, 700, seed=11, shared=0.5)
    vocab = synth.sample_vocabulary(images, 1000, seed=12)
    pairs = _check(matcher, images, vocab, 4)
    same = group[pairs[:, 0]] == group[pairs[:, 1]]
    assert same.sum() == 8 * 10 and same.mean() > 0.9


@pytest.mark.parametrize("nimg,n,nwords,topk", [(1, 50, 10, 3), (2, 33, 1, 1), (3, 600, 513, 2), (70, 90, 97, 5),
                                                  (130, 40, 64, 200), (65, 257, 2048, 7)])
def test_shapes_sift(matcher, nimg, n, nwords, topk):
    rng = np.random.default_rng(nimg * 1000 + n)
    images = [synth.sift_like(max(int(rng.integers(n // 2, n + 1)), 1), rng) for _ in range(nimg)]
    # a few near-copies so that equal scores and shared words occur
    for k in range(1, nimg, 3):
        images[k] = images[k - 1].copy()
    vocab = synth.sample_vocabulary(images, nwords, seed=n)
    _check(matcher, images, vocab, topk)


def test_empty_image_and_duplicate_words(matcher):
    rng = np.random.default_rng(2)
    images = [synth.sift_like(100, rng), np.zeros((0, 128), np.uint8), synth.sift_like(80, rng), synth.sift_like(100, rng)]
    vocab = synth.sift_like(20, rng)
    vocab[7] = vocab[3]          # equal words: the lower row takes all
    pairs = _check(matcher, images, vocab, 2)
    assert not (pairs == 1).any()   # the empty image scores 0 with everything


@pytest.mark.parametrize("dim", [256, 64])
def test_float_descriptors(matcher, dim):
    images, group = synth.image_groups(5, 4, 400, seed=21, shared=0.5, noise=0.01, dtype="f32", dim=dim)
    vocab = synth.sample_vocabulary(images, 1000, seed=22)
    pairs = _check(matcher, images, vocab, 3)
    same = group[pairs[:, 0]] == group[pairs[:, 1]]
    assert same.sum() == 5 * 6


def test_rejects_bad_arguments(matcher):
    rng = np.random.default_rng(3)
    a = matcher.upload_u8(synth.sift_like(10, rng))
    f = matcher.upload_f32(synth.unit_f32(10, 256, rng))
    with pytest.raises(vsm.VsmError):
        matcher.shortlist_pairs([a, f], a, 2)        # mixed kinds
    with pytest.raises(vsm.VsmError):
        matcher.shortlist_pairs([a, a], 999, 2)      # no such vocabulary
    with pytest.raises(vsm.VsmError):
        matcher.shortlist_pairs([a, a], a, 0)        # topk < 1
    assert len(matcher.shortlist_pairs([], a, 2)) == 0
    # the context still works afterwards
    assert len(matcher.shortlist_pairs([a, a], a, 1)) == 1


def test_vocabulary_larger_than_the_lds_histogram(matcher):
    rng = np.random.default_rng(8)
    images = [synth.sift_like(300, rng) for _ in range(5)]
    images[3] = images[1].copy()
    vocab = synth.sift_like(17000, rng)
    vocab[:15000:50] = images[1]           # 300 words that are hit exactly
    pairs = _check(matcher, images, vocab, 1)
    assert [1, 3] in pairs.tolist()


def test_thousands_of_small_images(matcher):
    rng = np.random.default_rng(9)
    vocab = synth.sift_like(40, rng)
    images = [synth.sift_like(int(rng.integers(1, 9)), rng) for _ in range(4100)]
    _check(matcher, images, vocab, 3)


@pytest.mark.parametrize("sizes,nwords,iterations", [([300, 0, 200, 513], 50, 0), ([300, 0, 200, 513], 50, 4),
                                                      ([40], 100, 3), ([700] * 6, 257, 6), ([5, 3], 1, 2)])
def test_trained_vocabulary_equals_the_oracle(matcher, sizes, nwords, iterations):
    rng = np.random.default_rng(sum(sizes) + nwords)
    pool = synth.sift_like(150, rng)
    images = []
    for n in sizes:
        img = synth.sift_like(n, rng)
        k = n // 2          # half of every image: noisy copies of a few pool rows, so that words attract many rows
        img[:k] = np.clip(pool[rng.integers(0, 150, k)].astype(np.int32) + rng.integers(-4, 5, (k, 128)), 0, 255)
        images.append(img)
    ids = [matcher.upload_u8(img) for img in images]
    try:
        words = matcher.train_vocabulary(ids, nwords, iterations)
    finally:
        for s in ids:
            matcher.release(s)
    np.testing.assert_array_equal(words, orc.train_vocabulary(images, nwords, iterations))


def test_train_vocabulary_rejects_float_sets(matcher):
    f = matcher.upload_f32(synth.unit_f32(10, 256, np.random.default_rng(1)))
    with pytest.raises(vsm.VsmError):
        matcher.train_vocabulary([f], 4, 1)
    assert len(matcher.train_vocabulary([], 4, 1)) == 0


def test_small_scratch_budget_cuts_the_assignment_into_groups(matcher):
    rng = np.random.default_rng(12)
    images = [synth.sift_like(int(rng.integers(500, 900)), rng) for _ in range(40)]
    for k in range(1, 40, 4):
        images[k][:300] = images[k - 1][:300]
    matcher.set_scratch_budget(1 << 20)       # about 6500 descriptor rows per submission
    try:
        ids = [matcher.upload_u8(img) for img in images]
        words = matcher.train_vocabulary(ids, 64, 2)
        for s in ids:
            matcher.release(s)
        np.testing.assert_array_equal(words, orc.train_vocabulary(images, 64, 2))
        _check(matcher, images, words, 3)
        tiny = [synth.sift_like(4, rng) for _ in range(300)]
        ids = [matcher.upload_u8(img) for img in tiny]
        big = matcher.upload_u8(synth.sift_like(1000, rng))
        with pytest.raises(vsm.VsmError):
            matcher.shortlist_pairs(ids, big, 2)     # 300 x 1000 counters alone pass the budget
        for s in ids + [big]:
            matcher.release(s)
    finally:
        matcher.set_scratch_budget(8 << 30)


def test_float_descriptors_on_the_f32_only_path(matcher):
    images, _ = synth.image_groups(3, 3, 200, seed=41, shared=0.5, noise=0.01, dtype="f32", dim=128)
    vocab = synth.sample_vocabulary(images, 150, seed=42)
    matcher.set_f32_mode(vsm.F32_MFMA_F32)
    try:
        _check(matcher, images, vocab, 2)
    finally:
        matcher.set_f32_mode(vsm.F32_FAST_EXACT)


@pytest.mark.parametrize("sizes,stride", [([300, 0, 200, 513], 7), ([40], 1), ([5, 3, 9], 100), ([700] * 4, 64)])
def test_sampled_rows_are_every_nth_row_of_the_concatenation(matcher, sizes, stride):
    rng = np.random.default_rng(stride + len(sizes))
    images = [synth.sift_like(n, rng) for n in sizes]
    ids = [matcher.upload_u8(img) for img in images]
    sample = matcher.sample_rows(ids, stride)
    expect = np.concatenate(images)[::stride]
    assert matcher.count(sample) == len(expect)
    # a vocabulary of as many words as rows, no iterations: the rows themselves
    np.testing.assert_array_equal(matcher.train_vocabulary([sample], len(expect), 0), expect)
    np.testing.assert_array_equal(matcher.train_vocabulary([sample], 16, 3), orc.train_vocabulary([expect], 16, 3))
    for s in ids + [sample]:
        matcher.release(s)


def test_training_a_vocabulary_of_more_than_8192_words(matcher):
    rng = np.random.default_rng(15)
    images = [synth.sift_like(3000, rng) for _ in range(3)]
    images[1][:2000] = images[0][:2000]
    ids = [matcher.upload_u8(img) for img in images]
    words = matcher.train_vocabulary(ids, 8500, 2)
    for s in ids:
        matcher.release(s)
    np.testing.assert_array_equal(words, orc.train_vocabulary(images, 8500, 2))
