"""vsm_match_pair_noalloc (the entry point that allocates nothing) against vsm_match_pair on the same
resident sets and against the CPU oracle: same lists, bit for bit, on calls in a row that share the per-GPU
scratch array, a smaller pair after a larger one included; and the device's free memory does not move
across a hundred calls."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as orc
import visual_slam_amd as vsm
from visual_slam_amd import synth

pytestmark = pytest.mark.gpu

D2H = 2


class DevOut:
    def __init__(self, cap):
        self.hip = C.CDLL("libamdhip64.so")          # the runtime the library already loaded
        self.cap = cap
        self.ptrs = []
        for nbytes in (4 * cap, 4 * cap, 256):
            p = C.c_void_p()
            assert self.hip.hipMalloc(C.byref(p), C.c_size_t(nbytes)) == 0
            assert self.hip.hipMemset(p, 0xAB, C.c_size_t(nbytes)) == 0     # stale contents must not show
            self.ptrs.append(p)

    def read(self):
        cnt = np.zeros(1, np.int32)
        assert self.hip.hipMemcpy(cnt.ctypes.data_as(C.c_void_p), self.ptrs[2], C.c_size_t(4), D2H) == 0
        n = int(cnt[0])
        assert 0 <= n <= self.cap
        i0, i1 = np.empty(max(n, 1), np.int32), np.empty(max(n, 1), np.int32)
        if n:
            assert self.hip.hipMemcpy(i0.ctypes.data_as(C.c_void_p), self.ptrs[0], C.c_size_t(4 * n), D2H) == 0
            assert self.hip.hipMemcpy(i1.ctypes.data_as(C.c_void_p), self.ptrs[1], C.c_size_t(4 * n), D2H) == 0
        return i0[:n], i1[:n]

    def free(self):
        for p in self.ptrs:
            self.hip.hipFree(p)


def _run(matcher, out, s0, s1, n0, **kw):
    matcher.match_pair_noalloc(s0, s1, out.ptrs[0].value, out.ptrs[1].value, out.ptrs[2].value, n0, **kw)
    matcher.synchronize()
    return out.read()


def test_noalloc_equals_match_pair_on_calls_in_a_row(matcher):
    shapes = [(8192, 8192), (700, 300), (16384, 5000), (1, 1), (513, 16384), (31, 2)]
    out = DevOut(16384)
    try:
        for k, (n0, n1) in enumerate(shapes):        # the second call is smaller than the first
            a, b = synth.sift_pair(n0, n1, seed=300 + k)
            s0, s1 = matcher.upload_u8(a), matcher.upload_u8(b)
            try:
                for kw in (dict(ratio=0.8, mutual=True), dict(ratio=0.8, mutual=False), dict(ratio=None, mutual=True)):
                    w0, w1 = matcher.match_pair(s0, s1, **kw)
                    g0, g1 = _run(matcher, out, s0, s1, n0, **kw)
                    np.testing.assert_array_equal(g0, w0)
                    np.testing.assert_array_equal(g1, w1)
                    o0, o1, _ = orc.match_u8(a, b, orc.params((4, 5) if kw["ratio"] else None, kw["mutual"]))
                    np.testing.assert_array_equal(g0, o0)
                    np.testing.assert_array_equal(g1, o1)
                if n0 >= 700:
                    assert len(w0) > 0
            finally:
                matcher.release(s0)
                matcher.release(s1)
    finally:
        out.free()


def _free_bytes(hip):
    free, total = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def test_noalloc_leaves_the_free_memory_where_it_was(matcher):
    """A hundred calls, shapes and rules changing from call to call (every row of the duplicate sets takes the rescan
    pass): the free device memory reads the same before and after, and the last lists equal the oracle's."""
    rng = np.random.default_rng(5)
    out = DevOut(4096)
    a, b = synth.sift_pair(4096, 3000, seed=77)
    dup = np.ascontiguousarray(synth.sift_like(30, rng)[rng.integers(0, 30, 2500)])
    ids = [matcher.upload_u8(x) for x in (a, b, dup)]
    sizes = {ids[0]: 4096, ids[1]: 3000, ids[2]: 2500}
    try:
        _run(matcher, out, ids[0], ids[1], 4096)          # first call on this process's context: kernels get loaded
        before = _free_bytes(out.hip)
        for k in range(100):
            s0, s1 = [(ids[0], ids[1]), (ids[1], ids[0]), (ids[2], ids[2]), (ids[2], ids[0])][k % 4]
            matcher.match_pair_noalloc(s0, s1, out.ptrs[0].value, out.ptrs[1].value, out.ptrs[2].value, sizes[s0],
                                       ratio=(0.8 if k % 3 else None), mutual=bool(k % 2))
        matcher.synchronize()
        assert _free_bytes(out.hip) == before
        g0, g1 = _run(matcher, out, ids[2], ids[0], 2500)
        o0, o1, _ = orc.match_u8(dup, a, orc.params((4, 5), True))
        np.testing.assert_array_equal(g0, o0)
        np.testing.assert_array_equal(g1, o1)
        g0, g1 = _run(matcher, out, ids[2], ids[2], 2500, ratio=None, mutual=False)
        o0, o1, _ = orc.match_u8(dup, dup, orc.params(None, False))
        np.testing.assert_array_equal(g0, o0)
        np.testing.assert_array_equal(g1, o1)
    finally:
        for sid in ids:
            matcher.release(sid)
        out.free()


def test_noalloc_empty_set_and_limits(matcher):
    out = DevOut(64)
    a = synth.sift_pair(40, 8, seed=9)[0]
    s0, se = matcher.upload_u8(a), matcher.upload_u8(np.zeros((0, 128), np.uint8))
    big = matcher.upload_u8(np.zeros((16385, 128), np.uint8))
    f0 = matcher.upload_f32(np.ones((4, 8), np.float32))
    try:
        g0, g1 = _run(matcher, out, s0, se, 40)
        assert len(g0) == 0
        g0, g1 = _run(matcher, out, se, s0, 0)
        assert len(g0) == 0
        with pytest.raises(vsm.VsmError) as e:       # beyond the fixed scratch: refused, with the limit in the message
            matcher.match_pair_noalloc(s0, big, out.ptrs[0].value, out.ptrs[1].value, out.ptrs[2].value, 64)
        assert e.value.code == -4 and "16384" in str(e.value)
        with pytest.raises(vsm.VsmError) as e:       # output buffers too small for one entry per image-0 row
            matcher.match_pair_noalloc(s0, s0, out.ptrs[0].value, out.ptrs[1].value, out.ptrs[2].value, 39)
        assert e.value.code == -4
        with pytest.raises(vsm.VsmError):
            matcher.match_pair_noalloc(f0, f0, out.ptrs[0].value, out.ptrs[1].value, out.ptrs[2].value, 64)
    finally:
        for sid in (s0, se, big, f0):
            matcher.release(sid)
        out.free()
