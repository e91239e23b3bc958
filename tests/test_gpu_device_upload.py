"""GPU parity of the device-pointer entry points (vsm_upload_u8_device, vsm_upload_f32_device in both
layouts): descriptors that already sit in device memory, as the benchmark and a PyTorch-side caller hand
them over.  The buffers come from hipMalloc through the runtime the library has loaded; results must
equal the host-upload path and the CPU oracle bit for bit."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as orc
import visual_slam_amd as vsm
from visual_slam_amd import synth

pytestmark = pytest.mark.gpu

H2D = 1


class DevMem:
    def __init__(self, arr, offset=0):
        self.hip = C.CDLL("libamdhip64.so")          # already loaded by libvsm.so
        self.base = C.c_void_p()
        arr = np.ascontiguousarray(arr)
        assert self.hip.hipMalloc(C.byref(self.base), C.c_size_t(arr.nbytes + offset + 256)) == 0
        self.ptr = self.base.value + offset
        if arr.nbytes:
            assert self.hip.hipMemcpy(C.c_void_p(self.ptr), arr.ctypes.data_as(C.c_void_p), C.c_size_t(arr.nbytes), H2D) == 0

    def free(self):
        self.hip.hipFree(self.base)


SIZES_U8 = [(1, 1), (31, 513), (513, 31), (8192 + 300, 700)]


@pytest.mark.parametrize("n0,n1", SIZES_U8)
def test_u8_device_upload_equals_host_upload_and_oracle(matcher, n0, n1):
    a, b = synth.sift_pair(n0, n1, seed=500 + n0 + n1)
    # a row offset into a larger allocation, as a slice of an arena has (128-byte steps stay 16-byte aligned)
    da, db = DevMem(a, offset=128 * 3), DevMem(b)
    ids = []
    try:
        d0, d1 = matcher.upload_u8_device(da.ptr, n0), matcher.upload_u8_device(db.ptr, n1)
        h0, h1 = matcher.upload_u8(a), matcher.upload_u8(b)
        ids = [d0, d1, h0, h1]
        assert matcher.count(d0) == n0 and matcher.count(d1) == n1
        oi, od1, od2 = orc.knn2_u8(a, b)
        for q, t in ((d0, d1), (d0, h1), (h0, d1)):
            gi, gd1, gd2 = matcher.knn2(q, t)
            np.testing.assert_array_equal(gi, oi)
            np.testing.assert_array_equal(gd1, od1)
            np.testing.assert_array_equal(gd2, od2)
        o0, o1, od = orc.match_u8(a, b, orc.params((4, 5), True))
        g0, g1, gd = matcher.match_pair(d0, d1, ratio=0.8, mutual=True, return_dist=True)
        np.testing.assert_array_equal(g0, o0)
        np.testing.assert_array_equal(g1, o1)
        np.testing.assert_array_equal(gd, od)
        # the submission path the benchmark times
        matcher.submit_pairs(np.asarray([[d0, d1], [d1, d0]], np.int32), ratio=0.8, mutual=True)
        off, i0, i1 = matcher.collect_matches()
        np.testing.assert_array_equal(i0[off[0]:off[1]], o0)
        np.testing.assert_array_equal(i1[off[0]:off[1]], o1)
        r0, r1, _ = orc.match_u8(b, a, orc.params((4, 5), True))
        np.testing.assert_array_equal(i0[off[1]:off[2]], r0)
        np.testing.assert_array_equal(i1[off[1]:off[2]], r1)
    finally:
        for sid in ids:
            matcher.release(sid)
        da.free()
        db.free()


def test_u8_device_upload_rejects_unaligned_pointer(matcher):
    a = synth.sift_pair(8, 8, seed=1)[0]
    d = DevMem(a, offset=4)
    try:
        with pytest.raises(vsm.VsmError):
            matcher.upload_u8_device(d.ptr, 8)
    finally:
        d.free()


def test_u8_device_upload_empty(matcher):
    sid = matcher.upload_u8_device(0, 0)
    assert matcher.count(sid) == 0
    matcher.release(sid)


@pytest.mark.parametrize("n0,n1,dim", [(1, 1, 256), (31, 513, 256), (513, 31, 100), (8192 + 300, 700, 256), (257, 90, 64)])
@pytest.mark.parametrize("layout", [vsm.LAYOUT_ROWS, vsm.LAYOUT_DIM_MAJOR], ids=["rows", "dim_major"])
@pytest.mark.parametrize("mode", [vsm.F32_FAST_EXACT, vsm.F32_MFMA_F32], ids=["bf16+rerank", "f32mfma"])
def test_f32_device_upload_equals_host_upload_and_oracle(matcher, n0, n1, dim, layout, mode):
    a, b = synth.superpoint_pair(n0, n1, dim, seed=900 + n0 + n1 + dim)
    src_a = a if layout == vsm.LAYOUT_ROWS else np.ascontiguousarray(a.T)
    src_b = b if layout == vsm.LAYOUT_ROWS else np.ascontiguousarray(b.T)
    da, db = DevMem(src_a, offset=4 * 5), DevMem(src_b)     # 4-byte alignment is all the float path asks
    ids = []
    matcher.set_f32_mode(mode)
    try:
        d0 = matcher.upload_f32_device(da.ptr, n0, dim, layout)
        d1 = matcher.upload_f32_device(db.ptr, n1, dim, layout)
        h1 = matcher.upload_f32(b)
        ids = [d0, d1, h1]
        for metric in (vsm.METRIC_DOT, vsm.METRIC_L2):
            oi, ok1, ok2 = orc.knn2_f32(a, b, metric)
            for t in (d1, h1):
                gi, gk1, gk2 = matcher.knn2(d0, t, metric)
                np.testing.assert_array_equal(gi, oi)
                np.testing.assert_array_equal(gk1, ok1)
                np.testing.assert_array_equal(gk2, ok2)
        o0, o1, _, _ = orc.match_dense_f32(a, b, orc.METRIC_DOT, orc.params(None, True, min_score=0.5))
        g0, g1 = matcher.match_pair_dense(d0, d1, ratio=None, mutual=True, metric=vsm.METRIC_DOT, min_score=0.5)
        np.testing.assert_array_equal(g0, o0)
        np.testing.assert_array_equal(g1, o1)
    finally:
        matcher.set_f32_mode(vsm.F32_FAST_EXACT)
        for sid in ids:
            matcher.release(sid)
        da.free()
        db.free()
