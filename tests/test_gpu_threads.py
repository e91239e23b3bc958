"""Two contexts on one GPU, each driven by its own thread (a context is single-threaded, distinct contexts
may run side by side): both must produce the oracle's lists."""
import threading

import numpy as np
import pytest

import oracle_lib as orc
import visual_slam_amd as vsm
from visual_slam_amd import synth

pytestmark = pytest.mark.gpu


def test_two_contexts_in_two_threads():
    data = []
    for t in range(2):
        a, b = synth.sift_pair(1500 + 300 * t, 1800, seed=70 + t)
        fa, fb = synth.superpoint_pair(700, 900 + 100 * t, 256, seed=80 + t)
        data.append((a, b, fa, fb))
    results, errors = [None, None], []

    def work(t):
        try:
            m = vsm.Matcher(0)
            a, b, fa, fb = data[t]
            out = []
            for _ in range(6):
                sa, sb = m.upload_u8(a), m.upload_u8(b)
                i0, i1 = m.match_pair(sa, sb, ratio=0.8, mutual=True)
                sfa, sfb = m.upload_f32(fa), m.upload_f32(fb)
                m0, m1 = m.match_pair_dense(sfa, sfb, ratio=None, mutual=True, metric=vsm.METRIC_DOT)
                off, p0, p1 = m.match_pairs([[sa, sb], [sb, sa]], ratio=0.8, mutual=True)
                out.append((i0.copy(), i1.copy(), m0.copy(), off.copy(), p0.copy(), p1.copy()))
                for s in (sa, sb, sfa, sfb):
                    m.release(s)
            m.close()
            results[t] = out
        except Exception as e:   # noqa: BLE001 - reported below, in the main thread
            errors.append(e)

    threads = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(120)
    assert not errors, errors
    for t in range(2):
        a, b, fa, fb = data[t]
        e0, _ = orc.match_dense_u8(a, b, orc.params((4, 5), True))
        r0, _ = orc.match_dense_u8(b, a, orc.params((4, 5), True))
        f0, _, _, _ = orc.match_dense_f32(fa, fb, orc.METRIC_DOT, orc.params(None, True))
        ei = np.nonzero(e0 >= 0)[0]
        ri = np.nonzero(r0 >= 0)[0]
        for i0, i1, m0, off, p0, p1 in results[t]:
            np.testing.assert_array_equal(i0, ei)
            np.testing.assert_array_equal(i1, e0[ei])
            np.testing.assert_array_equal(m0, f0)
            assert off.tolist() == [0, len(ei), len(ei) + len(ri)]
            np.testing.assert_array_equal(p0, np.concatenate([ei, ri]))
            np.testing.assert_array_equal(p1, np.concatenate([e0[ei], r0[ri]]))
