"""The sliced map-database query (vsm_dbq_*: slices over devices, RCCL all-gather called from C++, device merge)
against the search over the whole database on one context and against the CPU oracle.  On the one-GPU box the
all-gather has one rank; the slicing, the gathered layout and the merge are what several devices run."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as orc
import visual_slam_amd as vsm
from visual_slam_amd import synth

pytestmark = pytest.mark.gpu
CLI = os.path.join(os.path.dirname(vsm.lib_path()), "vsm_query_db")


def _data(nq, ndb, dim, seed):
    rng = np.random.default_rng(seed)
    db = synth.unit_f32(ndb, dim, rng)
    q = synth.unit_f32(nq, dim, rng)
    k = min(nq, ndb) // 3
    noisy = db[rng.choice(ndb, k, replace=False)] + 0.05 * rng.standard_normal((k, dim)).astype(np.float32)
    q[rng.choice(nq, k, replace=False)] = noisy / np.linalg.norm(noisy, axis=1, keepdims=True)
    return np.ascontiguousarray(q, np.float32), np.ascontiguousarray(db, np.float32)


@pytest.mark.parametrize("metric", [vsm.METRIC_L2, vsm.METRIC_DOT])
@pytest.mark.parametrize("nq,ndb,dim,nslices", [(300, 5000, 256, 4), (1000, 20011, 128, 7), (64, 3, 256, 2), (17, 1, 64, 3)])
def test_sliced_query_equals_the_whole_database_search(matcher, metric, nq, ndb, dim, nslices):
    q, db = _data(nq, ndb, dim, seed=nq + ndb)
    with vsm.SlicedDatabase([0], nslices) as sd:
        sd.load(db)
        base = sd.slices()
        assert base[0] == 0 and base[-1] == ndb and len(base) == nslices + 1 and np.all(np.diff(base) >= 0)
        k1, k2, i1, i2 = sd.query(q, metric)
        k1b, _, i1b, _ = sd.query(q[: nq // 2 + 1], metric)          # a second, smaller query batch on the same handle
    sq, sdb = matcher.upload_f32(q), matcher.upload_f32(db)
    try:
        w1, w2, wk1, wk2 = matcher.knn2(sq, sdb, metric, want_idx2=True)
    finally:
        matcher.release(sq)
        matcher.release(sdb)
    np.testing.assert_array_equal(i1, w1.astype(np.int64))
    np.testing.assert_array_equal(k1.view(np.uint32), wk1.view(np.uint32))
    if ndb >= 2:
        np.testing.assert_array_equal(i2, w2.astype(np.int64))
        np.testing.assert_array_equal(k2.view(np.uint32), wk2.view(np.uint32))
    else:
        assert np.all(i2 == -1)
    np.testing.assert_array_equal(i1b, i1[: nq // 2 + 1])
    np.testing.assert_array_equal(k1b.view(np.uint32), k1[: nq // 2 + 1].view(np.uint32))
    oi, ok1, ok2 = orc.knn2_f32(q, db, metric)
    np.testing.assert_array_equal(i1, oi.astype(np.int64))
    np.testing.assert_array_equal(k1.view(np.uint32), ok1.view(np.uint32))


def test_dim_major_database_and_bad_arguments():
    q, db = _data(100, 3000, 256, seed=5)
    with vsm.SlicedDatabase([0], 3) as sd:
        sd.load(np.ascontiguousarray(db.T), layout=vsm.LAYOUT_DIM_MAJOR)
        k1, k2, i1, i2 = sd.query(np.ascontiguousarray(q.T), vsm.METRIC_L2, layout=vsm.LAYOUT_DIM_MAJOR)
    oi, ok1, _ = orc.knn2_f32(q, db, orc.METRIC_L2)
    np.testing.assert_array_equal(i1, oi.astype(np.int64))
    np.testing.assert_array_equal(k1.view(np.uint32), ok1.view(np.uint32))
    with pytest.raises(vsm.VsmError):
        vsm.SlicedDatabase([0, 0], 2)          # a collective takes one rank per device
    with pytest.raises(vsm.VsmError):
        vsm.SlicedDatabase([0], -2)
    with vsm.SlicedDatabase([0], 1) as sd:
        with pytest.raises(vsm.VsmError):
            sd.query(q)                        # nothing loaded


def test_query_db_cli(tmp_path):
    q, db = _data(257, 9000, 256, seed=9)
    np.save(str(tmp_path / "db.npy"), db)
    np.save(str(tmp_path / "q.npy"), q)
    out = str(tmp_path / "nn.txt")
    res = subprocess.run([CLI, "--db", str(tmp_path / "db.npy"), "--queries", str(tmp_path / "q.npy"), "--out", out,
                          "--devices", "0", "--slices", "5"], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    oi, ok1, ok2 = orc.knn2_f32(q, db, orc.METRIC_L2)
    rows = [line.split() for line in open(out).read().splitlines()]
    assert len(rows) == len(q)
    np.testing.assert_array_equal(np.array([int(r[0]) for r in rows]), oi)
    np.testing.assert_array_equal(np.array([np.float32(r[1]) for r in rows]).view(np.uint32), ok1.view(np.uint32))
    np.testing.assert_array_equal(np.array([np.float32(r[3]) for r in rows]).view(np.uint32), ok2.view(np.uint32))
    assert "5 slices on 1 device" in res.stderr
    res = subprocess.run([CLI, "--synthetic", "512,20000,256", "--slices", "2", "--reps", "2"], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0 and "G evals/s" in res.stderr, res.stderr
