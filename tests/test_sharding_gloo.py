"""Multi-process paths on CPU (gloo, world_size 2): the pair-list sharding bench.py uses, and the
sliced-database 2-NN merge (per-rank local 2-NN -> all_gather -> host merge).  The per-rank local
search is done by the CPU oracle here (no GPU in this tier); the exchange and the merge are the
product code paths."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import oracle_lib as orc
    import visual_slam_amd as vsm
    from visual_slam_amd import _binding as B
    from visual_slam_amd import synth

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    # ---- pair-list sharding: contiguous shards, every pair exactly once -------------------
    # (the cut every rank of bench.py makes: bench.shard_bounds; 23 images give 253 pairs, an odd count)
    import bench
    pairs = vsm.generate_exhaustive_pairs(23)
    lo, hi = bench.shard_bounds(len(pairs), world, rank)
    shard = pairs[lo:hi]
    bounds = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(bounds, torch.tensor([lo, hi], dtype=torch.int64))
    bounds = [b.tolist() for b in bounds]
    assert bounds[0][0] == 0 and bounds[-1][1] == len(pairs)            # every pair ...
    assert all(bounds[r][1] == bounds[r + 1][0] for r in range(world - 1))   # ... exactly once
    assert max(b[1] - b[0] for b in bounds) - min(b[1] - b[0] for b in bounds) <= 1
    firsts = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(firsts, torch.tensor(shard[0].tolist(), dtype=torch.int64))
    assert firsts[rank].tolist() == pairs[lo].tolist()

    # ---- sliced database: local 2-NN per rank, all_gather, merge -----------------------------
    rng = np.random.default_rng(42)          # same stream on every rank
    q = synth.unit_f32(64, 256, rng)
    db = synth.unit_f32(1000, 256, rng)
    db[900] = db[10]                         # duplicate across slices: lower global index must win
    per_db = (len(db) + world - 1) // world
    lo, hi = rank * per_db, min(len(db), (rank + 1) * per_db)
    base = np.asarray([min(len(db), r * per_db) for r in range(world + 1)], np.int64)
    for metric in (vsm.METRIC_L2, vsm.METRIC_DOT):
        # local top-2 with indices: nearest, then nearest with the winner removed
        i1, k1, k2 = orc.knn2_f32(q, db[lo:hi], metric)
        i2 = np.empty_like(i1)
        for r in range(len(q)):
            sl = db[lo:hi].copy()
            sl[i1[r]] = -q[r] * 10 if metric == vsm.METRIC_DOT else q[r] + 100.0
            i2[r] = orc.knn2_f32(q[r:r + 1], sl, metric)[0][0]
        pack = torch.from_numpy(np.stack([k1.view(np.int32), k2.view(np.int32), i1, i2]))
        allp = [torch.empty_like(pack) for _ in range(world)]
        dist.all_gather(allp, pack)
        allp = torch.stack(allp).numpy()
        ok1, ok2, oi1, oi2 = B.merge_top2_f32(allp[:, 0].view(np.float32), allp[:, 1].view(np.float32),
                                              allp[:, 2], allp[:, 3], base, metric)
        wi, wk1, wk2 = orc.knn2_f32(q, db, metric)
        np.testing.assert_array_equal(oi1, wi)
        np.testing.assert_array_equal(ok1, wk1)
        np.testing.assert_array_equal(ok2, wk2)
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(tmpdir, "ok%d" % rank), "w").write("ok")


def test_world_size_two(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")


def _dry(args):
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run"] + args, env=env,
                         stdout=subprocess.PIPE, check=True, timeout=300).stdout.decode()
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out           # exactly one JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_launcher_spawns_ranks_and_shards_every_pair():
    """`bench.py --gpus 2` with no launcher around it starts two ranks itself; their shards of the pair
    list are contiguous, differ by at most one pair and cover every pair (253 pairs: an odd count)."""
    r = _dry(["--gpus", "2", "--images", "23"])
    assert r["n_gpus"] == 2 and r["items"] == 253
    assert r["shards"] == [[0, 0, 126], [1, 126, 253]]
    r = _dry(["--gpus", "3", "--workload", "db_query"])
    assert r["n_gpus"] == 3
    assert [s[1:] for s in r["shards"]] == [[0, 333334], [333334, 666668], [666668, 1000000]]
    r = _dry(["--images", "23"])
    assert r["n_gpus"] == 1 and r["shards"] == [[0, 0, 253]]


def test_bench_launcher_fails_when_a_rank_fails():
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--gpus", "2", "--images", "1"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    # one image has no pairs: generate_exhaustive_pairs(1) is empty but valid, so this still succeeds
    assert p.returncode == 0
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--gpus", "2", "--images", "-5"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode != 0 and not p.stdout.strip()


def test_bench_finds_the_committed_counter_figures():
    """The bench line quotes HBM bytes per launch and the matrix-pipe share from the committed counter passes, looked
    up by kernel name: a renamed kernel (a new template argument) must not turn them into silent nulls."""
    sys.path.insert(0, ROOT)
    import bench

    sift = bench.measured_traffic("void vsm::sift_query_tile_kernel<true>")
    assert sift is not None and 1e9 < sift < 4.3e9          # below the algorithmic operand bytes of one launch
    coarse = bench.measured_traffic("void vsm::bf16_coarse_kernel<256, false, false>", "r03_traffic_db_query.json")
    assert coarse is not None and 5e8 < coarse < 3e9        # the bf16 database is 0.52 GB
    busy = bench.measured_mfma_busy("void vsm::sift_query_tile_kernel<true>")
    assert busy is not None and 0.5 < busy < 1.0
    for kernel, summary in (("void vsm::bf16_coarse_kernel<256, false, false>", "r03_dbquery_mfma_utilisation.txt"),
                            ("void vsm::bf16_coarse_kernel<256, true, false>", "r03_f32pairs_mfma_utilisation.txt")):
        busy = bench.measured_mfma_busy(kernel, summary)
        assert busy is not None and 0.5 < busy < 1.0, kernel
    assert bench.shard_bounds(10, 3, 0) == (0, 3) and bench.shard_bounds(10, 3, 2) == (6, 10)
