#!/usr/bin/env python3
"""Throughput of the descriptor-matching hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

Default workload = exhaustive SIFT matching (1000 images x 8192 SIFT-128 u8 features, all
i<j image pairs; ratio 0.8 + mutual check), the configuration the pairs/s and evals/s
metric is quoted on.  The descriptor arena is generated on the device and is resident in
HBM before the timed region.  One step = one submission of --pairs-per-step image pairs of
this rank's shard of the pair list through the whole path (distance kernel, top-2/ratio/
mutual, compaction, match lists copied back to the host).  Ranks shard the pair list; there
is no data-path collective.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

INT8_DENSE_PEAK_TOPS = 5000.0   # MI355X dense int8 MFMA, 2x the ~2.5 PFLOP/s bf16 figure
F32_MFMA_PEAK_TFLOPS = 157.3    # f32-input MFMA (exact f32), = vector peak
BF16_DENSE_PEAK_TFLOPS = 2500.0 # dense bf16 MFMA
HBM_PEAK_GBPS = 8000.0          # HBM3E


def measured_traffic(kernel, summary="r03_traffic_sift_exhaustive.json"):
    """HBM bytes per launch of `kernel` from the committed hardware-counter passes (rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE over tools/traffic_probe.py or tools/db_query_probe.py at this workload's
    shape, folded by tools/pmc_summary.py); None when no such pass is in the tree."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", summary)
    try:
        with open(path) as f:
            return float(json.load(f)["kernels"][kernel]["hbm_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None


def measured_mfma_busy(kernel, summary="r03_mfma_utilisation.txt"):
    """Fraction of the elapsed SIMD cycles in which the matrix pipes were busy, from a committed counter
    pass (profiles/<summary>, written by tools/refresh_profiles.sh or tools/bf16_counters.sh: the kernel's name at the
    start of a line, the fraction after "SIMDs) = " on that line or the next); None when absent."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", summary)
    try:
        lines = open(path).read().splitlines()
        for k, line in enumerate(lines):
            if line.startswith(kernel + ":") or line.startswith(kernel + " launches"):
                for cand in lines[k:k + 2]:
                    if "SIMDs) = " in cand:
                        return float(cand.split("SIMDs) = ")[1].split(";")[0])
    except (OSError, ValueError, IndexError):
        pass
    return None


def f32_roofline(m, flops, kms, busy_of=None):
    """Roofline object of a float-descriptor search: priced against the matrix-core peak of the
    path that actually ran (bf16 coarse pass + exact re-rank, or the f32-input kernel).  busy_of = (kernel name,
    summary file under profiles/) of the committed counter pass over this workload's shape, if there is one."""
    fast = m.last_f32_path() == 0
    peak = BF16_DENSE_PEAK_TFLOPS if fast else F32_MFMA_PEAK_TFLOPS
    achieved = flops / (kms * 1e-3) / 1e12
    busy = measured_mfma_busy(*busy_of) if (fast and busy_of) else None
    return {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
            "traffic": None, "kernel": "bf16_coarse_kernel" if fast else "f32_knn_kernel", "kernel_ms": kms,
            "mfma_busy_frac": busy,
            "mfma_busy_note": ("matrix-pipe busy cycles / (elapsed cycles x SIMDs) of the coarse kernel: a constant read from a "
                               "committed counter pass over the same shape (profiles/%s), not a measurement of this run"
                               % busy_of[1]) if busy is not None else None,
            "note": ("bf16 MFMA coarse pass + exact f32 re-rank of the survivors (bit-identical results); "
                     "kernel_ms brackets coarse + re-rank of every direction searched") if fast else
                    "f32-input MFMA path; kernel_ms includes the merge kernels"}

OPS_PER_EVAL_U8 = 256           # 128 multiply-accumulates
OPS_PER_EVAL_F32 = 512          # 256 fused multiply-adds


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="sift_exhaustive",
                    choices=["sift_exhaustive", "superpoint_pair", "superpoint_pairs", "sift_large_pair", "db_query",
                             "geom_verify", "shortlist"])
    ap.add_argument("--images", type=int, default=1000)
    ap.add_argument("--features", type=int, default=8192)
    ap.add_argument("--pairs-per-step", type=int, default=2048)
    ap.add_argument("--data", default="planted", choices=["planted", "clustered", "duplicates", "ones"],
                    help="sift_exhaustive: what the synthetic descriptors look like (planted: random rows with 30 %% "
                         "noisy copies of a shared pool; clustered: every row within +-2 of one of 12 centres; duplicates: "
                         "rows drawn from a pool of 200, a third of them with one byte changed by one; ones: every byte 1, "
                         "the dummy descriptors of FeaturePoints.cpp:358-367)")
    ap.add_argument("--u8-mode", type=int, default=0, help="sift workloads: vsm_set_u8_mode (0 auto, 1 packed form only, "
                                                           "2 / 3 tile-minimum form with separate passes / fused tail)")
    ap.add_argument("--f32-mode", type=int, default=0,
                    help="float workloads: 0 = bf16 coarse pass + exact re-rank, 1 = f32-input MFMA kernel only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-verify", action="store_true",
                    help="skip the check of a sample of the timed run's output against the CPU oracle")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: start the ranks, build the pair list / database slices, exchange the shard "
                         "bookkeeping over gloo and print it (what the CPU test of the launcher runs)")
    return ap.parse_args()


def shard_bounds(n, world, rank):
    """Contiguous shard [lo, hi) of n items for `rank` of `world`; the shards differ by at most one item
    and cover every item exactly once."""
    return (n * rank) // world, (n * (rank + 1)) // world


def host_threads():
    """Threads for the CPU baseline: every CPU this process may run on, unless OMP_NUM_THREADS (the
    box's CPU share) says fewer.  Returns (threads, cpus available to the process)."""
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    env = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    return (max(1, min(env, avail)) if env > 0 else avail), avail


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher around it: this process touches no GPU (it does not
    even import torch); it starts N fresh children, one rank per GPU, with the rendezvous variables
    torch.distributed.run would set, relays rank 0's JSON line and fails if any rank fails."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    rc = 0
    pending = set(range(args.gpus))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                sys.stderr.write("bench.py: rank %d exited with %d; stopping the other ranks\n" % (r, code))
                for o in pending:           # the others would wait in a barrier for ever
                    procs[o].terminate()
        if pending:
            time.sleep(0.05)
    out = procs[0].stdout.read().decode()
    if rc == 0:
        sys.stdout.write(out)
        sys.stdout.flush()
    return rc


def dist_setup(args):
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.dry_run:
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", rank=rank, world_size=world)
        return torch, dist, world, rank, local
    # Rehearsal of the multi-rank path on a one-GPU box: VSM_BENCH_REHEARSAL=1 puts every rank on
    # device 0 and runs the collectives over gloo on host tensors (RCCL refuses two ranks per GPU).
    global REHEARSAL
    REHEARSAL = os.environ.get("VSM_BENCH_REHEARSAL", "") == "1"
    if REHEARSAL:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if REHEARSAL:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    return torch, dist, world, rank, local


REHEARSAL = False


def comm_device():
    return "cpu" if REHEARSAL else "cuda"


def barrier_sync(torch, dist, world):
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(torch, dist, world, seconds):
    if world == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=comm_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(torch, dist, world, v):
    if world == 1:
        return v
    t = torch.tensor([float(v)], dtype=torch.float64, device=comm_device())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gen_sift_arena(torch, nimg, nfeat, seed, data="planted"):
    """SIFT-like u8 descriptors on the device: |N(0,1)| components, L2-normalised to 512, rounded
    (the convention of FeaturePoints.cpp:680-684).  30 % of every image's rows are noisy copies of
    rows of a shared 'scene' pool, so image pairs have real correspondences.
    Other values of `data` build the inputs the matcher likes least: rows in tight clusters, rows drawn from a small
    pool (exact duplicates and rows one unit apart everywhere), or the all-ones dummy descriptors the reference's own
    converter writes (FeaturePoints.cpp:358-367), where every distance is tied."""
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)

    def raw(n):
        x = torch.randn((n, 128), generator=g, device="cuda").abs_()
        x *= 512.0 / x.norm(dim=1, keepdim=True).clamp_min(1e-6)
        return x

    arena = torch.empty((nimg, nfeat, 128), dtype=torch.uint8, device="cuda")
    if data == "ones":
        arena.fill_(1)
        torch.cuda.synchronize()
        return arena
    if data in ("clustered", "duplicates"):
        centres = raw(12 if data == "clustered" else 200).round_().clamp_(0, 255)
        for i in range(nimg):
            x = centres[torch.randint(0, len(centres), (nfeat,), generator=g, device="cuda")].clone()
            if data == "clustered":
                x += torch.randint(-2, 3, (nfeat, 128), generator=g, device="cuda").float()
            else:
                rows = torch.randperm(nfeat, generator=g, device="cuda")[: nfeat // 3]
                cols = torch.randint(0, 128, (len(rows),), generator=g, device="cuda")
                x[rows, cols] = (x[rows, cols].to(torch.int32) ^ 1).float()
            arena[i] = x.clamp_(0, 255).to(torch.uint8)
        torch.cuda.synchronize()
        return arena
    pool = raw(nfeat)
    k = int(0.3 * nfeat)
    for i in range(nimg):
        x = raw(nfeat)
        src = torch.randperm(nfeat, generator=g, device="cuda")[:k]
        dst = torch.randperm(nfeat, generator=g, device="cuda")[:k]
        x[dst] = pool[src] + 6.0 * torch.randn((k, 128), generator=g, device="cuda")
        arena[i] = x.round_().clamp_(0, 255).to(torch.uint8)
    torch.cuda.synchronize()
    return arena


def cpu_baseline_u8(sets, budget_s):
    """The scalar CPU restatement (oracle) on the host cores: one-directional 2-NN over pairs of
    the given descriptor sets until the time budget is used, so its evals are counted exactly
    like the GPU's."""
    import oracle_lib as orc

    threads, avail = host_threads()
    orc.knn2_u8(sets[0][:256], sets[1], threads)  # page in, spin up the thread pool
    evals = 0
    npairs = 0
    t0 = time.perf_counter()
    while True:
        a = sets[npairs % len(sets)]
        b = sets[(npairs + 1) % len(sets)]
        orc.knn2_u8(a, b, threads)
        evals += len(a) * len(b)
        npairs += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or npairs >= 4096:
            break
    return {
        "value": evals / dt / 1e9, "unit": "G evals/s", "cores": threads, "host_cpus": avail, "kind": "port",
        "sample": "oracle 2-NN, %d pairs of %d x %d SIFT-128 u8 from the same arena, %.1f s"
                  % (npairs, len(sets[0]), len(sets[1]), dt),
    }


def run_sift_exhaustive(args, torch, dist, world, rank):
    import visual_slam_amd as vsm

    nimg, nfeat, pps = args.images, args.features, args.pairs_per_step
    arena = gen_sift_arena(torch, nimg, nfeat, seed=1234, data=args.data)
    m = vsm.Matcher(torch.cuda.current_device())
    m.set_u8_mode(args.u8_mode)
    ids = [m.upload_u8_device(arena[i].data_ptr(), nfeat) for i in range(nimg)]
    pairs_all = vsm.generate_exhaustive_pairs(nimg)
    # contiguous shard per rank: consecutive pairs share image 0, so it stays cache-resident
    lo, hi = shard_bounds(len(pairs_all), world, rank)
    shard = pairs_all[lo:hi]
    id_arr = np.asarray(ids, np.int32)

    # A step = one GPU submission of `pps` pairs through the whole path, match lists copied to the
    # host.  Two submissions are kept in flight (submission k + 1 is queued before the lists of
    # submission k are collected), all inside the timed region.
    def pairs_of(k):
        lo = (k * pps) % max(len(shard) - pps + 1, 1)
        return id_arr[shard[lo:lo + pps]]

    cap = pps * nfeat
    bufs = [(np.empty(pps + 1, np.int64), np.empty(cap, np.int32), np.empty(cap, np.int32)) for _ in range(2)]

    last = {}

    def run(first, count):
        npairs = nmatch = 0
        kms, vms = [], []
        if count > 0:
            m.submit_pairs(pairs_of(first), ratio=0.8, mutual=True)
        for k in range(count):
            if k + 1 < count:
                m.submit_pairs(pairs_of(first + k + 1), ratio=0.8, mutual=True)
            off, i0, i1 = m.collect_matches(bufs[k & 1])
            npairs += len(off) - 1
            nmatch += int(off[-1])
            kms.append(m.last_kernel_ms())   # brackets of the submission just collected
            vms.append(m.last_verify_ms())
            last.update(step=first + k, off=off, i0=i0, i1=i1)
        return npairs, nmatch, kms, vms

    m.set_timing(True)
    run(0, args.warmup)
    barrier_sync(torch, dist, world)
    t0 = time.perf_counter()
    npairs, nmatch, kernel_ms, verify_ms = run(args.warmup, args.steps)
    barrier_sync(torch, dist, world)
    dt = max_over_ranks(torch, dist, world, time.perf_counter() - t0)
    total_pairs = sum_over_ranks(torch, dist, world, npairs)
    verified = None
    if not args.no_verify and last:
        # outside the timed region: three pairs of the last timed submission, exactly the lists the
        # timed run copied to the host, against the CPU oracle on the same descriptors
        import oracle_lib as orc
        threads, _ = host_threads()
        lo_p = (last["step"] * pps) % max(len(shard) - pps + 1, 1)
        step_pairs = shard[lo_p:lo_p + pps]
        ok = True
        for k in sorted({0, len(step_pairs) // 2, len(step_pairs) - 1}):
            ia, ib = int(step_pairs[k][0]), int(step_pairs[k][1])
            o0, o1, _ = orc.match_u8(arena[ia].cpu().numpy(), arena[ib].cpu().numpy(), orc.params((4, 5), True), threads)
            a, b = int(last["off"][k]), int(last["off"][k + 1])
            ok = ok and np.array_equal(last["i0"][a:b], o0) and np.array_equal(last["i1"][a:b], o1)
        verified = bool(sum_over_ranks(torch, dist, world, 0 if ok else 1) == 0)
    evals = total_pairs * float(nfeat) * float(nfeat)
    kms = float(np.mean(kernel_ms))
    launch_pairs = m.last_submission_pairs()
    evals_per_launch = float(launch_pairs) * nfeat * nfeat
    achieved = OPS_PER_EVAL_U8 * evals_per_launch / (kms * 1e-3) / 1e12
    out = {
        "metric": "descriptor distance evals/sec", "value": evals / dt / 1e9, "unit": "G/s",
        "pairs_per_s": total_pairs / dt,
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic", "verified": verified,
        "config": {"workload": "exhaustive SIFT-128 matching, %d images x %d features, ratio 0.8 + mutual, "
                               "%d pairs per step per GPU, two steps in flight, match lists copied to host"
                               % (nimg, nfeat, pps),
                   "data": args.data, "u8_mode": args.u8_mode,
                   "pairs_total": int(len(pairs_all)), "parallelism": "pair list sharded x%d" % world,
                   "matches_per_pair": nmatch / max(npairs, 1)},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": INT8_DENSE_PEAK_TOPS, "unit": "TOP/s",
                     "frac": achieved / INT8_DENSE_PEAK_TOPS,
                     "frac_step": OPS_PER_EVAL_U8 * float(pps) * nfeat * nfeat / (dt / args.steps) / 1e12
                                  / INT8_DENSE_PEAK_TOPS,
                     "traffic": measured_traffic("void vsm::sift_query_tile_kernel<true>")
                     if (nimg, nfeat, launch_pairs) == (1000, 8192, 2048) else None,
                     "mfma_busy_frac": measured_mfma_busy("void vsm::sift_query_tile_kernel<true>")
                     if (nimg, nfeat, launch_pairs) == (1000, 8192, 2048) else None,
                     "mfma_busy_note": "matrix-pipe busy cycles / (elapsed cycles x SIMDs): a constant read from a committed "
                                       "counter pass over the same submission shape (profiles/r03_mfma_utilisation.txt), "
                                       "not a measurement of this run",
                     "traffic_note": "HBM bytes per launch: a constant read from committed rocprofv3 --pmc passes over the same "
                                     "submission shape (profiles/r03_probe1000_pmc_*.csv; FETCH_SIZE doubled as gfx950 needs), "
                                     "not a measurement of this run; algorithmic operand bytes per launch: %d"
                                     % (launch_pairs * 2 * nfeat * 128),
                     "kernel": "sift_query_tile_kernel" if m.last_u8_path() in (0, 2) else "sift_query_kernel", "kernel_ms": kms,
                     "verify_kernel_ms": float(np.mean(verify_ms)),
                     "note": "int8 ops: 256 per descriptor pair (128 MAC), one direction counted once"},
    }
    if out["roofline"]["traffic"]:
        # the committed counter figure over this run's kernel time, against the 8 TB/s of HBM3E
        out["roofline"]["hbm_gb_per_s"] = out["roofline"]["traffic"] / (kms * 1e-3) / 1e9
        out["roofline"]["hbm_frac_of_peak"] = out["roofline"]["hbm_gb_per_s"] / HBM_PEAK_GBPS
    if world == 1 and not args.no_cpu_baseline:
        sets = [arena[i].cpu().numpy() for i in range(min(8, nimg))]
        out["cpu_baseline"] = cpu_baseline_u8(sets, args.cpu_seconds)
    m.close()
    return out


def run_superpoint_pair(args, torch, dist, world, rank):
    """2 images x 4096 SuperPoint-256 float descriptors, mutual nearest neighbour (dot metric)."""
    import visual_slam_amd as vsm
    from visual_slam_amd import synth

    n = 4096
    a, b = synth.superpoint_pair(n, n, 256, seed=77 + rank)
    m = vsm.Matcher(torch.cuda.current_device())
    s0, s1 = m.upload_f32(a), m.upload_f32(b)
    m.set_timing(True)
    m.set_f32_mode(args.f32_mode)

    def step():
        return m.match_pair_dense(s0, s1, ratio=None, mutual=True, metric=vsm.METRIC_DOT)

    # A step is 70 us: with torch in the process the HIP runtime stalls once for some 35 ms a few hundred launches
    # after start-up (whatever is launched), which a short warm-up would leave inside the timed region.
    warm = max(args.warmup, 400)
    for _ in range(warm):
        step()
    barrier_sync(torch, dist, world)
    t0 = time.perf_counter()
    kernel_ms = []
    for _ in range(args.steps):
        m0, m1 = step()
        kernel_ms.append(m.last_kernel_ms())
    barrier_sync(torch, dist, world)
    dt = max_over_ranks(torch, dist, world, time.perf_counter() - t0)
    evals = float(world) * args.steps * n * n
    kms = float(np.mean(kernel_ms))
    verified = None
    if not args.no_verify:
        import oracle_lib as orc
        o0, o1, _, _ = orc.match_dense_f32(a, b, orc.METRIC_DOT, orc.params(None, True), host_threads()[0])
        ok = np.array_equal(m0, o0) and np.array_equal(m1, o1)
        verified = bool(sum_over_ranks(torch, dist, world, 0 if ok else 1) == 0)
    out = {
        "metric": "descriptor distance evals/sec", "value": evals / dt / 1e9, "unit": "G/s",
        "pairs_per_s": world * args.steps / dt,
        "n_gpus": world, "steps": args.steps, "warmup": warm,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic", "verified": verified,
        "config": {"workload": "SuperPoint-256 f32, 2 images x 4096 keypoints, mutual-NN, one pair per step",
                   "parallelism": "replicas x%d" % world, "matches": int((m0 >= 0).sum())},
        "roofline": f32_roofline(m, OPS_PER_EVAL_F32 * float(n) * n, kms),
    }
    if world == 1 and not args.no_cpu_baseline:
        import oracle_lib as orc
        threads, avail = host_threads()
        rows = 1024
        t0 = time.perf_counter()
        orc.knn2_f32(a[:rows], b, orc.METRIC_DOT, threads)
        c = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": rows * n / c / 1e9, "unit": "G evals/s", "cores": threads, "host_cpus": avail,
                               "kind": "port",
                               "sample": "oracle 2-NN, %d x %d float-256, %.1f s" % (rows, n, c)}
    m.close()
    return out


def run_superpoint_pairs(args, torch, dist, world, rank):
    """Exhaustive mutual-NN matching of 64 images x 4096 SuperPoint-256 float descriptors, 256 pairs per
    submission (the float counterpart of the SIFT pair-list workload)."""
    import visual_slam_amd as vsm

    nimg, n, pps = 64, 4096, 256
    g = torch.Generator(device="cuda")
    g.manual_seed(555)
    pool = torch.randn((n, 256), generator=g, device="cuda")
    ids = []
    keep = {}
    m = vsm.Matcher(torch.cuda.current_device())
    for i in range(nimg):
        x = torch.randn((n, 256), generator=g, device="cuda")
        k = n // 4
        dst = torch.randperm(n, generator=g, device="cuda")[:k]
        src = torch.randperm(n, generator=g, device="cuda")[:k]
        x[dst] = pool[src] + 0.05 * torch.randn((k, 256), generator=g, device="cuda")
        x /= x.norm(dim=1, keepdim=True)
        torch.cuda.synchronize()
        ids.append(m.upload_f32_device(x.data_ptr(), n, 256))
        keep[i] = x.cpu().numpy()
    m.set_timing(True)
    m.set_f32_mode(args.f32_mode)
    pairs_all = vsm.generate_exhaustive_pairs(nimg)
    lo_s, hi_s = shard_bounds(len(pairs_all), world, rank)
    mine = pairs_all[lo_s:hi_s]
    idmap = np.asarray(ids, np.int32)
    last = {}

    def step(k):
        lo = (k * pps) % max(len(mine) - pps + 1, 1)
        pr = idmap[np.asarray(mine[lo:lo + pps], np.int64)]
        tot = m.match_pairs_device(pr, ratio=None, mutual=True, metric=vsm.METRIC_DOT, min_score=0.5)
        last.update(pairs=mine[lo:lo + pps], lists=m.fetch_matches(len(pr), tot))
        return len(pr), tot

    for k in range(args.warmup):
        step(k)
    barrier_sync(torch, dist, world)
    t0 = time.perf_counter()
    kernel_ms, npairs, nmatch = [], 0, 0
    for k in range(args.steps):
        a, b = step(args.warmup + k)
        npairs += a
        nmatch += b
        kernel_ms.append(m.last_kernel_ms())
    barrier_sync(torch, dist, world)
    dt = max_over_ranks(torch, dist, world, time.perf_counter() - t0)
    total_pairs = sum_over_ranks(torch, dist, world, npairs)
    kms = float(np.mean(kernel_ms))
    verified = None
    if not args.no_verify and last:
        import oracle_lib as orc
        off, i0, i1 = last["lists"]
        ok = True
        for k in sorted({0, len(last["pairs"]) - 1}):
            ia, ib = int(last["pairs"][k][0]), int(last["pairs"][k][1])
            o0, _, _, _ = orc.match_dense_f32(keep[ia], keep[ib], orc.METRIC_DOT, orc.params(None, True, min_score=0.5),
                                              host_threads()[0])
            sel = np.nonzero(o0 >= 0)[0].astype(np.int32)
            lo, hi = int(off[k]), int(off[k + 1])
            ok = ok and np.array_equal(i0[lo:hi], sel) and np.array_equal(i1[lo:hi], o0[sel])
        verified = bool(sum_over_ranks(torch, dist, world, 0 if ok else 1) == 0)
    out = {
        "metric": "descriptor distance evals/sec", "value": total_pairs * float(n) * n / dt / 1e9, "unit": "G/s",
        "pairs_per_s": total_pairs / dt,
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic", "verified": verified,
        "config": {"workload": "exhaustive SuperPoint-256 f32 matching, %d images x %d keypoints, mutual-NN, "
                               "%d pairs per step per GPU, match lists copied to host" % (nimg, n, pps),
                   "parallelism": "pair list sharded x%d" % world, "matches_per_pair": nmatch / max(npairs, 1)},
        "roofline": f32_roofline(m, OPS_PER_EVAL_F32 * float(pps) * n * n, kms,
                                 ("void vsm::bf16_coarse_kernel<256, true, false>", "r03_f32pairs_mfma_utilisation.txt")),
    }
    m.close()
    return out


def run_sift_large_pair(args, torch, dist, world, rank):
    """One pair of 65536 x 65536 SIFT-128 descriptors per step."""
    import visual_slam_amd as vsm

    n = 65536
    arena = gen_sift_arena(torch, 2, n, seed=4321 + rank)
    m = vsm.Matcher(torch.cuda.current_device())
    ids = [m.upload_u8_device(arena[i].data_ptr(), n) for i in range(2)]
    m.set_timing(True)
    pr = np.asarray([[ids[0], ids[1]]], np.int32)

    last = {}

    def step():
        last["lists"] = m.match_pairs(pr, ratio=0.8, mutual=True)   # (offsets, idx0, idx1) in host arrays
        return len(last["lists"][1])

    for _ in range(args.warmup):
        step()
    barrier_sync(torch, dist, world)
    t0 = time.perf_counter()
    kernel_ms = []
    for _ in range(args.steps):
        tot = step()
        kernel_ms.append(m.last_kernel_ms())
    barrier_sync(torch, dist, world)
    dt = max_over_ranks(torch, dist, world, time.perf_counter() - t0)
    evals = float(world) * args.steps * n * n
    kms = float(np.mean(kernel_ms))
    achieved = OPS_PER_EVAL_U8 * float(n) * n / (kms * 1e-3) / 1e12
    verified = None
    if not args.no_verify and last:
        import oracle_lib as orc
        o0, o1, _ = orc.match_u8(arena[0].cpu().numpy(), arena[1].cpu().numpy(), orc.params((4, 5), True),
                                 host_threads()[0])
        ok = np.array_equal(last["lists"][1], o0) and np.array_equal(last["lists"][2], o1)
        verified = bool(sum_over_ranks(torch, dist, world, 0 if ok else 1) == 0)
    out = {
        "metric": "descriptor distance evals/sec", "value": evals / dt / 1e9, "unit": "G/s",
        "pairs_per_s": world * args.steps / dt,
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic", "verified": verified,
        "config": {"workload": "one SIFT-128 pair of 65536 x 65536 features, ratio 0.8 + mutual",
                   "parallelism": "replicas x%d" % world, "matches": int(tot)},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": INT8_DENSE_PEAK_TOPS, "unit": "TOP/s",
                     "frac": achieved / INT8_DENSE_PEAK_TOPS, "traffic": None,
                     "kernel": "sift_query_tile_kernel" if m.last_u8_path() in (0, 2) else "sift_query_kernel", "kernel_ms": kms},
    }
    if world == 1 and not args.no_cpu_baseline:
        sets = [arena[i, :8192].cpu().numpy() for i in range(2)]
        out["cpu_baseline"] = cpu_baseline_u8(sets, args.cpu_seconds)
    m.close()
    return out


def run_geom_verify(args, torch, dist, world, rank):
    """Geometric verification of the putative matches of 4096 image pairs per step (768 matches per pair,
    about what the exhaustive SIFT workload leaves; 256 RANSAC samples per pair).  The pair list shards
    like the matching does: no data-path collective."""
    import visual_slam_amd as vsm
    from visual_slam_amd import synth

    npairs, nmatch, hyp, nscene = 4096, 768, 256, 32
    m = vsm.Matcher(torch.cuda.current_device())
    scenes = []
    for k in range(nscene):
        xy0, xy1, i0, i1, _, _ = synth.two_view(nmatch, seed=1000 * rank + k, inlier_fraction=0.7, noise=0.5)
        scenes.append((m.upload_xy(xy0), m.upload_xy(xy1), i0, i1, xy0, xy1))
    xp = np.asarray([[scenes[p % nscene][0], scenes[p % nscene][1]] for p in range(npairs)], np.int32)
    off = np.arange(npairs + 1, dtype=np.int64) * nmatch
    i0 = np.concatenate([scenes[p % nscene][2] for p in range(npairs)])
    i1 = np.concatenate([scenes[p % nscene][3] for p in range(npairs)])
    m.set_timing(True)

    def step():
        return m.verify_matches(xp, off, i0, i1, max_sampson=2.0, hypotheses=hyp, seed=7)

    for _ in range(args.warmup):
        step()
    barrier_sync(torch, dist, world)
    t0 = time.perf_counter()
    kernel_ms = []
    for _ in range(args.steps):
        F, nf, in_off, g0, g1 = step()
        kernel_ms.append(m.last_kernel_ms())
    barrier_sync(torch, dist, world)
    dt = max_over_ranks(torch, dist, world, time.perf_counter() - t0)
    kms = float(np.mean(kernel_ms))
    verified = None
    if not args.no_verify:
        # the first pairs of the last step against the scalar restatement: F bit for bit, the same inlier lists
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests"))
        import oracle_lib as orc
        ok = True
        for p in range(0, npairs, max(1, npairs // 16)):
            sc = scenes[p % nscene]
            oF, o0, o1 = orc.geom_pair(sc[4], sc[5], sc[2], sc[3], max_sampson=2.0, hypotheses=hyp, seed=7, pair_pos=p)
            lo, hi = int(in_off[p]), int(in_off[p + 1])
            ok = ok and np.array_equal(np.asarray(F[p], np.float32).reshape(-1).view(np.uint32), oF.reshape(-1).view(np.uint32))
            ok = ok and np.array_equal(g0[lo:hi], o0) and np.array_equal(g1[lo:hi], o1)
        verified = bool(sum_over_ranks(torch, dist, world, 0 if ok else 1) == 0)
    tests = float(npairs) * hyp * nmatch           # Sampson tests of one step
    flops = 33.0 * tests                           # 18 multiplies + 15 adds per test (15 of the pairs fused)
    achieved = flops / (kms * 1e-3) / 1e12
    out = {
        "metric": "image pairs verified/sec", "value": world * args.steps * npairs / dt, "unit": "pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic", "verified": verified,
        "config": {"workload": "fundamental-matrix RANSAC, %d pairs x %d matches, %d samples per pair per step"
                               % (npairs, nmatch, hyp),
                   "parallelism": "pair list sharded x%d" % world, "inliers_per_pair": float(np.mean(nf))},
        "roofline": {"bound": "valu", "achieved": achieved, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / F32_MFMA_PEAK_TFLOPS, "traffic": None, "kernel": "geom_solve_kernel + geom_score_kernel + geom_emit_kernel",
                     "kernel_ms": kms,
                     "note": "f32 vector peak; kernel_ms brackets gather + RANSAC + final scoring; the eight-point "
                             "solves (f64, register-resident) are not counted as work"},
    }
    if world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests"))
        import oracle_lib as orc
        threads, avail = host_threads()
        orc.lib().oracle_geom_set_threads(threads)    # the oracle's loop over samples: the threads the SIFT baseline uses
        t1 = time.perf_counter()
        done = 0
        while time.perf_counter() - t1 < args.cpu_seconds:
            sc = scenes[done % nscene]
            orc.geom_pair(sc[4], sc[5], sc[2], sc[3], max_sampson=2.0, hypotheses=hyp, seed=7, pair_pos=done)
            done += 1
        el = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": done / el, "unit": "pairs/s", "cores": threads, "host_cpus": avail, "kind": "port",
                               "sample": "oracle RANSAC on %d pairs of the same shape, %.1f s" % (done, el)}
    m.close()
    return out


def run_shortlist(args, torch, dist, world, rank):
    """Pair shortlist for the exhaustive workload's collection: 1000 images x 8192 SIFT features, a
    vocabulary of 4096 words drawn from them, 20 partners per image.  One step = one whole shortlist
    (word assignment of 8.2 M descriptors, histograms, 1000 x 1000 scores, choices).  Does not shard:
    with more than one rank every rank computes the same list and the value counts all of them (replicas only)."""
    import visual_slam_amd as vsm

    nimg, nfeat, nwords, topk = args.images, args.features, 4096, 20
    arena = gen_sift_arena(torch, nimg, nfeat, seed=1234)
    g = torch.Generator(device="cuda")
    g.manual_seed(77)
    pick = torch.randperm(nimg * nfeat, generator=g, device="cuda")[:nwords]
    vocab = arena.reshape(-1, 128)[pick].contiguous()
    torch.cuda.synchronize()
    m = vsm.Matcher(torch.cuda.current_device())
    ids = [m.upload_u8_device(arena[i].data_ptr(), nfeat) for i in range(nimg)]
    vid = m.upload_u8_device(vocab.data_ptr(), nwords)
    m.set_u8_mode(args.u8_mode)
    m.set_timing(True)
    for _ in range(args.warmup):
        m.shortlist_pairs(ids, vid, topk)
    barrier_sync(torch, dist, world)
    t0 = time.perf_counter()
    kernel_ms = []
    for _ in range(args.steps):
        pairs = m.shortlist_pairs(ids, vid, topk)
        kernel_ms.append(m.last_kernel_ms())
    barrier_sync(torch, dist, world)
    dt = max_over_ranks(torch, dist, world, time.perf_counter() - t0)
    kms = float(np.mean(kernel_ms))
    verified = None
    if not args.no_verify:
        # the whole list against the oracle: nearest word of every descriptor, then the integer scoring
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests"))
        import oracle_lib as orc
        vh = vocab.cpu().numpy()
        words = [orc.knn2_u8(arena[i].cpu().numpy(), vh, host_threads()[0])[0] for i in range(nimg)]
        opairs, _ = orc.shortlist(words, nwords, topk)
        ok = np.array_equal(np.asarray(pairs, np.int32).reshape(-1, 2), opairs)
        verified = bool(sum_over_ranks(torch, dist, world, 0 if ok else 1) == 0)
    ops = OPS_PER_EVAL_U8 * float(nimg) * nfeat * nwords    # the word assignment; the rest is < 1 % of the work
    achieved = ops / (kms * 1e-3) / 1e12
    out = {
        "metric": "images shortlisted/sec", "value": world * args.steps * nimg / dt, "unit": "images/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "replicas",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic", "verified": verified,
        "config": {"workload": "vocabulary shortlist, %d images x %d SIFT-128 features, %d words, %d partners per image"
                               % (nimg, nfeat, nwords, topk),
                   "parallelism": "replicas only", "pairs_listed": int(len(pairs))},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": INT8_DENSE_PEAK_TOPS, "unit": "TOP/s",
                     "frac": achieved / INT8_DENSE_PEAK_TOPS, "traffic": None, "kernel": "sift_query_kernel",
                     "kernel_ms": kms,
                     "note": "kernel_ms = the word-assignment launch (descriptors x words distances)"},
    }
    m.close()
    return out


def run_db_query(args, torch, dist, world, rank):
    """16384 float-256 queries against a 1M-row database cut into `world` contiguous slices;
    every rank searches its slice, the per-slice 2-NN lists are all-gathered and merged."""
    import visual_slam_amd as vsm
    from visual_slam_amd import _binding as B

    nq, ndb, dim = 16384, 1000000, 256
    g = torch.Generator(device="cuda")
    g.manual_seed(99)
    q = torch.randn((nq, dim), generator=g, device="cuda")
    q /= q.norm(dim=1, keepdim=True)
    torch.cuda.synchronize()
    per = (ndb + world - 1) // world
    lo, hi = rank * per, min(ndb, (rank + 1) * per)
    g.manual_seed(1000 + rank)
    db = torch.randn((hi - lo, dim), generator=g, device="cuda")
    db /= db.norm(dim=1, keepdim=True)
    torch.cuda.synchronize()
    m = vsm.Matcher(torch.cuda.current_device())
    sq = m.upload_f32_device(q.data_ptr(), nq, dim)
    sdb = m.upload_f32_device(db.data_ptr(), hi - lo, dim)
    m.set_timing(True)
    m.set_f32_mode(args.f32_mode)
    base = np.asarray([min(ndb, r * per) for r in range(world + 1)], np.int64)

    # per-slice result stays on the device: 4 x nq words (keys as float bits, slice-local rows), all-gathered
    # over RCCL and merged by a kernel (key, then global row); only the merged lists travel to the host
    pack = torch.empty((4, nq), dtype=torch.int32, device="cuda")
    allp = torch.empty((world, 4, nq), dtype=torch.int32, device="cuda") if world > 1 else pack

    def step():
        m.knn2_device(sq, sdb, vsm.METRIC_L2, pack.data_ptr())
        if world > 1:
            if REHEARSAL:
                parts = [torch.empty((4, nq), dtype=torch.int32) for _ in range(world)]
                dist.all_gather(parts, pack.cpu())
                allp.copy_(torch.stack(parts))
            else:
                dist.all_gather_into_tensor(allp, pack)
            torch.cuda.synchronize()
        return m.merge_top2_device(allp.data_ptr(), base, world, nq, vsm.METRIC_L2)

    for _ in range(args.warmup):
        step()
    barrier_sync(torch, dist, world)
    t0 = time.perf_counter()
    kernel_ms = []
    for _ in range(args.steps):
        res = step()
        kernel_ms.append(m.last_kernel_ms())
    barrier_sync(torch, dist, world)
    dt = max_over_ranks(torch, dist, world, time.perf_counter() - t0)
    evals = float(args.steps) * nq * ndb
    kms = float(np.mean(kernel_ms))
    verified = None
    if not args.no_verify:
        # every rank: a 64-query sample of its slice's 2-NN (the words that go into the all-gather) against
        # the oracle over the whole slice; with one rank the merged global lists as well
        import oracle_lib as orc
        sample = np.arange(0, nq, nq // 64)[:64]
        qs = q[torch.as_tensor(sample, device="cuda")].cpu().numpy()
        oi, ok1, ok2 = orc.knn2_f32(qs, db.cpu().numpy(), orc.METRIC_L2, host_threads()[0])
        pk = pack.cpu().numpy()
        ok = (np.array_equal(pk[2][sample], oi) and np.array_equal(pk[0][sample].view(np.float32), ok1)
              and np.array_equal(pk[1][sample].view(np.float32), ok2))
        if world == 1:
            ok = ok and np.array_equal(res[2][sample], oi.astype(np.int64)) and np.array_equal(res[0][sample], ok1)
        verified = bool(sum_over_ranks(torch, dist, world, 0 if ok else 1) == 0)
    out = {
        "metric": "descriptor distance evals/sec", "value": evals / dt / 1e9, "unit": "G/s",
        "queries_per_s": args.steps * nq / dt,
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic", "verified": verified,
        "config": {"workload": "16384 float-256 queries vs 1M-row database, k=2, L2",
                   "parallelism": "database sliced x%d, %s of the per-slice 2-NN lists, device merge"
                                  % (world, "all-gather over gloo on host copies (rehearsal on one GPU)" if REHEARSAL else "RCCL all-gather")},
        "roofline": f32_roofline(m, OPS_PER_EVAL_F32 * float(nq) * (hi - lo), kms,
                                 ("void vsm::bf16_coarse_kernel<256, false, false>", "r03_dbquery_mfma_utilisation.txt")
                                 if world == 1 else None),
    }
    if world == 1 and m.last_f32_path() == 0:   # the counter passes were taken at this shape
        out["roofline"]["traffic"] = measured_traffic("void vsm::bf16_coarse_kernel<256, false, false>", "r03_traffic_db_query.json")
        if out["roofline"]["traffic"]:
            out["roofline"]["traffic_note"] = ("HBM bytes per launch of the coarse kernel: a constant read from committed rocprofv3 "
                                               "--pmc passes over the same search (profiles/r03_dbquery_pmc_*.csv), not a "
                                               "measurement of this run")
            out["roofline"]["hbm_gb_per_s"] = out["roofline"]["traffic"] / (kms * 1e-3) / 1e9
            out["roofline"]["hbm_frac_of_peak"] = out["roofline"]["hbm_gb_per_s"] / HBM_PEAK_GBPS
    m.close()
    return out


def run_dry(args, torch, dist, world, rank):
    """The rank / shard bookkeeping of the sharded workloads without a GPU: every rank derives its own
    shard exactly as the real run does, the shards are all-gathered and rank 0 reports them."""
    import visual_slam_amd as vsm

    if args.workload == "db_query":
        n = 1000000
        per = (n + world - 1) // world
        mine = [min(n, rank * per), min(n, (rank + 1) * per)]
    else:
        n = len(vsm.generate_exhaustive_pairs(args.images if args.workload == "sift_exhaustive" else 64))
        mine = list(shard_bounds(n, world, rank))
    shards = [None] * world
    if world > 1:
        dist.all_gather_object(shards, [rank] + mine)
    else:
        shards = [[rank] + mine]
    return {"dry_run": True, "workload": args.workload, "n_gpus": world, "items": n,
            "shards": [[int(v) for v in sh] for sh in shards]}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    torch, dist, world, rank, local = dist_setup(args)
    if args.dry_run:
        out = run_dry(args, torch, dist, world, rank)
        if rank == 0:
            print(json.dumps(out))
        if world > 1:
            dist.destroy_process_group()
        return
    fn = {"sift_exhaustive": run_sift_exhaustive, "superpoint_pair": run_superpoint_pair,
          "superpoint_pairs": run_superpoint_pairs,
          "sift_large_pair": run_sift_large_pair, "db_query": run_db_query, "geom_verify": run_geom_verify,
          "shortlist": run_shortlist}[args.workload]
    out = fn(args, torch, dist, world, rank)
    failed = out.get("verified") is False     # (the same on every rank: the flag is summed over the ranks)
    if failed:
        # a result that differs from the oracle's is not a measurement: no value, no roofline, a failing exit status
        out = {k: v for k, v in out.items() if k not in ("value", "roofline", "pairs_per_s", "queries_per_s")}
        out["error"] = "the timed run's output differs from the CPU oracle's"
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    if failed:
        sys.exit(3)


if __name__ == "__main__":
    main()
