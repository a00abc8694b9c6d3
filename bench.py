#!/usr/bin/env python3
"""bench.py -- query-vectors/sec of brute-force L2 top-1 over a 1M x 512 float32 gallery.

Contract: `python bench.py --gpus N --steps K --warmup W` (N > 1 is launched by
torch.distributed.run, one rank per GPU). A "step" is one batch of `--batch` query vectors
matched against the whole gallery: ceil(batch / queries_per_pass) gallery passes of the scan
kernel plus, for N > 1, one RCCL all-reduce(MIN) over the packed (distance, index) keys.
Queries and gallery are resident in HBM before the timed region starts.

Multi-GPU: the SAME 1M x 512 gallery is sharded by rows over the N ranks ("scaling":
"strong"); every rank scans its shard for all queries and the global nearest neighbour is the
integer minimum of the ranks' packed keys (exact first-minimum tie-break).

Rank 0 prints ONE JSON line. `roofline` is measured live with HIP events around every scan
launch on the stream the kernel runs on; `cpu_baseline` (N = 1 only) times the reference's own
recognize_image_bf (oracle/_ref, built from /root/reference in the build container) on the
host cores over a bounded sample of the same queries, and cross-checks the GPU answers.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import __graft_entry__ as ge  # noqa: E402

CHUNK_ROWS = 15625      # generation granule: 64 chunks make the 1M-row gallery, any 1/2/4/8 sharding is whole chunks


def gen_chunk(chunk, rows, d, device):
    """Rows [chunk*CHUNK_ROWS, ...) of the global synthetic gallery: Uniform[0,1), |x|<1e-4 -> 0,
    L2-normalised (qt_cpp/db_features.cpp:85-101). Depends only on the chunk id."""
    g = torch.Generator(device=device)
    g.manual_seed(1_000_003 * 13 + chunk)
    x = torch.rand((rows, d), generator=g, device=device, dtype=torch.float32)
    x = torch.where(x.abs() < 1e-4, torch.zeros_like(x), x)
    return x / x.norm(dim=1, keepdim=True)


# The contract is ONE JSON line on stdout. Libraries underneath print there too (RCCL writes a five-line version banner to
# stdout when its first communicator comes up), so file descriptor 1 is pointed at stderr for the whole run and the line goes
# to a private duplicate of the original stdout.
_REAL_STDOUT = None


def capture_stdout():
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit_line(text):
    sys.stdout.flush()
    data = (text + "\n").encode()
    fd = _REAL_STDOUT if _REAL_STDOUT is not None else 1
    while data:
        data = data[os.write(fd, data):]


def main():
    capture_stdout()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=512)
    ap.add_argument("--batch", type=int, default=256, help="query vectors per step")
    ap.add_argument("--qpp", type=int, default=0, help="queries per gallery pass (0 = library default)")
    ap.add_argument("--waves", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--no-mfma", action="store_true", help="skip the (untimed) matrix-core cross-check of the same step")
    ap.add_argument("--pmc-child", action="store_true", help="internal: this process IS the counter pass (no nested pass, short run)")
    ap.add_argument("--no-pmc", action="store_true", help="do not start the rocprofv3 counter pass; report the recorded one")
    ap.add_argument("--force-dist", action="store_true", help="initialise the process group even for one rank (exercises the RCCL key exchange on a 1-GPU box)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend; 'gloo' rehearses the N>1 path with all ranks on ONE GPU")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:   # rehearsal: ranks may share a GPU, the key exchange goes through host memory
            dist.init_process_group(args.backend, rank=rank, world_size=world)
            local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    fir = ge.load_package()
    from fast_image_recognition_amd import sharding

    n, d, qb = args.rows, args.dim, args.batch
    row_lo, row_hi = sharding.shard_bounds(n, world, rank, granule=CHUNK_ROWS)
    c_lo, c_hi = row_lo // CHUNK_ROWS, (row_hi + CHUNK_ROWS - 1) // CHUNK_ROWS

    # ---- gallery shard, generated on the device, re-tiled by the library, source freed ----
    shard = torch.empty((row_hi - row_lo, d), device=dev, dtype=torch.float32)
    for c in range(c_lo, c_hi):
        r0 = c * CHUNK_ROWS
        rows = min(CHUNK_ROWS, n - r0)
        shard[r0 - row_lo: r0 - row_lo + rows] = gen_chunk(c, rows, d, dev)
    torch.cuda.synchronize()
    # all timed work runs on one explicit (non-default) stream: the library launches on it and torch
    # orders the RCCL all-reduce after it
    work_stream = torch.cuda.Stream(device=dev)
    stream = work_stream.cuda_stream
    g = fir.Gallery(dev_ptr=shard.data_ptr(), n=shard.shape[0], d=d, metric=fir.METRIC_L2, device=local_rank, stream=stream)
    g.set_row_offset(row_lo)
    if args.qpp or args.waves:
        g.set_tuning(args.qpp, args.waves)

    # ---- queries: even = fresh draws, odd = perturbed copies of known gallery rows of chunk 0 ----
    c0 = gen_chunk(0, min(CHUNK_ROWS, n), d, dev)
    gq = torch.Generator(device=dev)
    gq.manual_seed(424243)
    fresh = torch.rand((qb, d), generator=gq, device=dev)
    planted_rows = (torch.arange(qb, device=dev) * 977 + 11) % c0.shape[0]
    noise = (torch.rand((qb, d), generator=gq, device=dev) - 0.5) * 0.05 * c0.mean()
    pert = (c0[planted_rows] + noise).clamp_min(0)
    q = torch.where((torch.arange(qb, device=dev) % 2 == 0)[:, None], fresh, pert)
    q = (q / q.norm(dim=1, keepdim=True)).contiguous()
    del c0, fresh, pert, noise
    keys = torch.empty(qb, device=dev, dtype=torch.int64)   # packed u64 keys (viewed int64 for torch)
    host_shard = None
    if world == 1 and args.cpu_seconds > 0:
        host_shard = shard.cpu().numpy()
    del shard
    torch.cuda.empty_cache()

    torch.cuda.synchronize()

    def step():
        with torch.cuda.stream(work_stream):
            g.search_top1_keys_dev(q.data_ptr(), qb, keys.data_ptr(), stream=stream)
            if dist is not None:
                k = sharding.keys_as_int64(keys)           # order-preserving u64 -> i64 (x ^ 2^63)
                if args.backend == "nccl":
                    sharding.allreduce_min_keys(k)         # RCCL all-reduce(MIN) over xGMI
                else:
                    kh = k.cpu()
                    sharding.allreduce_min_keys(kh)
                    k = kh.to(dev)
                keys.copy_(sharding.keys_from_int64(k))

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    g.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    t1 = time.perf_counter()
    kernel_ms, bytes_alg = g.profile_read()
    g.profile_enable(False)
    tuning = g.get_tuning()
    elapsed = t1 - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # outside the timed region: the same step through the matrix-core path (fir_gemm_*), which must return the same keys
    mfma = None
    if world == 1 and not args.no_mfma:
        gm = fir.GemmSearch(g)
        k2 = torch.empty_like(keys)
        with torch.cuda.stream(work_stream):
            gm.search_top1_keys_dev(q.data_ptr(), qb, k2.data_ptr(), stream=stream)
            torch.cuda.synchronize()
            tg0 = time.perf_counter()
            for _ in range(5):
                gm.search_top1_keys_dev(q.data_ptr(), qb, k2.data_ptr(), stream=stream)
            torch.cuda.synchronize()
            tg = (time.perf_counter() - tg0) / 5
        mfma = {"kernel": "fp16 MFMA (one term, power-of-two-scaled operands, 128 queries per gallery read) nominates rows, reference arithmetic re-ranks every row inside the rounding window, certificate + exact-scan fallback",
                "queries_per_s": qb / tg, "ms_per_step": tg * 1e3, "tflops_dot_products": 2.0 * n * d * qb / tg / 1e12,
                "gallery_GBps": (row_hi - row_lo) * d * 2.0 * (-(-qb // 128)) / tg / 1e9,     # fp16 fragments: 2 B per feature, once per 128 queries
                "identical_keys_to_scan": bool(torch.equal(keys, k2)), "fallback_queries": gm.stats()["fallback_queries"]}
        gm.close()

    # also outside the timed region (N = 1): the other scans of the same gallery, for the record
    also = None
    if world == 1 and not args.no_mfma:
        def rate(fn, nq, reps):
            fn()
            torch.cuda.synchronize()
            t_0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return nq * reps / (time.perf_counter() - t_0)

        with torch.cuda.stream(work_stream):
            k5 = torch.empty((qb, 5), device=dev, dtype=torch.int64)
            also = {"l2_top5_queries_per_s": rate(lambda: g.search_topk_keys_dev(q.data_ptr(), qb, 5, k5.data_ptr(), stream=stream), qb, 2)}
            top5_first_is_top1 = bool(torch.equal(k5[:, 0], keys))
            # the same step with 16 queries per gallery pass: more queries/s, but the kernel is then bound by the f32 vector
            # pipes (3 un-fused ops per feature and query), not by HBM -- the timed step keeps the library's choice (8)
            k16 = torch.empty_like(keys)
            g.set_tuning(16, 0)
            r16 = rate(lambda: g.search_top1_keys_dev(q.data_ptr(), qb, k16.data_ptr(), stream=stream), qb, 3)
            g.set_tuning(-1, 0)
            also["l2_top1_16_queries_per_pass"] = {"queries_per_s": r16, "gallery_GBps": (row_hi - row_lo) * d * 4.0 * (qb / 16) * (r16 / qb) / 1e9,
                                                   "identical_keys": bool(torch.equal(k16, keys))}
            # chi-square / KL compare non-negative feature vectors that went through the loader's |x| < 1e-4 -> 0 rule
            # (db_features.cpp:85-86) like the gallery rows did; the L2 step's planted queries carry signed noise
            q32 = q[:32].clamp_min(0.0)
            q32 = torch.where(q32 < 1e-4 / 13.0, torch.zeros_like(q32), q32).contiguous()
            k32 = torch.empty(32, device=dev, dtype=torch.int64)
            for name, metric in (("chi2", fir.METRIC_CHI2), ("kl", fir.METRIC_KL)):
                g.set_metric(metric)
                also[f"{name}_top1_queries_per_s"] = rate(lambda: g.search_top1_keys_dev(q32.data_ptr(), 32, k32.data_ptr(), stream=stream), 32, 2)
            g.set_metric(fir.METRIC_L2)
            also["l2_top5_first_column_is_top1"] = top5_first_is_top1

    idx, dd = fir.keys_unpack(keys.cpu().numpy().view(np.uint64))
    # size-independent property at full size: every planted query finds its source row, closer than any fresh one does
    planted = planted_rows.cpu().numpy()
    odd = np.arange(qb) % 2 == 1
    planted_ok = bool(np.all(idx[odd] == planted[odd])) if n >= CHUNK_ROWS else None

    out = None
    if rank == 0:
        launches_per_step = len(kernel_ms) / max(args.steps, 1)
        avg_ms = float(np.mean(kernel_ms)) if len(kernel_ms) else float("nan")
        achieved = bytes_alg / (avg_ms * 1e-3) / 1e9 if len(kernel_ms) else float("nan")
        peak_gbs = fir.device_peak_hbm_gbs(local_rank)            # 8000: MI355X_MICROARCH.md
        traffic, traffic_how = pmc_traffic(args, n, d, world)
        out = {
            "metric": "query-vectors/sec brute-force L2 top-1, 1Mx512 gallery",
            "value": qb * args.steps / elapsed,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{n}x{d} f32 gallery, batched L2 top-1 (configs[1] kernel at the metric's 1Mx512 size)",
                "query_batch": qb,
                "queries_per_pass": tuning["queries_per_pass"],
                "gallery_passes_per_step": -(-qb // max(tuning["queries_per_pass"], 1)),
                "scan_launches_per_step": launches_per_step,
                "waves": tuning["waves"],
                "row_sharding": f"{world} shard(s) of {row_hi - row_lo} rows",
                "planted_queries_found": planted_ok,
                "same_step_through_mfma_path": mfma,
                "other_scans_same_gallery": also,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": peak_gbs,
                "unit": "GB/s",
                "frac": achieved / peak_gbs,
                "traffic": traffic,
                "traffic_source": traffic_how,
                "kernel": "fir::k_scan_l2_lds<1,8,4>" if tuning["queries_per_pass"] == 8 else "fir::k_scan*",
                "kernel_avg_ms": avg_ms,
                "bytes_per_launch": bytes_alg,
            },
        }
        if host_shard is not None:
            out["cpu_baseline"] = cpu_baseline(host_shard, q.cpu().numpy(), idx, dd, args.cpu_seconds)
    g.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        emit_line(json.dumps(out))


def pmc_traffic(args, n, d, world):
    """HBM bytes per scan launch from the PMC counters: FETCH_SIZE (rocprofv3 unit: KiB) x 1024 x 2, as
    MI355X_MICROARCH.md prescribes for 16 B/lane streams on gfx950, averaged over the scan launches. Counters cannot be
    collected from inside the timed run, so after it this process starts `rocprofv3 --pmc FETCH_SIZE -- python3 bench.py
    --pmc-child ...` (its own pass with only that counter, a few steps of the same workload) as a child process and
    reads its CSV. Without rocprofv3 (or if the pass fails) the pass recorded under profiles/ is reported for the shape
    it was taken on. Returns (bytes, how)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    under_profiler = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", "")
    if world == 1 and not args.pmc_child and not args.no_pmc and not under_profiler and shutil.which("rocprofv3"):
        out_dir = tempfile.mkdtemp(prefix="fir_pmc_")
        cmd = ["rocprofv3", "--pmc", "FETCH_SIZE", "--output-format", "csv", "-d", out_dir, "-o", "pmc", "--",
               sys.executable, os.path.abspath(__file__), "--pmc-child", "--steps", "3", "--warmup", "1", "--rows", str(n), "--dim", str(d),
               "--batch", str(args.batch), "--cpu-seconds", "0", "--no-mfma"]
        if args.qpp:
            cmd += ["--qpp", str(args.qpp)]
        try:
            subprocess.run(cmd, cwd=out_dir, env=dict(os.environ, TMPDIR=out_dir), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                           timeout=150, check=True)
            vals = []
            for f in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r["Counter_Name"] == "FETCH_SIZE" and "k_scan_l2" in r["Kernel_Name"]:
                        vals.append(float(r["Counter_Value"]))
            if vals:
                return sum(vals) / len(vals) * 1024 * 2, f"rocprofv3 --pmc FETCH_SIZE child pass of this run, {len(vals)} scan launches"
        except Exception:
            pass
        finally:
            shutil.rmtree(out_dir, ignore_errors=True)
    path = os.path.join(ROOT, "profiles", "r01_rocprofv3_pmc_fetch_size.json")
    if os.path.exists(path) and (n, d, world) == (1_000_000, 512, 1):
        for e in json.load(open(path)):
            if "k_scan_l2" in e["kernel"] and e["counter"] == "FETCH_SIZE":
                return e["bytes_per_launch_corrected"], "recorded pass profiles/r01_rocprofv3_pmc_fetch_size.json"
    return None, None


def cpu_baseline(rows, queries, gpu_idx, gpu_dist, budget_s):
    """The reference's recognize_image_bf (qt_cpp/db_features.cpp:319-335) on the host cores:
    one query per thread at a time (the reference itself is single threaded; queries are
    independent). Falls back to the C restatement when oracle/_ref is not present."""
    import oracle_lib

    cores = max(1, min(len(os.sched_getaffinity(0)), 16))
    d = rows.shape[1]
    kind = "reference" if oracle_lib.have_ref() else "port"
    if kind == "reference":
        db = oracle_lib.load_ref("l2").db(rows, None, 0)
        fn = lambda qv: db.recognize_image_bf(qv, d)  # noqa: E731
    else:
        orc = oracle_lib.load_oracle()
        fn = lambda qv: orc.recognize_bf(rows, qv, 0, d, 0)[0]  # noqa: E731
    # calibrate with one query, then run whole rounds of `cores` queries inside the budget
    t0 = time.perf_counter()
    first = fn(queries[0])
    t_one = time.perf_counter() - t0
    rounds = max(1, int(budget_s / max(t_one * 1.3, 1e-6)))
    nq = min(queries.shape[0], rounds * cores)
    res = [None] * nq
    res[0] = first

    def work(tid):
        for i in range(tid, nq, cores):
            res[i] = fn(queries[i])

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(t,)) for t in range(cores)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    dt = time.perf_counter() - t0
    agree = int(sum(int(res[i]) == int(gpu_idx[i]) for i in range(nq)))
    return {
        "value": nq / dt,
        "unit": "queries/s",
        "cores": cores,
        "kind": kind,
        "sample": f"{nq} of the step's {queries.shape[0]} queries against the full {rows.shape[0]}x{d} gallery, "
                  f"{cores} threads x recognize_image_bf ({t_one:.2f} s/query/thread); GPU index identical on {agree}/{nq}",
    }


if __name__ == "__main__":
    main()
