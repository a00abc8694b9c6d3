#!/usr/bin/env python3
"""bench.py -- query-vectors/sec of brute-force L2 top-1 over a 1M x 512 float32 gallery.

Contract: `python bench.py --gpus N --steps K --warmup W`. Invoked plainly with N > 1 it starts the N ranks itself
(`python -m torch.distributed.run`, one child process per GPU, before this process has made any GPU call) and relays
rank 0's line; invoked by torch.distributed.run it is one of those ranks.

A "step" is one batch of `--batch` query vectors matched against the whole gallery through the library's default
dispatch (fir_search_top1_keys_dev): for batches >= 128 that is the fp16 matrix-core nomination pass + exact re-rank +
certificate (identical keys to the exact scan, checked in this run), plus, for N > 1, the RCCL all-reduce(MIN) of the
packed (distance, index) keys that the LIBRARY issues (fir_sharded_*, ncclAllReduce(ncclMin, ncclUint64)). Queries and
gallery are resident in HBM before the timed region starts. `value` is that loop.

The same run also times, with the same steps / warmup:
  * the exact streaming scan at 8 queries per gallery pass (matrix-core path switched off, 256-query steps): the
    metric's "achieved HBM GB/s" clause -> `roofline`;
  * N = 1: chi-square / KL / top-5 scans of the same gallery (BASELINE config 3), the 100k x 512 gallery (config 2),
    the float64 PNN / kNN classifiers (K3);
  * a 10M x 512 gallery split over the N ranks (BASELINE config 4) -> `config4`;
  * N = 1: the reference's own recognize_image_bf and the OpenMP restatement on ALL host cores -> `cpu_baseline`, `cpu_all`.

Multi-GPU: the SAME 1M x 512 gallery is sharded by rows over the N ranks ("scaling": "strong"); `config4` is the
larger gallery of BASELINE.json configs[3].

Rank 0 prints ONE JSON line. Kernel times are HIP events on the stream the kernels run on (fir_profile_read); kernel
names, grids, LDS bytes and registers come from the library (fir_gallery_last_dispatch), not from this script.
"""
import argparse
import json
import os
import socket
import subprocess
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
PEAK_MFMA_F16_TFLOPS = 2500.0      # dense fp16 / bf16 MFMA peak, MI355X_MICROARCH.md
L2_TO_CU_GBPS = (66.0, 73.0)        # what one CU takes from its XCD's L2, rows served from L2, ~72 KiB in flight (MI355X_MICROARCH.md, "Indexed rows: gather into LDS")
N_CUS = 256


def l2_to_cu(n_rows, d, queries_per_launch, kernel_ms):
    """The gallery stream of a matrix-core launch priced against the L2 -> CU path ("what bounds the 16-row kernels"):
    every fp16 fragment is read once by each of the queries_per_launch / 128 workgroups that hold a 128-query tile (one of them from HBM,
    the others from the XCD's L2); rows longer than 512 features re-stream the query slabs as well, half as many bytes again."""
    dk = (d + 31) // 32 * 32
    tiles = queries_per_launch / 128.0
    stream = tiles * n_rows * dk * 2.0 * (1.5 if dk > 512 else 1.0)
    per_cu = stream / (kernel_ms * 1e-3) / 1e9 / N_CUS
    return {"bytes_per_launch_through_l1": stream, "GBps_per_cu": per_cu, "ceiling_GBps_per_cu": list(L2_TO_CU_GBPS),
            "frac_of_ceiling": per_cu / L2_TO_CU_GBPS[0], "bytes_per_mfma": 128 * (1.5 if dk > 512 else 1.0)}
PEAK_VALU_WAVE_INSTR_PER_S = 256 * 4 * 2.4e9 / 4.0   # one wave64 VALU instruction per 4 cycles per SIMD (profiles/r01_ubench_valu_issue_rate.txt)


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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=512)
    ap.add_argument("--batch", type=int, default=131072, help="query vectors per step of the headline loop (131 072: the driver's 20 steps then time > 2 s of GPU work)")
    ap.add_argument("--scan-batch", type=int, default=256, help="query vectors per step of the exact-scan loop (the HBM roofline)")
    ap.add_argument("--qpp", type=int, default=0, help="queries per gallery pass of the exact scan (0 = library default)")
    ap.add_argument("--waves", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of each CPU baseline leg (0 = skip)")
    ap.add_argument("--no-extras", "--no-mfma", dest="no_extras", action="store_true",
                    help="only the two timed loops: skip the config-2/3 scans, the K3 classifiers and config 4")
    ap.add_argument("--config4-rows", type=int, default=10_000_000, help="rows of the config-4 gallery (0 = skip)")
    ap.add_argument("--config5-dim", type=int, default=1280, help="row length of the config-5 gallery (BASELINE configs[4]; 0 = skip)")
    ap.add_argument("--extra-batch", type=int, default=32768, help="query vectors per call of the config-4 / config-5 / other-scan blocks")
    ap.add_argument("--no-verify", action="store_true", help="skip the exact-scan verification of the WHOLE headline batch outside the timed regions (profiling runs: "
                                                              "the kernel table then holds the two timed loops only); the first scan-batch queries are still compared")
    ap.add_argument("--pmc-child", action="store_true", help="internal: this process IS the counter pass (no nested pass, short run)")
    ap.add_argument("--pmc-config2", action="store_true", help="internal (counter pass): run the 100k x 512 cold / warm sequence first")
    ap.add_argument("--no-pmc", action="store_true", help="do not start the rocprofv3 counter pass; report the recorded one")
    ap.add_argument("--force-dist", action="store_true", help="one rank, but through the sharded handle and its RCCL communicator")
    ap.add_argument("--shards-per-device", type=int, default=1, help="logical shards per rank (exercises the split on few GPUs)")
    ap.add_argument("--backend", default="nccl", help="'gloo' rehearses the N>1 path with all ranks on ONE GPU (keys exchanged through host memory)")
    ap.add_argument("--dry-run", action="store_true", help="rendezvous, shard arithmetic and key exchange only, no GPU work (CPU test of the launch path)")
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(args):
    """`python bench.py --gpus N` invoked plainly: start N rank processes (fresh children; this parent has made no GPU
    call and makes none), wait, relay rank 0's JSON line."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    p = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    line = None
    for ln in p.stdout.decode(errors="replace").splitlines():
        ln = ln.strip()
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        elif ln:
            print(ln, file=sys.stderr)
    if p.returncode != 0 or line is None:
        print(f"bench.py: the {args.gpus}-rank run failed (exit {p.returncode}, result line {'found' if line else 'missing'})", file=sys.stderr)
        sys.exit(p.returncode or 1)
    emit_line(line)


def host_cores():
    """CPUs this process may actually use: the affinity mask, cut to the cgroup's CPU quota when one is set (a one-GPU box hands
    out a share of the host: 256 hardware threads in the mask, 16 CPUs of quota)."""
    n = max(1, len(os.sched_getaffinity(0)))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, int(quota / period + 0.5)))
        except (OSError, ValueError):
            pass
    return n


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def dry_run(args, world, rank):
    """No GPU: the ranks rendezvous (gloo), split the rows and reduce made-up packed keys exactly as the real run does."""
    import torch.distributed as dist
    fir = ge.load_package()
    from fast_image_recognition_amd import sharding

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = sharding.shard_bounds(args.rows, world, rank, granule=CHUNK_ROWS)
    qb = args.batch
    # query i's best row lives in shard i % world: that rank holds (distance 0.25, row), the others a worse one of their own
    keys = np.array([fir.key_pack(0.25 if i % world == rank else 0.5, min(lo + i, max(hi - 1, lo))) for i in range(qb)], np.uint64)
    t = sharding.keys_as_int64(torch.from_numpy(keys.view(np.int64)).clone())
    if world > 1:
        sharding.allreduce_min_keys(t)
    merged = sharding.keys_from_int64(t).numpy().view(np.uint64)
    idx, dd = fir.keys_unpack(merged)
    ok = bool(np.all(dd == np.float32(0.25)))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        emit_line(json.dumps({"metric": "query-vectors/sec brute-force L2 top-1, 1Mx512 gallery", "value": None, "unit": "queries/s", "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "strong",
                              "vs_baseline": None, "dtype": "f32", "data": "synthetic", "dry_run": True,
                              "config": {"workload": "dry run: rendezvous + row split + key exchange only", "row_sharding": f"{world} shard(s)",
                                         "exchange_ok": ok}}))


class Matcher:
    """One rank's view of the (sharded) gallery: `step(q, qb, keys)` = scan of the local rows + the exchange."""

    def __init__(self, fir, args, shard, n_local, d, row_lo, world, rank, local_rank, dist, dev, work_stream):
        from fast_image_recognition_amd import sharding
        self.fir, self.args, self.dist, self.dev, self.world = fir, args, dist, dev, world
        self.sharding = sharding
        self.ws = work_stream
        self.stream = work_stream.cuda_stream
        self.sh = None
        self.torch_nccl = False
        self.in_library_rccl = (dist is not None and args.backend == "nccl") or args.shards_per_device > 1 or args.force_dist
        if self.in_library_rccl:
            # the key exchange is the library's: every rank passes the id rank 0 made
            idt = torch.zeros(fir.capi.COMM_ID_BYTES, dtype=torch.uint8, device=dev)
            if rank == 0:
                idt.copy_(torch.frombuffer(bytearray(fir.comm_unique_id()), dtype=torch.uint8))
            if dist is not None and world > 1:
                dist.broadcast(idt, 0)
            err = None
            try:
                self.sh = fir.ShardedGallery(dev_ptr=shard.data_ptr(), n=n_local, d=d, metric=fir.METRIC_L2, devices=[local_rank],
                                             shards_per_device=args.shards_per_device, first_global_row=row_lo,
                                             comm_id=bytes(idt.cpu().numpy().tobytes()), proc_rank=rank, nprocs=world)
            except fir.FirError as e:      # the ranks must agree on what happens next
                err = e
            ok = torch.tensor([0 if err else 1], device=dev, dtype=torch.int32)
            if dist is not None and world > 1:
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                if world == 1 or dist is None:
                    raise err
                print(f"bench.py: rank {rank}: the library's RCCL communicator could not be created ({err}); every rank falls back to torch.distributed's", file=sys.stderr)
                if self.sh is not None:
                    self.sh.close()
                    self.sh = None
                self.in_library_rccl = False
                self.torch_nccl = True
        if self.sh is not None:
            self.parts = [self.sh.shard(i)[0] for i in range(self.sh.info()["nshards"])]
            self.parts = [p for p in self.parts if p is not None]
            self.g = self.parts[0]
        else:
            self.g = fir.Gallery(dev_ptr=shard.data_ptr(), n=n_local, d=d, metric=fir.METRIC_L2, device=local_rank, stream=self.stream)
            self.g.set_row_offset(row_lo)
            self.parts = [self.g]

    def set_mfma(self, v):
        for p in self.parts:
            p.set_large_batch_mfma(v)

    def set_tuning(self, qpp, waves):
        for p in self.parts:
            p.set_tuning(qpp, waves)

    def step(self, q, qb, keys):
        with torch.cuda.stream(self.ws):
            if self.sh is not None:
                self.sh.search_top1_keys_dev(q.data_ptr(), qb, keys.data_ptr(), stream=self.stream)      # scan + ncclAllReduce(min, u64)
                return
            self.g.search_top1_keys_dev(q.data_ptr(), qb, keys.data_ptr(), stream=self.stream)
            if self.dist is not None and self.torch_nccl:      # fallback: torch.distributed's RCCL all-reduce(MIN) on the order-preserving int64 view
                k = self.sharding.keys_as_int64(keys)
                self.sharding.allreduce_min_keys(k)
                keys.copy_(self.sharding.keys_from_int64(k))
            elif self.dist is not None:          # gloo rehearsal: ranks share a GPU, the keys go through host memory
                k = self.sharding.keys_as_int64(keys)
                kh = k.cpu()
                self.sharding.allreduce_min_keys(kh)
                keys.copy_(self.sharding.keys_from_int64(kh.to(self.dev)))

    def close(self):
        if self.sh is not None:
            self.sh.close()
        else:
            self.g.close()


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        capture_stdout()
        return spawn_ranks(args)
    capture_stdout()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_run:
        return dry_run(args, world, rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:   # rehearsal: ranks may share a GPU
            dist.init_process_group(args.backend, rank=rank, world_size=world)
            local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    fir = ge.load_package()
    from fast_image_recognition_amd import sharding

    d = args.dim
    work_stream = torch.cuda.Stream(device=dev)   # all timed work runs on one explicit (non-default) stream

    if args.pmc_child and args.pmc_config2:
        pmc_config2_sequence(fir, dev, work_stream, d)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def build(n):
        """This rank's row block of the n-row synthetic gallery, generated on the device -> (Matcher, row_lo, row_hi, host copy or None)."""
        row_lo, row_hi = sharding.shard_bounds(n, world, rank, granule=CHUNK_ROWS)
        shard = torch.empty((row_hi - row_lo, d), device=dev, dtype=torch.float32)
        for c in range(row_lo // CHUNK_ROWS, (row_hi + CHUNK_ROWS - 1) // CHUNK_ROWS):
            r0 = c * CHUNK_ROWS
            rows = min(CHUNK_ROWS, n - r0)
            shard[r0 - row_lo: r0 - row_lo + rows] = gen_chunk(c, rows, d, dev)
        torch.cuda.synchronize()
        m = Matcher(fir, args, shard, row_hi - row_lo, d, row_lo, world, rank, local_rank, dist if world > 1 else None, dev, work_stream)
        return m, row_lo, row_hi, shard

    def make_queries(qb, n):
        """even = fresh draws, odd = perturbed copies of known gallery rows of chunk 0"""
        c0 = gen_chunk(0, min(CHUNK_ROWS, n), d, dev)
        gq = torch.Generator(device=dev)
        gq.manual_seed(424243)
        fresh = torch.rand((qb, d), generator=gq, device=dev)
        planted_rows = (torch.arange(qb, device=dev) * 977 + 11) % c0.shape[0]
        noise = (torch.rand((qb, d), generator=gq, device=dev) - 0.5) * 0.05 * c0.mean()
        pert = (c0[planted_rows] + noise).clamp_min(0)
        q = torch.where((torch.arange(qb, device=dev) % 2 == 0)[:, None], fresh, pert)
        return (q / q.norm(dim=1, keepdim=True)).contiguous(), planted_rows.cpu().numpy()

    def timed_loop(m, q, qb, keys, steps, warmup):
        """W untimed + K timed steps, barrier + synchronize on both sides, MAX over ranks; the launches of the dominant
        kernel are bracketed by HIP events inside the library."""
        for _ in range(warmup):
            m.step(q, qb, keys)
        fence()
        for p in m.parts:
            p.profile_enable(True)
        if m.sh is not None:
            m.sh.profile_enable(True)
        st0 = [p.mfma_stats() for p in m.parts]
        t0 = time.perf_counter()
        for _ in range(steps):
            m.step(q, qb, keys)
        fence()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        if m.sh is not None:
            m.sh.sync()                      # the asynchronous calls' sticky status: a peer's failure during the loop surfaces here, not never
        kernel_ms = np.concatenate([p.profile_read()[0] for p in m.parts]) if m.parts else np.zeros(0)
        exch_ms = m.sh.profile_read() if m.sh is not None else np.zeros(0)
        for p in m.parts:
            p.profile_enable(False)
        if m.sh is not None:
            m.sh.profile_enable(False)
        disp = m.g.last_dispatch()
        st1 = [p.mfma_stats() for p in m.parts]
        # queries of the timed steps whose first certificate did not hold (second matrix-core pass) / that the exact device scan answered
        disp["second_pass_queries"] = sum(b["second_pass_queries"] - a["second_pass_queries"] for a, b in zip(st0, st1))
        disp["fallback_queries"] = sum(b["fallback_queries"] - a["fallback_queries"] for a, b in zip(st0, st1))
        return elapsed, kernel_ms, exch_ms, disp

    n, qb = args.rows, args.batch
    m, row_lo, row_hi, shard = build(n)
    host_shard = shard.cpu().numpy() if (world == 1 and args.cpu_seconds > 0 and not args.pmc_child) else None
    small_src = shard[:100_000].clone() if (world == 1 and not args.no_extras and n >= 100_000) else None
    del shard
    torch.cuda.empty_cache()
    q, planted = make_queries(qb, n)
    keys = torch.empty(qb, device=dev, dtype=torch.int64)      # packed u64 keys (viewed int64 for torch)

    # ---- headline: the library's default dispatch ----
    elapsed, k_ms, x_ms, disp = timed_loop(m, q, qb, keys, args.steps, args.warmup)
    keys_default = keys.clone()
    head = {"elapsed": elapsed, "kernel_ms": k_ms, "exch_ms": x_ms, "disp": disp}
    # "t_batch includes H2D of queries and D2H of results; report kernel-only too" -- the same step through the HOST-pointer entry
    # point (fir_search_top1: what the reference's call takes, db_features.cpp:319): queries from pinned host memory, indices and distances back
    # to host arrays, inside the timed region. `value` stays the device-resident rate.
    host_rate = None
    if world == 1 and not args.pmc_child and m.sh is None:
        qh = q.cpu().pin_memory().numpy()
        hsteps = max(2, args.steps // 5)
        hidx, _ = m.g.search_top1(qh)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(hsteps):
            hidx, hdist = m.g.search_top1(qh)
        hdt = time.perf_counter() - t0
        kidx, kdist = fir.keys_unpack(keys_default.cpu().numpy().view(np.uint64))
        host_rate = {"value_host_pointers": qb * hsteps / hdt, "unit": "queries/s", "steps": hsteps, "ms_per_step": hdt / hsteps * 1e3,
                     "bytes_up_per_step": int(qb) * d * 4, "bytes_down_per_step": int(qb) * 8,
                     "identical_to_the_device_pointer_call": bool(np.array_equal(hidx, kidx) and np.array_equal(hdist.view(np.uint32), kdist.view(np.uint32))),
                     "what": "fir_search_top1 (host pointers): H2D of the queries super-batch by super-batch under the passes before, D2H of indices and distances, all inside the timed region"}
        del qh
    mem_report = m.g.memory_bytes()            # fir_gallery_memory_bytes: the tiled rows and what the default dispatch added to them

    # ---- the exact streaming scan, matrix-core path off: the metric's HBM clause ----
    sqb = min(args.scan_batch, qb)
    m.set_mfma(0)
    if args.qpp or args.waves:
        m.set_tuning(args.qpp, args.waves)
    skeys = torch.empty(sqb, device=dev, dtype=torch.int64)
    s_elapsed, s_ms, s_x_ms, s_disp = timed_loop(m, q, sqb, skeys, args.steps, args.warmup)
    tuning = m.g.get_tuning()
    identical = bool(torch.equal(skeys, keys_default[:sqb]))
    # every query of the headline batch through the exact scan once (outside the timed regions): the keys must be identical
    if not args.pmc_child and not args.no_verify and qb > sqb:
        ek = torch.empty(qb, device=dev, dtype=torch.int64)
        m.step(q, qb, ek)
        torch.cuda.synchronize()
        identical = identical and bool(torch.equal(ek, keys_default))
        del ek
    m.set_mfma(-1)
    if args.qpp or args.waves:
        m.set_tuning(-1, 0)

    idx, dd = fir.keys_unpack(keys_default.cpu().numpy().view(np.uint64))
    odd = np.arange(qb) % 2 == 1
    planted_ok = bool(np.all(idx[odd] == planted[odd])) if n >= CHUNK_ROWS else None   # every planted query finds its source row

    also = None
    cfg2 = None
    k3 = None
    small = None
    if world == 1 and not args.no_extras and not args.pmc_child:
        xb = min(args.extra_batch, qb)
        small = small_batches(fir, m.g, q, dev, work_stream)
        also = other_scans(fir, m.g, q[:xb], keys_default[:xb], dev, work_stream, n, d)
        k3 = k3_classifiers(fir, dev, args)
        if small_src is not None:
            cfg2 = config2(fir, small_src, q, dev, work_stream, d)
    del small_src
    m_n = row_hi - row_lo

    cpu = None
    if host_shard is not None:
        cpu = cpu_baselines(host_shard, q.cpu().numpy(), idx, dd, args.cpu_seconds)
        del host_shard
    m.close()
    torch.cuda.empty_cache()

    # ---- BASELINE config 4: 10M x 512 split over the ranks ----
    cfg4 = None
    if args.config4_rows > 0 and not args.no_extras and not args.pmc_child:
        cfg4 = config4(args, build, make_queries, timed_loop, fir, dev, world)

    # ---- BASELINE config 5: 1M x 1280 (EfficientNet-B7 width), matrix-core nomination + exact re-rank vs the memory-bound scan ----
    cfg5 = None
    if world == 1 and args.config5_dim > 0 and not args.no_extras and not args.pmc_child:
        cfg5 = config5(args, fir, dev, work_stream)

    out = None
    if rank == 0:
        peak_gbs = fir.device_peak_hbm_gbs(local_rank)            # 8000: MI355X_MICROARCH.md
        traffic = pmc_traffic(args, n, d, world)

        def kernel_block(ms, dsp):
            avg = float(np.mean(ms)) if len(ms) else float("nan")
            return avg, {"kernel": dsp["kernel"], "kernel_avg_ms": avg, "launches_timed": int(len(ms)), "grid": dsp["grid"], "block": dsp["block"],
                         "lds_bytes_per_workgroup": dsp["lds_bytes"], "vgprs": dsp["vgprs"], "queries_per_gallery_read": dsp["queries_per_pass"],
                         "bytes_per_launch": dsp["bytes_per_launch"]}

        s_avg, s_blk = kernel_block(s_ms, s_disp)
        s_ach = s_disp["bytes_per_launch"] / (s_avg * 1e-3) / 1e9 if len(s_ms) else float("nan")
        roofline = {"bound": "hbm", "achieved": s_ach, "peak": peak_gbs, "unit": "GB/s", "frac": s_ach / peak_gbs,
                    "traffic": traffic.get("scan"), "traffic_source": traffic.get("how"),
                    "measured_in": f"exact-scan loop of this run: {args.steps} steps of {sqb} queries, matrix-core path off, {s_elapsed / args.steps * 1e3:.3f} ms per step, "
                                   f"{sqb * args.steps / s_elapsed:.0f} queries/s",
                    "queries_per_s": sqb * args.steps / s_elapsed, **s_blk}
        h_avg, h_blk = kernel_block(head["kernel_ms"], head["disp"])
        hd = head["disp"]
        roofline_mfma = None
        if hd["path"] == "mfma" and len(head["kernel_ms"]):
            tf = hd["flops_per_launch"] / (h_avg * 1e-3) / 1e12
            gbs = hd["bytes_per_launch"] / (h_avg * 1e-3) / 1e9
            roofline_mfma = {"bound": "mfma" if tf / PEAK_MFMA_F16_TFLOPS >= gbs / peak_gbs else "hbm",
                             "note": f"fp16 gallery fragments leave HBM once per {hd['queries_per_pass']} queries (the pairs of a launch walk the same rows together and share the stream through L2)",
                             "flops_per_launch": hd["flops_per_launch"], "achieved_tflops": tf, "peak_tflops": PEAK_MFMA_F16_TFLOPS, "frac_of_mfma_peak": tf / PEAK_MFMA_F16_TFLOPS,
                             "stream_GBps": gbs, "peak_GBps": peak_gbs, "frac_of_hbm_peak": gbs / peak_gbs, "traffic": traffic.get("mfma"),
                             "kernel_time_share_of_step": float(np.sum(head["kernel_ms"])) / (head["elapsed"] * 1e3) if world == 1 else None,
                             "l2_to_cu": l2_to_cu(row_hi - row_lo, d, hd["queries_per_pass"], h_avg), **h_blk}
        out = {
            "metric": "query-vectors/sec brute-force L2 top-1, 1Mx512 gallery",
            "value": qb * args.steps / head["elapsed"],
            "value_host_pointers": host_rate["value_host_pointers"] if host_rate else None,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": head["elapsed"] / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{n}x{d} f32 gallery, batched L2 top-1 through the library's default dispatch (the metric's 1Mx512 size; configs[1]'s kernel is the exact-scan loop under `roofline`)",
                "query_batch": qb,
                "path": hd["path"],
                "path_note": "fp16 MFMA (one term, power-of-two-scaled operands, 128 queries per gallery read) nominates rows, the reference's f32 arithmetic re-ranks every row "
                             "inside the rounding window, a rounding-error certificate proves the rest cannot win, uncertified queries go through the exact scan" if hd["path"] == "mfma" else "exact streaming scan",
                "identical_keys_to_exact_scan": identical,
                "fallback_queries_in_timed_steps": hd.get("fallback_queries"), "second_pass_queries_in_timed_steps": hd.get("second_pass_queries"),
                "hbm_bytes_held": mem_report,
                "host_pointer_step": host_rate,
                "small_batches_1mx512": small,
                "planted_queries_found": planted_ok,
                "row_sharding": f"{world} rank(s) x {args.shards_per_device} shard(s), {m_n} rows on this rank",
                "key_exchange": ("RCCL ncclAllReduce(ncclMin, ncclUint64) issued by libfir_amd.so (fir_sharded_search_top1_keys_dev)" if m.in_library_rccl
                                 else "torch.distributed RCCL all-reduce(MIN) (the library's communicator could not be created)" if m.torch_nccl
                                 else "torch.distributed gloo through host memory (rehearsal)") if (world > 1 or m.in_library_rccl) else None,
                "exchange_us_per_step": float(np.mean(head["exch_ms"]) * 1e3) if len(head["exch_ms"]) else None,
                "exact_scan": {"queries_per_pass": tuning["queries_per_pass"], "waves": tuning["waves"], "query_batch": sqb,
                               "gallery_passes_per_step": -(-sqb // max(tuning["queries_per_pass"], 1)), "launches_per_step": len(s_ms) / max(args.steps, 1) / max(len(m.parts), 1),
                               "exchange_us_per_step": float(np.mean(s_x_ms) * 1e3) if len(s_x_ms) else None},
                "kl_note": "KL top-1 identity is tolerance-graded (device v_log_f32 vs glibc logf: distances within 1e-5 relative, same winner unless the runner-up is closer than that); L2 and chi-square are bit-exact",
                "other_scans_same_gallery": also,
                "config2_100kx512": dict(cfg2, dram_bytes_cold_vs_warm=traffic.get("config2")) if cfg2 else None,
                "k3_float64_classifiers": k3,
            },
            "roofline": roofline,
            "roofline_mfma": roofline_mfma,
            "config4": cfg4,
            "config5": cfg5,
        }
        if also:
            for kname in ("roofline_chi2", "roofline_kl", "roofline_chi2_exact", "roofline_kl_exact"):
                out[kname] = also.pop(kname, None)
        if cpu is not None:
            out["cpu_baseline"] = cpu["reference"]
            out["cpu_all"] = cpu["all"]
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        emit_line(json.dumps(out))


def rate(fn, nq, reps):
    fn()
    torch.cuda.synchronize()
    t_0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return nq * reps / (time.perf_counter() - t_0)


# VALU issue slots (4 cycles of a SIMD's vector issue) per (row, feature, query) element of the plain-range chi-square / KL
# scans, counted in the compiled loops of k_scan<8, 3 / 4, ...> (llvm-objdump of libfir_amd.so, 384 elements per loop body):
# chi-square 10.0 vector instructions (2.94 v_pk_fma, 0.98 v_pk_add, 0.98 v_pk_mul, 1 v_add, 1 v_max, 1 v_rcp, ...) + 1 for the
# 8-cycle v_rcp_f32 = 11.0; KL 24.9 + 2.96 for v_rcp + 2 v_log = 27.9. The s_nop hazard padding the compiler adds (1.7 / 5.7
# per element) is not counted: it is part of what keeps the achieved rate below the peak.
CHI2_SLOTS, KL_SLOTS = 11.0, 27.9
# ... and of the nomination scans (csrc/fir_kernels.h, TileAcc::chunk<kChi2Harm / kKLEnt>; DESIGN section 4 "chi-square and KL nomination")
NOM_SLOTS = {"chi2": 2.25, "kl": 3.0}
NOM_MODEL = {"chi2": "harmonic form, two terms per reciprocal: per pair of features and pair of queries four packed adds / multiplies, two v_rcp_f32 (two issue slots each) and one packed fma "
                     "= 2.25 issue slots per (row, feature, query)",
             "kl": "entropy form: per feature and pair of queries a packed add, two v_log_f32 (two issue slots each) and a packed fma = 3.0 issue slots per (row, feature, query)"}
NOM_CLOCK_NOTE = {"chi2": "rocprofv3 --pmc pass of this kernel (profiles/r04_rocprofv3_pmc_chi2_nominate.json): 20.75 M cycles per 256-query launch of ~10.9 ms = 1.90 GHz sustained, "
                          "SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x kernel cycles) = 0.956: the vector pipes are 96 % busy at the clock the chip holds; `frac` prices it at 2.4 GHz",
                  "kl": "same kernel family as the chi-square nomination scan (profiles/r04_rocprofv3_pmc_chi2_nominate.json: vector pipes 96 % busy at the 1.9 GHz the chip sustains); `frac` prices it at 2.4 GHz"}


def small_batches(fir, g, q, dev, ws):
    """The headline gallery (1M x 512) at small batches: exact scan, forced matrix cores, default dispatch; queries whose first certificate did not hold
    (they take a second matrix-core pass on the device) and queries the exact device scan had to answer, per call. VERDICT r3 item 1."""
    stream = ws.cuda_stream
    out = {}
    with torch.cuda.stream(ws):
        for sq in (1, 2, 4, 8, 32, 256):
            ks = torch.empty(sq, device=dev, dtype=torch.int64)
            km = torch.empty(sq, device=dev, dtype=torch.int64)
            kd = torch.empty(sq, device=dev, dtype=torch.int64)
            g.set_large_batch_mfma(0)
            r_scan = rate(lambda: g.search_top1_keys_dev(q.data_ptr(), sq, ks.data_ptr(), stream=stream), sq, 5)
            g.set_large_batch_mfma(1)
            r_mfma = rate(lambda: g.search_top1_keys_dev(q.data_ptr(), sq, km.data_ptr(), stream=stream), sq, 10)
            g.set_large_batch_mfma(-1)
            for _ in range(18 if sq == 1 else 1):          # (a gallery's first 16 one-query calls take the exact scan: the fp16 copy is not built for a handful)
                g.search_top1_keys_dev(q.data_ptr(), sq, kd.data_ptr(), stream=stream)
            st0 = g.mfma_stats()
            r_def = rate(lambda: g.search_top1_keys_dev(q.data_ptr(), sq, kd.data_ptr(), stream=stream), sq, 10)
            st1 = g.mfma_stats()
            dd = g.last_dispatch()
            out[str(sq)] = {"exact_scan_queries_per_s": r_scan, "matrix_core_queries_per_s": r_mfma, "default_dispatch_queries_per_s": r_def,
                            "default_over_best_of_the_two_forms": r_def / max(r_scan, r_mfma), "default_dispatch_path": dd["path"], "default_dispatch_kernel": dd["kernel"],
                            "identical_keys": bool(torch.equal(ks, km)) and bool(torch.equal(ks, kd)),
                            "second_pass_queries_per_call": (st1["second_pass_queries"] - st0["second_pass_queries"]) / 11.0,
                            "fallback_queries_per_call": (st1["fallback_queries"] - st0["fallback_queries"]) / 11.0}
        g.set_large_batch_mfma(-1)
    return out


def other_scans(fir, g, q, keys, dev, ws, n, d):
    """Outside the timed regions (N = 1): the other scans of the same gallery -- BASELINE config 3's chi-square / KL / top-5."""
    stream = ws.cuda_stream
    qb = min(q.shape[0], 256)
    with torch.cuda.stream(ws):
        k5 = torch.empty((qb, 5), device=dev, dtype=torch.int64)
        also = {"l2_top5_queries_per_s": rate(lambda: g.search_topk_keys_dev(q.data_ptr(), qb, 5, k5.data_ptr(), stream=stream), qb, 2)}
        also["l2_top5_first_column_is_top1"] = bool(torch.equal(k5[:, 0], keys[:qb]))
        # the whole query batch through the default dispatch (matrix-core nomination + exact re-rank of the K-th window), and
        # the exact top-K scan on the first 256 of them
        qall = q.shape[0]
        k5a = torch.empty((qall, 5), device=dev, dtype=torch.int64)
        r5a = rate(lambda: g.search_topk_keys_dev(q.data_ptr(), qall, 5, k5a.data_ptr(), stream=stream), qall, 2)
        path5 = g.last_dispatch()["path"]
        g.set_large_batch_mfma(0)
        r5e = rate(lambda: g.search_topk_keys_dev(q.data_ptr(), qb, 5, k5.data_ptr(), stream=stream), qb, 1)
        g.set_large_batch_mfma(-1)
        also["l2_top5_whole_batch"] = {"query_batch": qall, "queries_per_s": r5a, "path": path5, "exact_topk_scan_queries_per_s": r5e,
                                       "identical_keys_to_exact_topk_scan": bool(torch.equal(k5a[:qb], k5))}
        del k5a
        # the exact scan with 16 queries per gallery pass: more queries/s, but bound by the f32 vector pipes (3 un-fused ops per
        # feature and query), not by HBM
        k16 = torch.empty(qb, device=dev, dtype=torch.int64)
        g.set_large_batch_mfma(0)
        g.set_tuning(16, 0)
        r16 = rate(lambda: g.search_top1_keys_dev(q.data_ptr(), qb, k16.data_ptr(), stream=stream), qb, 3)
        g.set_tuning(-1, 0)
        g.set_large_batch_mfma(-1)
        also["l2_top1_exact_scan_16_queries_per_pass"] = {"queries_per_s": r16, "gallery_GBps": n * d * 4.0 * (r16 / 16) / 1e9,
                                                          "identical_keys": bool(torch.equal(k16, keys[:qb]))}
        # chi-square / KL compare non-negative feature vectors that went through the loader's |x| < 1e-4 -> 0 rule
        # (db_features.cpp:85-86) like the gallery rows did; the L2 step's planted queries carry signed noise
        q32 = q[:32].clamp_min(0.0)
        q32 = torch.where(q32 < 1e-4 / 13.0, torch.zeros_like(q32), q32).contiguous()
        k32 = torch.empty(32, device=dev, dtype=torch.int64)
        k32_5 = torch.empty((32, 5), device=dev, dtype=torch.int64)
        for name, metric, slots in (("chi2", fir.METRIC_CHI2, CHI2_SLOTS), ("kl", fir.METRIC_KL, KL_SLOTS)):
            g.set_metric(metric)
            g.profile_enable(True)
            r1 = rate(lambda: g.search_top1_keys_dev(q32.data_ptr(), 32, k32.data_ptr(), stream=stream), 32, 2)
            ms, _ = g.profile_read()
            g.profile_enable(False)
            dsp = g.last_dispatch()
            plain = g.value_range()
            k32x = k32.clone()
            r5 = rate(lambda: g.search_topk_keys_dev(q32.data_ptr(), 32, 5, k32_5.data_ptr(), stream=stream), 32, 2)
            # the library's default dispatch for this batch (chi-square: harmonic-form nomination scan + exact re-rank; KL: entropy-form nomination + exact re-rank)
            q256 = q[:256].clamp_min(0.0)
            q256 = torch.where(q256 < 1e-4 / 13.0, torch.zeros_like(q256), q256).contiguous()
            k256 = torch.empty(256, device=dev, dtype=torch.int64)
            g.profile_enable(True)
            g.profile_read()
            r1d = rate(lambda: g.search_top1_keys_dev(q256.data_ptr(), 256, k256.data_ptr(), stream=stream), 256, 2)
            ms_nom, _ = g.profile_read()
            g.profile_enable(False)
            dsp_nom = g.last_dispatch()
            k256_5 = torch.empty((256, 5), device=dev, dtype=torch.int64)
            r5d = rate(lambda: g.search_topk_keys_dev(q256.data_ptr(), 256, 5, k256_5.data_ptr(), stream=stream), 256, 2)
            also[f"{name}_top1_default_dispatch_queries_per_s_batch256"] = r1d
            also[f"{name}_top1_default_dispatch_equals_exact_scan"] = bool(torch.equal(k256[:32], k32x))
            also[f"{name}_top5_default_dispatch_queries_per_s_batch256"] = r5d
            also[f"{name}_top5_default_dispatch_first_column_is_exact_top1"] = bool(torch.equal(k256_5[:32, 0], k32x))
            also[f"{name}_top1_queries_per_s"] = r1
            also[f"{name}_top5_queries_per_s"] = r5
            also[f"{name}_top5_first_column_is_top1"] = bool(torch.equal(k32_5[:, 0], k32))
            # each pass is launched as a (full, plain-range) pair of which one returns at once: per-pass time = sum over the pair
            per_pass_ms = float(np.mean(ms)) if len(ms) else float("nan")
            qpp = dsp["queries_per_pass"]
            passes = max(1, dsp["grid"][1])                          # query tiles folded into one launch (blockIdx.y)
            per_pass_ms /= passes
            elems_per_pass = float(n) * d * qpp
            wave_instr_per_s = elems_per_pass / 64.0 * slots / (per_pass_ms * 1e-3) if per_pass_ms == per_pass_ms else float("nan")
            # what the default dispatch runs for a 256-query batch: the NOMINATION scan (k_nominate, 16 queries per read of the gallery) + the exact re-rank
            # of the appended rows; its launch is timed by the library's HIP events (one launch = grid_y reads)
            nslots = NOM_SLOTS[name]
            reads = max(1, dsp_nom["grid"][1])
            ms_read = float(np.mean(ms_nom)) / reads if len(ms_nom) else float("nan")
            nom_rate = float(n) * d * dsp_nom["queries_per_pass"] / 64.0 * nslots / (ms_read * 1e-3) if ms_read == ms_read else float("nan")
            also[f"roofline_{name}"] = {"bound": "valu", "kernel": dsp_nom["kernel"], "what": "the nomination scan the default dispatch runs (cheap metric, threshold widened by its error bound, exact re-rank of the appended rows)",
                                       "model": NOM_MODEL[name] + "; peak = 256 CUs x 4 SIMDs x one wave64 instruction per 4 cycles at the 2.4 GHz maximum clock",
                                       "achieved": nom_rate / 1e9, "peak": PEAK_VALU_WAVE_INSTR_PER_S / 1e9, "unit": "G wave-instructions/s", "frac": nom_rate / PEAK_VALU_WAVE_INSTR_PER_S,
                                       "kernel_ms_per_launch": float(np.mean(ms_nom)) if len(ms_nom) else None, "gallery_reads_per_launch": reads, "kernel_ms_per_read": ms_read,
                                       "queries_per_read": dsp_nom["queries_per_pass"], "launches_timed": int(len(ms_nom)), "gallery_GBps": n * d * 4.0 / (ms_read * 1e-3) / 1e9,
                                       "frac_of_hbm_peak": n * d * 4.0 / (ms_read * 1e-3) / 1e9 / 8000.0, "vgprs": dsp_nom["vgprs"], "queries_per_s_whole_call": r1d,
                                       "sustained_clock_note": NOM_CLOCK_NOTE[name]}
            also[f"roofline_{name}_exact"] = {"bound": "valu", "model": f"{slots} VALU issue slots per (row, feature, query) element in the plain-range form (counted in the compiled loop, bench.py), "
                                                                    "peak = 256 CUs x 4 SIMDs x one wave64 instruction per 4 cycles at 2.4 GHz",
                                       "achieved": wave_instr_per_s / 1e9, "peak": PEAK_VALU_WAVE_INSTR_PER_S / 1e9, "unit": "G wave-instructions/s",
                                       "frac": wave_instr_per_s / PEAK_VALU_WAVE_INSTR_PER_S, "kernel": dsp["kernel"] + " (launched next to its plain-range twin; the one that applies runs)",
                                       "plain_range_form_ran": bool(plain[0] and plain[1]), "kernel_ms_per_pass": per_pass_ms, "queries_per_pass": qpp,
                                       "gallery_GBps": n * d * 4.0 / (per_pass_ms * 1e-3) / 1e9, "frac_of_hbm_peak": n * d * 4.0 / (per_pass_ms * 1e-3) / 1e9 / 8000.0,
                                       "vgprs": dsp["vgprs"], "lds_bytes_per_workgroup": dsp["lds_bytes"], "queries_per_s": r1}
        g.set_metric(fir.METRIC_L2)
    return also


def config2(fir, src, q, dev, ws, d):
    """BASELINE config 2: 100k x 512, batched L2 top-1 with the library's automatic tuning (cache-resident gallery)."""
    stream = ws.cuda_stream
    out = {}
    with torch.cuda.stream(ws):
        g = fir.Gallery(dev_ptr=src.data_ptr(), n=src.shape[0], d=d, metric=fir.METRIC_L2, device=dev.index, stream=stream)
        for qb in (256, min(q.shape[0], 4096)):
            k = torch.empty(qb, device=dev, dtype=torch.int64)
            g.set_large_batch_mfma(0)
            r_scan = rate(lambda: g.search_top1_keys_dev(q.data_ptr(), qb, k.data_ptr(), stream=stream), qb, 5)
            dsp = g.last_dispatch()
            ks = k.clone()
            g.set_large_batch_mfma(-1)
            r_def = rate(lambda: g.search_top1_keys_dev(q.data_ptr(), qb, k.data_ptr(), stream=stream), qb, 5)
            dd = g.last_dispatch()
            out[f"batch_{qb}"] = {"exact_scan_queries_per_s": r_scan, "exact_scan_kernel": dsp["kernel"], "exact_scan_queries_per_pass": dsp["queries_per_pass"],
                                  "default_dispatch_queries_per_s": r_def, "default_dispatch_path": dd["path"], "default_dispatch_kernel": dd["kernel"],
                                  "identical_keys": bool(torch.equal(ks, k))}
        # one 16-query pass of the exact scan from cold caches (right behind a 512 MiB write elsewhere) and warm, by the library's HIP events:
        # the 204.8 MB gallery fits the 256 MiB Infinity Cache, a warm pass is served from it
        flush = torch.empty(128 * 1024 * 1024, device=dev, dtype=torch.float32)
        k = torch.empty(16, device=dev, dtype=torch.int64)
        g.set_large_batch_mfma(0)
        g.search_top1_keys_dev(q.data_ptr(), 16, k.data_ptr(), stream=stream)
        g.profile_enable(True)
        cold, warm = [], []
        for rep in range(3):
            flush.fill_(float(rep))
            torch.cuda.synchronize()
            g.profile_read()
            g.search_top1_keys_dev(q.data_ptr(), 16, k.data_ptr(), stream=stream)
            torch.cuda.synchronize()
            cold.append(float(g.profile_read()[0].sum()))
            for _ in range(3):
                g.search_top1_keys_dev(q.data_ptr(), 16, k.data_ptr(), stream=stream)
            torch.cuda.synchronize()
            warm.append(float(np.median(g.profile_read()[0])))
        g.profile_enable(False)
        dsp = g.last_dispatch()
        gb = src.shape[0] * d * 4 / 1e9
        out["cold_vs_warm_pass_us"] = {"kernel": dsp["kernel"], "queries": 16, "cold_us": float(np.median(cold)) * 1e3, "warm_us": float(np.median(warm)) * 1e3,
                                       "cold_GBps": gb / (float(np.median(cold)) * 1e-3), "warm_GBps": gb / (float(np.median(warm)) * 1e-3),
                                       "note": "one gallery pass (204.8 MB); cold = first pass after a 512 MiB write to another buffer (rows come from HBM), warm = the passes after it (rows come from the Infinity Cache / L2)"}
        del flush
        g.close()
    # the proposed three-way-decision classifier (ImageTesting.cpp:207-288) on the same rows, one query per call like the reference's
    # recognize(): the one-launch form (default) against the launch-per-chunk forms, a query next to a gallery row (the loop ends after
    # the first 32-feature chunk) and a fresh one (all eight chunks); host pointers in, verdict out, median of 40 calls
    try:
        n = src.shape[0]
        cls_d = ((torch.arange(n, device=dev) // 30) % 1000).to(torch.int32)
        rows_h = src[:64].cpu().numpy()
        q_near = (rows_h[:1] * np.float32(0.9) + rows_h[32:33] * np.float32(0.1)).astype(np.float32)
        q_far = q[:1].cpu().numpy().astype(np.float32)
        with torch.cuda.stream(ws):
            gt = fir.Gallery(dev_ptr=src.data_ptr(), n=n, d=d, metric=fir.METRIC_L2, device=dev.index, stream=stream, dev_class_ptr=cls_d.data_ptr())
        torch.cuda.synchronize()
        twd = {}
        old_env = os.environ.get("FIR_TWD_FUSED")
        for mode, name in (("1", "one_launch"), ("0", "launch_per_chunk")):
            os.environ["FIR_TWD_FUSED"] = mode
            for qq, tag in ((q_near, "near_row"), (q_far, "fresh_query")):
                for _ in range(5):
                    res = gt.twd_proposed(qq, 32, 0.7)
                ts = []
                for _ in range(40):
                    t0 = time.perf_counter()
                    res = gt.twd_proposed(qq, 32, 0.7)
                    ts.append(time.perf_counter() - t0)
                twd[f"{name}_{tag}"] = {"us_per_call": float(np.median(ts)) * 1e6, "class": int(res[0][0]), "unreliable": int(res[1][0]), "chunks_used": int(res[2][0])}
        if old_env is None:
            os.environ.pop("FIR_TWD_FUSED", None)
        else:
            os.environ["FIR_TWD_FUSED"] = old_env
        twd["same_verdicts"] = all(twd[f"one_launch_{t}"][k] == twd[f"launch_per_chunk_{t}"][k] for t in ("near_row", "fresh_query") for k in ("class", "unreliable", "chunks_used"))
        out["twd_proposed_one_query"] = twd
        # ... and the conventional classifier (ImageTesting.cpp:108-186): posteriors and distance difference, first stage over 64 features
        conv = {}
        for mode, name in (("1", "one_launch"), ("0", "launch_per_stage")):
            os.environ["FIR_TWD_FUSED"] = mode
            for typ, th, tag in ((0, 0.24, "posteriors"), (1, 0.003, "difference")):
                for _ in range(5):
                    res = gt.twd_conventional(q_far, 1000, typ, th, 64)
                ts = []
                for _ in range(40):
                    t0 = time.perf_counter()
                    res = gt.twd_conventional(q_far, 1000, typ, th, 64)
                    ts.append(time.perf_counter() - t0)
                conv[f"{name}_{tag}"] = {"us_per_call": float(np.median(ts)) * 1e6, "class": int(res[0][0]), "unreliable": int(res[1][0])}
        if old_env is None:
            os.environ.pop("FIR_TWD_FUSED", None)
        else:
            os.environ["FIR_TWD_FUSED"] = old_env
        conv["same_verdicts"] = all(conv[f"one_launch_{t}"][k] == conv[f"launch_per_stage_{t}"][k] for t in ("posteriors", "difference") for k in ("class", "unreliable"))
        out["twd_conventional_one_query"] = conv
        gt.close()
    except Exception as e:      # noqa: BLE001 -- a side measurement must not take the bench line down
        out["twd_proposed_one_query"] = {"error": repr(e)}
    return out


def k3_classifiers(fir, dev, args, qb=64):
    """K3 (BASELINE.md section 3 row "GPU-1 chi2/KL/PNN, 1M x 512"): the float64 PNN / kNN classifiers
    (classification.cpp:116-226) over a 1M x 512 training set resident in HBM (4 GB of doubles)."""
    n, d, ncls = 1_000_000, 512, 1000
    g = torch.Generator(device=dev)
    g.manual_seed(31337)
    centres = torch.rand((ncls, d), generator=g, device=dev, dtype=torch.float64)
    tcls = (torch.arange(n, device=dev) * ncls // n).to(torch.int64)                      # class-major, 1000 rows per class
    tr = torch.empty((n, d), device=dev, dtype=torch.float64)
    for lo in range(0, n, 125_000):
        hi = lo + 125_000
        tr[lo:hi] = centres[tcls[lo:hi]] + 0.004 * torch.randn((hi - lo, d), generator=g, device=dev, dtype=torch.float64)
    avg = tr.mean(dim=0).cpu().numpy()
    pick = torch.randint(0, ncls, (qb,), generator=g, device=dev)
    q = (centres[pick] + 0.004 * torch.randn((qb, d), generator=g, device=dev, dtype=torch.float64)).cpu().numpy()
    torch.cuda.synchronize()
    m = fir.ClsModel(None, tcls.to(torch.int32).cpu().numpy(), ncls, avg, dev.index, dev_ptr=tr.data_ptr(), nt=n, d=d)
    del tr
    torch.cuda.empty_cache()
    out = {"workload": f"{n}x{d} float64 training rows ({n * d * 8 / 1e9:.1f} GB in HBM), {ncls} classes, {qb} queries per call (host pointers in, classes out)"}
    m.profile_enable(True)
    for name, fn in (("pnn_predict_bf", lambda: m.pnn_predict(q)), ("knn1_predict", lambda: m.knn_predict(q, 1))):
        res = fn()
        m.profile_read()
        t0 = time.perf_counter()
        for _ in range(3):
            res = fn()
        dt = (time.perf_counter() - t0) / 3
        ms, nbytes, kname = m.profile_read()
        cls = res[0] if isinstance(res, tuple) else res
        # the f64 scan over an HBM-streamed training set: tiles of 8 queries, TWO tiles per read of the rows since round 3 (k_cls_scan_lds<8, 2>;
        # the library's algorithmic byte count per launch says how many reads a call made)
        tiles8 = -(-qb // 8)
        out[name] = {"queries_per_s": qb / dt, "ms_per_call": dt * 1e3, "query_tiles_of_8": tiles8,
                     "class_of_the_planted_centre_found": float(np.mean(cls == pick.cpu().numpy()))}
        if len(ms):
            avg = float(np.mean(ms))
            reads = max(1, int(round(nbytes / (n * d * 8.0))))                                # reads of the 4.1 GB training set per launch
            q_per_read = 8 * tiles8 / reads
            # f64 issue model: per (row, feature, query) a subtraction, a multiplication and an addition, un-fused (classification.cpp:132-141):
            # 3 wave64 f64 instructions per 64 rows; peak = 256 CUs x 4 SIMDs x one wave64 instruction per 4 cycles at 2.4 GHz
            winst = (n / 64.0) * d * qb * 3.0
            out[name]["achieved_GBps"] = reads * n * d * 8.0 / dt / 1e9
            out[name]["frac_of_hbm_peak"] = reads * n * d * 8.0 / dt / 1e9 / 8000.0
            out[name]["roofline_k3"] = {"bound": "f64 vector issue" if q_per_read > 8 else "hbm", "kernel": kname, "kernel_avg_ms": avg, "launches_timed": int(len(ms)),
                                        "bytes_per_launch": nbytes, "training_set_reads_per_launch": reads, "queries_per_read": q_per_read,
                                        "hbm": {"achieved": nbytes / (avg * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": nbytes / (avg * 1e-3) / 1e9 / 8000.0},
                                        "f64_issue": {"achieved": winst / (avg * 1e-3) / 1e9, "peak": 614.4, "unit": "G wave-instructions/s",
                                                      "frac": winst / (avg * 1e-3) / 1e9 / 614.4},
                                        "achieved": nbytes / (avg * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": nbytes / (avg * 1e-3) / 1e9 / 8000.0,
                                        "kernel_time_share_of_call": float(np.sum(ms)) / (dt * 3 * 1e3)}
    # kNN batches through the matrix cores (csrc/fir_gemm_f64.h): fp16 fragments of the centred rows nominate, float64 re-ranks,
    # the certificate covers the rest, the vote runs over the nominated rows; what is not settled takes the exact scan. 4 096 queries per call.
    kq = 4096
    pickk = torch.randint(0, ncls, (kq,), generator=g, device=dev)
    qk = (centres[pickk] + 0.004 * torch.randn((kq, d), generator=g, device=dev, dtype=torch.float64)).cpu().numpy()
    m.set_knn_mfma(0)
    exact_cls = {kk: m.knn_predict(qk[:64], kk) for kk in (1, 3)}
    m.set_knn_mfma(-1)
    for kk in (1, 3):
        res = m.knn_predict(qk, kk)
        m.profile_read()
        s0 = m.knn_stats()
        t0 = time.perf_counter()
        for _ in range(3):
            res = m.knn_predict(qk, kk)
        dt = (time.perf_counter() - t0) / 3
        ms, _, _ = m.profile_read()
        s1 = m.knn_stats()
        dsp = m.last_dispatch()
        blk = {"queries_per_call": kq, "queries_per_s": kq / dt, "ms_per_call": dt * 1e3, "kernel": dsp["kernel"],
               "matrix_core_queries_per_call": (s1["matrix_core_queries"] - s0["matrix_core_queries"]) / 3.0,
               "exact_scan_queries_per_call": (s1["exact_scan_queries_of_them"] - s0["exact_scan_queries_of_them"]) / 3.0,
               "classes_equal_the_exact_scans_on_64": bool(np.array_equal(res[:64], exact_cls[kk])),
               "class_of_the_planted_centre_found": float(np.mean(res == pickk.cpu().numpy()))}
        if len(ms) and dsp["flops_per_launch"] > 0:
            kms = float(np.min(ms))                        # (the timed event pair brackets the FIRST full-pass launch of each call)
            blk["roofline_mfma"] = {"bound": "mfma", "kernel": dsp["kernel"], "kernel_ms_first_launch": kms, "flops_per_launch": dsp["flops_per_launch"],
                                    "achieved_tflops": dsp["flops_per_launch"] / (kms * 1e-3) / 1e12, "peak_tflops": PEAK_MFMA_F16_TFLOPS,
                                    "frac_of_mfma_peak": dsp["flops_per_launch"] / (kms * 1e-3) / 1e12 / PEAK_MFMA_F16_TFLOPS}
        out[f"knn{kk}_matrix_cores"] = blk
    m.close()
    return out


def config5(args, fir, dev, ws):
    """BASELINE configs[4] / BASELINE.md section 3 row GPU-GEMM: L2 top-1 over a 1M x 1280 gallery (EfficientNet-B7 width), the MFMA
    nomination + exact f32 re-rank + certificate path against the memory-bound exact scan, Qb in {8, 32, 256, 4096, 32768}:
    queries/s of both forms, what the default dispatch picks, the crossover batch, identical keys, uncertified (fallback) queries,
    and the roofline object of the dominant kernel at the largest batch (flops per launch / the library's HIP-event kernel time)."""
    n, d = args.rows, args.config5_dim
    stream = ws.cuda_stream
    rows = torch.empty((n, d), device=dev, dtype=torch.float32)
    for c in range((n + CHUNK_ROWS - 1) // CHUNK_ROWS):
        r0 = c * CHUNK_ROWS
        rows[r0:r0 + CHUNK_ROWS] = gen_chunk(c + 7000, min(CHUNK_ROWS, n - r0), d, dev)
    qmax = min(args.extra_batch, 32768)
    gq = torch.Generator(device=dev)
    gq.manual_seed(5151)
    fresh = torch.rand((qmax, d), generator=gq, device=dev)
    planted = (torch.arange(qmax, device=dev) * 977 + 11) % n
    pert = (rows[planted] + (torch.rand((qmax, d), generator=gq, device=dev) - 0.5) * 0.05 * rows[:4096].mean()).clamp_min(0)
    q = torch.where((torch.arange(qmax, device=dev) % 2 == 0)[:, None], fresh, pert)
    q = (q / q.norm(dim=1, keepdim=True)).contiguous()
    del fresh, pert
    out = {"workload": f"{n}x{d} f32 gallery (BASELINE configs[4]), L2 top-1, device-pointer calls", "batches": {}}
    with torch.cuda.stream(ws):
        g = fir.Gallery(dev_ptr=rows.data_ptr(), n=n, d=d, metric=fir.METRIC_L2, device=dev.index, stream=stream)
        del rows
        torch.cuda.empty_cache()
        crossover = None
        ident_all = True
        for qb in (8, 32, 256, 4096, qmax):
            k_scan = torch.empty(qb, device=dev, dtype=torch.int64)
            k_mfma = torch.empty(qb, device=dev, dtype=torch.int64)
            k_def = torch.empty(qb, device=dev, dtype=torch.int64)
            g.set_large_batch_mfma(0)
            r_scan = rate(lambda: g.search_top1_keys_dev(q.data_ptr(), qb, k_scan.data_ptr(), stream=stream), qb, 1 if qb > 4096 else 3)
            g.set_large_batch_mfma(1)                 # every batch through the matrix cores
            r_mfma = rate(lambda: g.search_top1_keys_dev(q.data_ptr(), qb, k_mfma.data_ptr(), stream=stream), qb, 5)
            g.set_large_batch_mfma(-1)                # the library's own choice
            st0 = g.mfma_stats()
            r_def = rate(lambda: g.search_top1_keys_dev(q.data_ptr(), qb, k_def.data_ptr(), stream=stream), qb, 5)
            st1 = g.mfma_stats()
            dd = g.last_dispatch()
            same = bool(torch.equal(k_scan, k_mfma)) and bool(torch.equal(k_scan, k_def))
            ident_all = ident_all and same
            if crossover is None and r_mfma > r_scan:
                crossover = qb
            out["batches"][str(qb)] = {"exact_scan_queries_per_s": r_scan, "matrix_core_queries_per_s": r_mfma, "default_dispatch_queries_per_s": r_def,
                                       "default_dispatch_path": dd["path"], "default_dispatch_kernel": dd["kernel"], "identical_keys": same,
                                       "default_over_best_of_the_two_forms": r_def / max(r_scan, r_mfma),
                                       # uncertified queries are dealt with on the device (csrc/fir_gemm_fb.h): a second matrix-core pass, then the exact scan
                                       "second_pass_queries_per_call": (st1["second_pass_queries"] - st0["second_pass_queries"]) / 6.0,
                                       "fallback_queries_per_call": (st1["fallback_queries"] - st0["fallback_queries"]) / 6.0}
        # the dominant kernel at the largest batch, timed by the library's HIP events
        qb = qmax
        k_def = torch.empty(qb, device=dev, dtype=torch.int64)
        g.profile_enable(True)
        g.profile_read()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            g.search_top1_keys_dev(q.data_ptr(), qb, k_def.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        ms, _ = g.profile_read()
        g.profile_enable(False)
        dd = g.last_dispatch()
        mem = g.memory_bytes()
        # diagnostics: the first (up to 8) queries of the last state's life whose FIRST certificate did not hold -- [list entries asked for (4096 fit),
        # the bound the pass appended below, the smallest stored proxy, |q|^2]; empty = every certificate held at the first pass
        out["first_uncertified_queries"] = g.uncertified_notes()
        g.close()
    if len(ms) and dd["path"] == "mfma":
        avg = float(np.mean(ms))
        tf = dd["flops_per_launch"] / (avg * 1e-3) / 1e12
        gbs = dd["bytes_per_launch"] / (avg * 1e-3) / 1e9
        out["roofline_mfma"] = {"bound": "mfma", "flops_per_launch": dd["flops_per_launch"], "achieved_tflops": tf, "peak_tflops": PEAK_MFMA_F16_TFLOPS,
                                "frac_of_mfma_peak": tf / PEAK_MFMA_F16_TFLOPS, "stream_GBps": gbs, "frac_of_hbm_peak": gbs / 8000.0, "kernel": dd["kernel"],
                                "kernel_avg_ms": avg, "launches_timed": int(len(ms)), "grid": dd["grid"], "block": dd["block"],
                                "lds_bytes_per_workgroup": dd["lds_bytes"], "vgprs": dd["vgprs"], "queries_per_gallery_read": dd["queries_per_pass"],
                                "bytes_per_launch": dd["bytes_per_launch"], "queries_per_s_whole_call": qb / dt,
                                "l2_to_cu": l2_to_cu(n, d, dd["queries_per_pass"], avg),
                                "kernel_time_share_of_call": float(np.sum(ms)) / (dt * reps * 1e3)}
    out["crossover_query_batch"] = crossover          # the smallest measured batch at which the matrix-core form beats the exact scan
    out["identical_keys_at_every_batch"] = ident_all
    out["hbm_bytes_held"] = mem
    torch.cuda.empty_cache()
    return out


def config4(args, build, make_queries, timed_loop, fir, dev, world):
    n4 = args.config4_rows
    steps = max(3, args.steps // 4)
    m, lo, hi, shard = build(n4)
    del shard
    torch.cuda.empty_cache()
    qb = min(args.batch, args.extra_batch)
    q, planted = make_queries(qb, n4)
    keys = torch.empty(qb, device=dev, dtype=torch.int64)
    el, k_ms, x_ms, disp = timed_loop(m, q, qb, keys, steps, 1)
    kd = keys.clone()
    sqb = min(args.scan_batch, qb)
    m.set_mfma(0)
    sk = torch.empty(sqb, device=dev, dtype=torch.int64)
    sel, s_ms, sx_ms, sdisp = timed_loop(m, q, sqb, sk, steps, 1)
    m.set_mfma(-1)
    idx, _ = fir.keys_unpack(kd.cpu().numpy().view(np.uint64))
    odd = np.arange(qb) % 2 == 1
    out = {"workload": f"{n4}x{args.dim} f32 gallery row-sharded over {world} rank(s): {hi - lo} rows on rank 0 (BASELINE configs[3])", "steps": steps,
           "query_batch": qb, "queries_per_s": qb * steps / el, "ms_per_step": el / steps * 1e3, "path": disp["path"], "kernel": disp["kernel"],
           "kernel_avg_ms": float(np.mean(k_ms)) if len(k_ms) else None,
           "stream_GBps_per_gpu": disp["bytes_per_launch"] / (float(np.mean(k_ms)) * 1e-3) / 1e9 if len(k_ms) else None,
           "exchange_us_per_step": float(np.mean(x_ms) * 1e3) if len(x_ms) else None,
           "exact_scan": {"query_batch": sqb, "queries_per_s": sqb * steps / sel, "ms_per_step": sel / steps * 1e3, "kernel": sdisp["kernel"],
                          "kernel_avg_ms": float(np.mean(s_ms)) if len(s_ms) else None,
                          "achieved_GBps_per_gpu": sdisp["bytes_per_launch"] / (float(np.mean(s_ms)) * 1e-3) / 1e9 if len(s_ms) else None,
                          "exchange_us_per_step": float(np.mean(sx_ms) * 1e3) if len(sx_ms) else None},
           "identical_keys_to_exact_scan": bool(torch.equal(sk, kd[:sqb])), "planted_queries_found": bool(np.all(idx[odd] == planted[odd]))}
    m.close()
    torch.cuda.empty_cache()
    return out


def pmc_config2_sequence(fir, dev, ws, d):
    """Counter pass only (BASELINE.md section 3 / SURVEY 8d: "100k x 512 fits Infinity Cache -> report rocprof DRAM bytes, cold vs
    warm"): a 100 000 x 512 gallery, 256 queries through the exact scan and 4 096 through the default dispatch, each once from cold
    caches -- right behind a 512 MiB write to another buffer -- and three times warm. The segments are delimited by marker
    launches the parent recognises by kernel name: a 512 MiB float64 fill (the flush) and an 8-element int16 fill."""
    n2 = 100_000
    src = torch.cat([gen_chunk(c + 9000, CHUNK_ROWS, d, dev) for c in range(-(-n2 // CHUNK_ROWS))])[:n2].contiguous()
    gq = torch.Generator(device=dev)
    gq.manual_seed(2222)
    q = torch.rand((4096, d), generator=gq, device=dev)
    q = (q / q.norm(dim=1, keepdim=True)).contiguous()
    big = torch.empty(64 * 1024 * 1024, device=dev, dtype=torch.float64)      # 512 MiB
    small = torch.empty(8, device=dev, dtype=torch.int16)
    k = torch.empty(4096, device=dev, dtype=torch.int64)
    stream = ws.cuda_stream
    with torch.cuda.stream(ws):
        g = fir.Gallery(dev_ptr=src.data_ptr(), n=n2, d=d, metric=fir.METRIC_L2, device=dev.index, stream=stream)
        for mode, qb in ((0, 256), (-1, 4096)):
            g.set_large_batch_mfma(mode)
            for _ in range(2):                       # set-up: code objects, scratch, the fp16 state
                g.search_top1_keys_dev(q.data_ptr(), qb, k.data_ptr(), stream=stream)
            torch.cuda.synchronize()
            big.fill_(1.0)                           # marker + flush
            torch.cuda.synchronize()
            g.search_top1_keys_dev(q.data_ptr(), qb, k.data_ptr(), stream=stream)      # cold
            torch.cuda.synchronize()
            small.fill_(1)                           # marker
            for _ in range(3):                       # warm
                g.search_top1_keys_dev(q.data_ptr(), qb, k.data_ptr(), stream=stream)
            torch.cuda.synchronize()
            small.fill_(2)                           # marker
            torch.cuda.synchronize()
        g.close()
    del big, src
    torch.cuda.empty_cache()


def pmc_traffic(args, n, d, world):
    """HBM bytes per launch of the two dominant kernels from the PMC counters: FETCH_SIZE (rocprofv3 unit: KiB) x 1024 x 2, as
    MI355X_MICROARCH.md prescribes for 16 B/lane streams on gfx950, averaged over the launches. Counters cannot be
    collected from inside the timed run, so after it this process starts `rocprofv3 --pmc FETCH_SIZE -- python3 bench.py
    --pmc-child ...` (its own pass with only that counter, a few steps of the same two loops) as a child process and
    reads its CSV. Without rocprofv3 (or if the pass fails) the scan pass recorded under profiles/ is reported for the shape
    it was taken on. Returns {"scan": bytes, "mfma": bytes, "how": str}."""
    import csv
    import glob
    import shutil
    import tempfile

    under_profiler = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", "")
    if world == 1 and not args.pmc_child and not args.no_pmc and not under_profiler and shutil.which("rocprofv3"):
        out_dir = tempfile.mkdtemp(prefix="fir_pmc_")
        cmd = ["rocprofv3", "--pmc", "FETCH_SIZE", "--output-format", "csv", "-d", out_dir, "-o", "pmc", "--",
               sys.executable, os.path.abspath(__file__), "--pmc-child", "--steps", "3", "--warmup", "1", "--rows", str(n), "--dim", str(d),
               "--batch", str(min(args.batch, 32768)), "--scan-batch", str(args.scan_batch), "--cpu-seconds", "0", "--no-extras"]
        if not args.no_extras and d == 512:
            cmd += ["--pmc-config2"]
        if args.qpp:
            cmd += ["--qpp", str(args.qpp)]
        try:
            subprocess.run(cmd, cwd=out_dir, env=dict(os.environ, TMPDIR=out_dir), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                           timeout=200, check=True)
            vals = {"scan": [], "mfma": []}
            seg, seg_bytes = -1, {}                  # config-2 sequence: segment number (bumped by every marker launch) -> FETCH_SIZE of the library's kernels in it
            for f in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
                rows_csv = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE"]
                rows_csv.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
                is_marker = lambda kn: "FillFunctor<double>" in kn or "FillFunctor<short>" in kn
                has_cfg2 = sum(1 for r in rows_csv if is_marker(r["Kernel_Name"])) >= 6      # the config-2 sequence ran first (six marker launches)
                for r in rows_csv:
                    kn = r["Kernel_Name"]
                    if is_marker(kn):
                        seg += 1
                        continue
                    mine = "fir::" in kn or "(anonymous namespace)::k_" in kn
                    if has_cfg2 and seg <= 4:        # segments 0 / 1: scan cold / warm, 2: matrix-core set-up, 3 / 4: default dispatch cold / warm (-1: the scan's set-up)
                        if mine and seg >= 0:
                            seg_bytes[seg] = seg_bytes.get(seg, 0.0) + float(r["Counter_Value"]) * 1024 * 2
                        continue                     # (the config-2 launches are not the headline's)
                    if "k_scan_l2" in kn:
                        vals["scan"].append(float(r["Counter_Value"]))
                    elif "k_gemm_proxy_f16" in kn and ("<1," in kn or "<3," in kn or "false>" in kn):
                        vals["mfma"].append(float(r["Counter_Value"]))
            cfg2 = None
            if all(i in seg_bytes for i in (0, 1, 3, 4)):
                alg = 100_000 * 512 * 4.0
                cfg2 = {"what": "bytes requested beyond L2 (PMC FETCH_SIZE x 1024 x 2, all of the library's kernels of the call) per call over a 100 000 x 512 gallery: cold = "
                                "first call after a 512 MiB write to another buffer, warm = mean of the three calls after it; the gallery is 204.8 MB of f32 rows "
                                "(+ 102.4 MB of fp16 fragments for the default dispatch) against 256 MiB of Infinity Cache. FETCH_SIZE is taken at the L2's "
                                "memory-side interface and counts Infinity-Cache hits like HBM reads (MI355X_MICROARCH.md, HBM): cold and warm read the same here, and "
                                "rocprofv3 -L offers no counter behind the Infinity Cache on this image -- what separates them is the pass time, `cold_vs_warm_pass_us`",
                        "exact_scan_256_queries": {"cold": seg_bytes[0], "warm": seg_bytes[1] / 3.0, "algorithmic_gallery_bytes": alg,
                                                   "gallery_passes_per_call": 16},
                        "default_dispatch_4096_queries": {"cold": seg_bytes[3], "warm": seg_bytes[4] / 3.0, "algorithmic_fragment_bytes": alg / 2}}
            if vals["scan"] or vals["mfma"]:
                res = {k: (sum(v) / len(v) * 1024 * 2 if v else None) for k, v in vals.items()}
                res["how"] = f"rocprofv3 --pmc FETCH_SIZE child pass of this run ({len(vals['scan'])} scan, {len(vals['mfma'])} matrix-core launches), KiB x 1024 x 2"
                res["config2"] = cfg2
                return res
        except Exception:
            pass
        finally:
            keep = os.environ.get("FIR_BENCH_KEEP_PMC")          # debugging: a directory that receives the counter pass's CSV files
            if keep:
                for f in glob.glob(os.path.join(out_dir, "**", "*.csv"), recursive=True):
                    shutil.copy(f, os.path.join(keep, os.path.basename(f)))
            shutil.rmtree(out_dir, ignore_errors=True)
    path = os.path.join(ROOT, "profiles", "r01_rocprofv3_pmc_fetch_size.json")
    if os.path.exists(path) and (n, d, world, args.scan_batch) == (1_000_000, 512, 1, 256):
        for e in json.load(open(path)):
            if "k_scan_l2" in e["kernel"] and e["counter"] == "FETCH_SIZE":
                return {"scan": e["bytes_per_launch_corrected"], "mfma": None, "how": "recorded pass profiles/r01_rocprofv3_pmc_fetch_size.json"}
    return {}


def cpu_baselines(rows, queries, gpu_idx, gpu_dist, budget_s):
    """Both CPU rows of BASELINE.md section 3 on ALL host cores of this box (count and model stated):
    `reference`: the reference's own recognize_image_bf (qt_cpp/db_features.cpp:319-335, oracle/_ref), one query per thread at a
    time -- the reference itself is single threaded, queries are independent (falls back to the C restatement when
    oracle/_ref is not present);  `all`: the C restatement with OpenMP over queries (bit-identical arithmetic)."""
    import oracle_lib

    cores = host_cores()
    mask = len(os.sched_getaffinity(0))
    model = cpu_model() + f" ({mask} hardware threads in the affinity mask, {cores} usable under the cgroup CPU quota)"
    d = rows.shape[1]
    kind = "reference" if oracle_lib.have_ref() else "port"
    orc = oracle_lib.load_oracle()
    if kind == "reference":
        db = oracle_lib.load_ref("l2").db(rows, None, 0)
        fn = lambda qv: db.recognize_image_bf(qv, d)  # noqa: E731
    else:
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
    ref = {"value": nq / dt, "unit": "queries/s", "cores": cores, "cpu_model": model, "kind": kind,
           "sample": f"{nq} of the step's {queries.shape[0]} queries against the full {rows.shape[0]}x{d} gallery, "
                     f"{cores} threads x recognize_image_bf ({t_one:.2f} s/query/thread); GPU index identical on {agree}/{nq}"}
    # CPU-all: OpenMP over queries, same sample size
    t0 = time.perf_counter()
    oi, od, threads = orc.top1_batch_omp(rows, queries[:nq], 0, d, 0, threads=cores)
    dt2 = time.perf_counter() - t0
    same = int(np.sum((oi == gpu_idx[:nq]) & (od.view(np.uint32) == np.ascontiguousarray(gpu_dist[:nq], np.float32).view(np.uint32))))
    allc = {"value": nq / dt2, "unit": "queries/s", "cores": threads, "cpu_model": model, "kind": "port",
            "effective_GBps": nq / dt2 * rows.shape[0] * d * 4.0 / 1e9,
            "sample": f"the same {nq} queries, oracle/oracle.c restatement with OpenMP over queries ({threads} threads); "
                      f"GPU index AND distance bits identical on {same}/{nq}"}
    return {"reference": ref, "all": allc}


if __name__ == "__main__":
    main()
