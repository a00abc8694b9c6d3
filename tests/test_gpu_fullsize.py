"""Full-size runs on the GPU (BASELINE.json's 1M x 512 and a 9M x 512 gallery whose element count passes 2^32), checked
through size-independent properties: planted rows are found with distance exactly 0, the scan, its row-sharded
form, the top-K scan and the matrix-core path agree with each other bit for bit, and a strided sample of rows is
checked against the oracle. Galleries are generated on the device (torch is plumbing here)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def make_gallery(n, d, seed):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    x = torch.empty((n, d), device="cuda", dtype=torch.float32)
    step = 1_000_000
    for lo in range(0, n, step):                      # bounded temporaries
        hi = min(n, lo + step)
        c = torch.rand((hi - lo, d), generator=g, device="cuda")
        c = torch.where(c < 1e-4, torch.zeros_like(c), c)      # db_features.cpp:85-86
        x[lo:hi] = c / c.norm(dim=1, keepdim=True)
    return x


def keys_of(fir, g, q, stream):
    keys = torch.empty(q.shape[0], device="cuda", dtype=torch.int64)
    g.search_top1_keys_dev(q.data_ptr(), q.shape[0], keys.data_ptr(), stream=stream.cuda_stream)
    stream.synchronize()
    return keys.cpu().numpy().view(np.uint64)


@pytest.mark.parametrize("n", [1_000_000, 9_000_000])
def test_planted_rows_shards_topk_and_mfma_agree(fir, oracle, n):
    d = 512
    assert n < 2 ** 31 and (n == 1_000_000 or n * d > 2 ** 32)
    x = make_gallery(n, d, 1234 + n)
    planted = np.array([0, 63, 64, n // 3, n // 2 + 1, n - 65, n - 2, n - 1], dtype=np.int64)
    st = torch.cuda.Stream()
    q_exact = x[torch.from_numpy(planted).cuda()].clone()                  # exact copies: distance 0.0
    gq = torch.Generator(device="cuda")
    gq.manual_seed(99)
    noise = (torch.rand((8, d), generator=gq, device="cuda") - 0.5) * 0.02 * x[:1000].mean()
    q_near = (q_exact + noise).clamp_min(0)
    q_near = q_near / q_near.norm(dim=1, keepdim=True)
    q = torch.cat([q_exact, q_near]).contiguous()
    sample = np.unique(np.concatenate([np.arange(0, n, max(n // 499, 1)), planted]))       # rows also checked on the CPU
    host_sample = x[torch.from_numpy(sample).cuda()].cpu().numpy()
    host_q = q.cpu().numpy()

    with torch.cuda.stream(st):
        g = fir.Gallery(dev_ptr=x.data_ptr(), n=n, d=d, metric=0, device=0, stream=st.cuda_stream)
        whole = keys_of(fir, g, q, st)
        idx, dist = fir.keys_unpack(whole)
        assert np.array_equal(idx[:8], planted) and np.all(dist[:8].view(np.uint32) == 0)      # +0.0 exactly
        assert np.array_equal(idx[8:], planted)                                              # 2 % noise cannot move the neighbour
        # the reported distances are the oracle's, to the bit (the planted rows are part of the CPU sample)
        for i in range(8, 16):
            exp = oracle.feature_distance(host_q[i], host_sample[np.searchsorted(sample, planted[i - 8])], 0, d, 0)
            assert np.float32(dist[i]).view(np.uint32) == np.float32(exp).view(np.uint32)
        # all distances of two queries (store epilogue): spot-checked against the oracle on the sampled rows
        out = torch.empty((2, n), device="cuda", dtype=torch.float32)
        g.range_distances_dev(q[8:10].contiguous().data_ptr(), 2, out.data_ptr(), 0, 0, stream=st.cuda_stream)
        st.synchronize()
        got = out[:, torch.from_numpy(sample).cuda()].cpu().numpy()
        for r in range(0, len(sample), 25):
            for qi in range(2):
                exp = oracle.feature_distance(host_q[8 + qi], host_sample[r], 0, d, 0)
                assert got[qi, r].view(np.uint32) == np.float32(exp).view(np.uint32), (qi, sample[r])
        assert float(out[0].min()) == dist[8]                                                # the minimum IS the reported best
        del out
        # top-5: first entry = top-1, ascending, and consistent with the shard merge below
        k5 = torch.empty((16, 5), device="cuda", dtype=torch.int64)
        g.search_topk_keys_dev(q.data_ptr(), 16, 5, k5.data_ptr(), stream=st.cuda_stream)
        st.synchronize()
        k5 = k5.cpu().numpy().view(np.uint64)
        assert np.array_equal(k5[:, 0], whole) and np.all(k5[:, 1:] > k5[:, :-1])
        # the matrix-core path returns the same keys
        gm = fir.GemmSearch(g)
        k2 = torch.empty(16, device="cuda", dtype=torch.int64)
        gm.search_top1_keys_dev(q.data_ptr(), 16, k2.data_ptr(), stream=st.cuda_stream)
        st.synchronize()
        assert np.array_equal(k2.cpu().numpy().view(np.uint64), whole)
        # ... and the same top-5 keys as the exact top-K scan, 256 queries (rows of the gallery, perturbed)
        qk = x[torch.arange(256, device="cuda") * (n // 256)] * 0.98 + x[:256] * 0.02
        qk = (qk / qk.norm(dim=1, keepdim=True)).contiguous()
        km = torch.empty((256, 5), device="cuda", dtype=torch.int64)
        ke = torch.empty((256, 5), device="cuda", dtype=torch.int64)
        gm.search_topk_keys_dev(qk.data_ptr(), 256, 5, km.data_ptr(), stream=st.cuda_stream)
        g.set_large_batch_mfma(0)
        g.search_topk_keys_dev(qk.data_ptr(), 256, 5, ke.data_ptr(), stream=st.cuda_stream)
        g.set_large_batch_mfma(-1)
        st.synchronize()
        assert torch.equal(km, ke)
        assert gm.stats()["fallback_queries"] <= 64
        del qk, km, ke
        gm.close()
        g.close()
        # two row shards with offsets: the minimum of their keys is the whole gallery's answer
        cut = (n // 2 // 64) * 64 + 64
        parts = []
        for lo, hi in ((0, cut), (cut, n)):
            gs = fir.Gallery(dev_ptr=x[lo:hi].data_ptr(), n=hi - lo, d=d, metric=0, device=0, stream=st.cuda_stream)
            gs.set_row_offset(lo)
            parts.append(keys_of(fir, gs, q, st))
            gs.close()
        assert np.array_equal(np.minimum(parts[0], parts[1]), whole)
    del x
    torch.cuda.empty_cache()


@pytest.mark.parametrize("metric", [1, 2])
def test_chi2_kl_top5_at_1m(fir, oracle, metric):
    """BASELINE config 3: 1M x 512, chi-square / KL, top-K = 5. Exact copies are at distance 0 and come first; the
    top-5 lists are ascending, start with the top-1 answer and equal the merge of two row shards' lists; distances are
    the oracle's (bit for bit for chi-square, 1e-5 relative for KL whose log is the device's)."""
    from fast_image_recognition_amd import sharding

    n, d, k = 1_000_000, 512, 5
    x = make_gallery(n, d, 777)
    x = x * x.norm(dim=1, keepdim=True)
    x = (x / x.sum(dim=1, keepdim=True)).contiguous()             # the chi2 / KL builds normalise by the sum (db_features.cpp:91)
    planted = np.array([1, n // 2, n - 1], dtype=np.int64)
    q = x[torch.from_numpy(planted).cuda()].clone()
    gq = torch.Generator(device="cuda")
    gq.manual_seed(5)
    fresh = torch.rand((5, d), generator=gq, device="cuda")
    q = torch.cat([q, fresh / fresh.sum(dim=1, keepdim=True)]).contiguous()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        g = fir.Gallery(dev_ptr=x.data_ptr(), n=n, d=d, metric=metric, device=0, stream=st.cuda_stream)
        top1 = keys_of(fir, g, q, st)
        k5 = torch.empty((8, k), device="cuda", dtype=torch.int64)
        g.search_topk_keys_dev(q.data_ptr(), 8, k, k5.data_ptr(), stream=st.cuda_stream)
        st.synchronize()
        k5 = k5.cpu().numpy().view(np.uint64)
        g.close()
        idx, dist = fir.keys_unpack(k5.reshape(-1))
        idx, dist = idx.reshape(8, k), dist.reshape(8, k)
        assert np.array_equal(k5[:, 0], top1) and np.all(k5[:, 1:] > k5[:, :-1])
        assert np.array_equal(idx[:3, 0], planted) and np.all(dist[:3, 0] == 0)
        rows_needed = torch.from_numpy(idx.reshape(-1).astype(np.int64)).cuda()
        host_rows = x[rows_needed].cpu().numpy().reshape(8, k, d)
        host_q = q.cpu().numpy()
        for i in range(8):
            for j in range(k):
                exp = oracle.feature_distance(host_q[i], host_rows[i, j], 0, d, metric)
                if metric == 1:
                    assert dist[i, j].view(np.uint32) == np.float32(exp).view(np.uint32), (i, j)
                else:
                    assert abs(float(dist[i, j]) - float(exp)) <= 1e-5 * abs(float(exp)) + 1e-12, (i, j)
        cut = 500_032
        parts = []
        for lo, hi in ((0, cut), (cut, n)):
            gs = fir.Gallery(dev_ptr=x[lo:hi].data_ptr(), n=hi - lo, d=d, metric=metric, device=0, stream=st.cuda_stream)
            gs.set_row_offset(lo)
            kk = torch.empty((8, k), device="cuda", dtype=torch.int64)
            gs.search_topk_keys_dev(q.data_ptr(), 8, k, kk.data_ptr(), stream=st.cuda_stream)
            st.synchronize()
            parts.append(sharding.keys_as_int64(kk.cpu()))
            gs.close()
        merged = sharding.keys_from_int64(sharding.merge_topk_keys(torch.stack(parts), k)).numpy().view(np.uint64)
        assert np.array_equal(merged, k5)
    del x
    torch.cuda.empty_cache()


def test_config2_100k_x_512_automatic_tuning_against_the_oracle(fir, oracle):
    """BASELINE config 2: 100k x 512, batched L2 top-1 with the library's own choices -- the cache-resident gallery takes 16
    queries per pass in the exact scan, and a 256-query batch takes the matrix cores by default. Both must return the
    oracle's index and distance bits (24 queries checked outright on the CPU, ~0.08 s each) and each other's keys."""
    import synth

    n, d, qb = 100_000, 512, 256
    rows = synth.make_gallery(53, n, d, 0)
    q, _ = synth.make_queries(53, rows, qb, 0)
    rows[n - 7] = rows[12]
    q[5] = rows[12]                                   # exact tie, first row wins
    with fir.Gallery(rows, None, 0, 0) as g:
        a_idx, a_dist = g.search_top1(q)              # default dispatch
        da = g.last_dispatch()
        g.set_large_batch_mfma(0)
        b_idx, b_dist = g.search_top1(q)              # exact scan, automatic queries-per-pass
        db = g.last_dispatch()
        tun = g.get_tuning()
        c_idx, c_dist = g.search_top1(q[:24])         # a small batch: the scan at whatever tile the library picks
    assert da["path"] == "mfma" and db["path"] == "scan"
    assert tun["queries_per_pass"] == 16 and db["queries_per_pass"] == 16 and "k_scan_l2_lds<2" in db["kernel"]     # cache-resident: 16 per pass
    assert np.array_equal(a_idx, b_idx) and np.array_equal(a_dist.view(np.uint32), b_dist.view(np.uint32))
    eidx, edist = oracle.top1_batch(rows, q[:24], 0, d, 0)
    assert np.array_equal(a_idx[:24], eidx) and np.array_equal(a_dist[:24].view(np.uint32), edist.view(np.uint32))
    assert np.array_equal(c_idx, eidx) and np.array_equal(c_dist.view(np.uint32), edist.view(np.uint32))
    assert a_idx[5] == 12 and a_dist[5] == 0.0


@pytest.mark.parametrize("precision", [2, 1])         # FIR_GEMM_F16, FIR_GEMM_BF16_SPLIT
def test_config5_1m_x_1280_matrix_core_path_equals_the_scan(fir, oracle, precision):
    """BASELINE config 5: 1M x 1280 (EfficientNet-B7 width). The MFMA nomination + exact re-rank + certificate path must
    return the exact scan's keys bit for bit on a 512-query batch; planted exact copies are found at distance +0.0;
    the winners' distances are the oracle's bits on the rows fetched back."""
    n, d, qb = 1_000_000, 1280, 512
    x = make_gallery(n, d, 4242)
    planted = np.array([0, 31, 32, n // 2 + 17, n - 33, n - 1], dtype=np.int64)
    gq = torch.Generator(device="cuda")
    gq.manual_seed(7)
    fresh = torch.rand((qb, d), generator=gq, device="cuda")
    q = fresh / fresh.norm(dim=1, keepdim=True)
    q[: len(planted)] = x[torch.from_numpy(planted).cuda()]
    noise = (torch.rand((64, d), generator=gq, device="cuda") - 0.5) * 0.02 * x[:1000].mean()
    near = (x[torch.arange(64, device="cuda") * 15013 + 5] + noise).clamp_min(0)
    q[64:128] = near / near.norm(dim=1, keepdim=True)
    q = q.contiguous()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        g = fir.Gallery(dev_ptr=x.data_ptr(), n=n, d=d, metric=0, device=0, stream=st.cuda_stream)
        g.set_large_batch_mfma(0)
        scan = keys_of(fir, g, q, st)
        assert g.last_dispatch()["path"] == "scan"
        gm = fir.GemmSearch(g, precision)
        k2 = torch.empty(qb, device="cuda", dtype=torch.int64)
        gm.search_top1_keys_dev(q.data_ptr(), qb, k2.data_ptr(), stream=st.cuda_stream)
        st.synchronize()
        got = k2.cpu().numpy().view(np.uint64)
        fb = gm.stats()["fallback_queries"]
        gm.close()
        small = {}
        if precision == 2:                              # and the default dispatch is this path
            g.set_large_batch_mfma(-1)
            auto = keys_of(fir, g, q, st)
            assert g.last_dispatch()["path"] == "mfma" and np.array_equal(auto, scan)
            # small batches of config 5 on benign data -- 8 and 32 queries, ten calls each through the default
            # dispatch: every certificate holds at the first pass (no second pass, no exact scan) and the keys are the scan's
            for sq in (8, 32):
                ks = torch.empty(sq, device="cuda", dtype=torch.int64)
                s0 = g.mfma_stats()
                for _ in range(10):
                    g.search_top1_keys_dev(q[128:].data_ptr(), sq, ks.data_ptr(), stream=st.cuda_stream)
                    st.synchronize()
                    assert np.array_equal(ks.cpu().numpy().view(np.uint64), scan[128:128 + sq])
                s1 = g.mfma_stats()
                small[sq] = (g.last_dispatch()["path"], s1["second_pass_queries"] - s0["second_pass_queries"], s1["fallback_queries"] - s0["fallback_queries"])
        g.close()
    for sq, (path, second, exact) in small.items():
        assert path == "mfma" and second == 0 and exact == 0, (sq, path, second, exact)
    assert np.array_equal(got, scan)
    assert fb <= qb // 8, fb                           # the certificate holds for (nearly) every query: the path is not the scan in disguise
    idx, dist = fir.keys_unpack(scan)
    assert np.array_equal(idx[: len(planted)], planted) and np.all(dist[: len(planted)].view(np.uint32) == 0)
    assert np.array_equal(idx[64:128], np.arange(64) * 15013 + 5)
    pick = np.arange(0, qb, 13)
    host_rows = x[torch.from_numpy(idx[pick].astype(np.int64)).cuda()].cpu().numpy()
    host_q = q[torch.from_numpy(pick).cuda()].cpu().numpy()
    for j, i in enumerate(pick):
        exp = oracle.feature_distance(host_q[j], host_rows[j], 0, d, 0)
        assert np.float32(dist[i]).view(np.uint32) == np.float32(exp).view(np.uint32), i
    del x
    torch.cuda.empty_cache()


def test_config4_10m_x_512_as_eight_shards_through_the_library_exchange(fir, oracle):
    """BASELINE config 4: 10M x 512 row-sharded 8 ways. On one GPU the eight shards are logical (20.5 GB of rows fit), but
    the split, the per-shard scans, the on-device minimum and the RCCL call are the library's (fir_sharded_*). The reduced
    keys must equal the scan of the whole gallery as ONE handle -- for the default dispatch (matrix cores in every shard)
    and for the exact scan -- and planted rows sit at +0.0, the last one in the last shard."""
    n, d, qb = 10_000_000, 512, 256
    x = make_gallery(n, d, 99)
    planted = np.array([0, 1_249_999, 1_250_048, 5_000_000, 8_750_016, n - 1], dtype=np.int64)
    gq = torch.Generator(device="cuda")
    gq.manual_seed(11)
    fresh = torch.rand((qb, d), generator=gq, device="cuda")
    q = fresh / fresh.norm(dim=1, keepdim=True)
    q[: len(planted)] = x[torch.from_numpy(planted).cuda()]
    x[n - 5] = x[3]
    q[10] = x[3]                                        # a tie between the first and the last shard: the first row wins
    q = q.contiguous()
    st = torch.cuda.Stream()
    keys = torch.empty(qb, device="cuda", dtype=torch.int64)
    with torch.cuda.stream(st):
        s = fir.ShardedGallery(dev_ptr=x.data_ptr(), n=n, d=d, metric=0, devices=[0], shards_per_device=8)
        bounds = [s.shard(i)[1:] for i in range(8)]
        assert [b[1] for b in bounds] == [1_250_048] * 7 + [n - 7 * 1_250_048]      # whole 64-row tiles per shard
        s.search_top1_keys_dev(q.data_ptr(), qb, keys.data_ptr(), stream=st.cuda_stream)
        st.synchronize()
        sharded_default = keys.cpu().numpy().view(np.uint64).copy()
        assert s.shard(0)[0].last_dispatch()["path"] == "mfma"
        for i in range(8):
            s.shard(i)[0].set_large_batch_mfma(0)
        s.search_top1_keys_dev(q.data_ptr(), qb, keys.data_ptr(), stream=st.cuda_stream)
        st.synchronize()
        sharded_scan = keys.cpu().numpy().view(np.uint64).copy()
        assert s.shard(7)[0].last_dispatch()["path"] == "scan"
        s.close()
        g = fir.Gallery(dev_ptr=x.data_ptr(), n=n, d=d, metric=0, device=0, stream=st.cuda_stream)
        g.set_large_batch_mfma(0)
        whole = keys_of(fir, g, q, st)
        g.close()
    assert np.array_equal(sharded_scan, whole) and np.array_equal(sharded_default, whole)
    idx, dist = fir.keys_unpack(whole)
    assert np.array_equal(idx[: len(planted)], planted) and np.all(dist[: len(planted)].view(np.uint32) == 0)
    assert idx[10] == 3 and dist[10] == 0.0
    pick = np.arange(16, qb, 31)
    host_rows = x[torch.from_numpy(idx[pick].astype(np.int64)).cuda()].cpu().numpy()
    host_q = q[torch.from_numpy(pick).cuda()].cpu().numpy()
    for j, i in enumerate(pick):
        exp = oracle.feature_distance(host_q[j], host_rows[j], 0, d, 0)
        assert np.float32(dist[i]).view(np.uint32) == np.float32(exp).view(np.uint32), i
    del x
    torch.cuda.empty_cache()


@pytest.mark.parametrize("qb", [8, 4096])
def test_power_of_two_scaling_and_row_reversal_at_1m(fir, qb):
    """Two size-independent properties of the whole 1M x 512 path (matrix-core nomination, exact re-rank, certificate), no oracle needed:
    (1) gallery and queries both multiplied by 2^e: every squared difference scales by 4^e exactly, so the same rows come back with the
    distance's exponent shifted by 2e and the same mantissa bits -- for e = 5 and e = -7 (the fp16 fragments are cut with power-of-two
    scales per gallery and per query: this walks their exponent arithmetic); (2) the gallery's rows in reverse order: row i becomes row
    n - 1 - i with the same distance bits (random rows: no ties)."""
    n, d = 1_000_000, 512
    x = make_gallery(n, d, 4242)
    gq = torch.Generator(device="cuda")
    gq.manual_seed(99)
    q = torch.rand((qb, d), generator=gq, device="cuda")
    q[::2] = x[(torch.arange(qb, device="cuda")[::2] * 977 + 5) % n] * 0.98 + q[::2] * 0.02
    q = (q / q.norm(dim=1, keepdim=True)).contiguous()
    st = torch.cuda.Stream()
    torch.cuda.synchronize()

    def run(rows, queries):
        with fir.Gallery(dev_ptr=rows.data_ptr(), n=n, d=d, metric=0, device=0, stream=st.cuda_stream) as g:
            keys = keys_of(fir, g, queries, st)
            assert g.last_dispatch()["path"] == "mfma"
        return fir.keys_unpack(keys)

    i0, d0 = run(x, q)
    for e in (5, -7):
        xs, qs = (x * 2.0 ** e).contiguous(), (q * 2.0 ** e).contiguous()
        torch.cuda.synchronize()
        ie, de = run(xs, qs)
        assert np.array_equal(ie, i0)
        assert np.array_equal(de.view(np.uint32), (d0 * np.float32(4.0 ** e)).view(np.uint32))
        del xs, qs
    xr = torch.flip(x, dims=[0]).contiguous()
    torch.cuda.synchronize()
    ir, dr = run(xr, q)
    assert np.array_equal(ir, n - 1 - i0) and np.array_equal(dr.view(np.uint32), d0.view(np.uint32))


@pytest.mark.parametrize("metric", [1, 2])
def test_chi2_kl_scale_with_their_operands_at_1m(fir, metric):
    """chi-square and KL are homogeneous of degree one: gallery and queries both multiplied by 2^e give the same rows (top-1 and top-5)
    and every distance times 2^e exactly -- (l - r)^2 / (l + r) and l log(2l / s) pick the factor up once, the quotients inside do not
    see it. The nomination forms (harmonic sums / entropies, thresholds widened by bounds relative to sum(l) + max sum(r)) are NOT
    scale-free in what they add up, so this walks their widening at other magnitudes; 40 queries take the nomination scan."""
    n, d, k, qb = 1_000_000, 512, 5, 40
    x = make_gallery(n, d, 778)
    x = x * x.norm(dim=1, keepdim=True)
    x = (x / x.sum(dim=1, keepdim=True)).contiguous()
    gq = torch.Generator(device="cuda")
    gq.manual_seed(6)
    fresh = torch.rand((qb, d), generator=gq, device="cuda")
    fresh = torch.where(fresh < 1e-4, torch.zeros_like(fresh), fresh)
    q = fresh / fresh.sum(dim=1, keepdim=True)
    q[::2] = x[(torch.arange(qb, device="cuda")[::2] * 7919 + 3) % n] * 0.97 + q[::2] * 0.03
    q = q.contiguous()
    st = torch.cuda.Stream()
    torch.cuda.synchronize()

    def run(rows, queries):
        with fir.Gallery(dev_ptr=rows.data_ptr(), n=n, d=d, metric=metric, device=0, stream=st.cuda_stream) as g:
            k1 = keys_of(fir, g, queries, st)
            assert "k_nominate" in g.last_dispatch()["kernel"]
            kk = torch.empty((qb, k), device="cuda", dtype=torch.int64)
            g.search_topk_keys_dev(queries.data_ptr(), qb, k, kk.data_ptr(), stream=st.cuda_stream)
            st.synchronize()
        i1, d1 = fir.keys_unpack(k1)
        i5, d5 = fir.keys_unpack(kk.cpu().numpy().view(np.uint64).reshape(-1))
        return i1, d1, i5, d5

    base = run(x, q)
    assert np.array_equal(base[2].reshape(qb, k)[:, 0], base[0])
    for e in (3, -3):
        xs, qs = (x * 2.0 ** e).contiguous(), (q * 2.0 ** e).contiguous()
        torch.cuda.synchronize()
        got = run(xs, qs)
        assert np.array_equal(got[0], base[0]) and np.array_equal(got[2], base[2])
        assert np.array_equal(got[1].view(np.uint32), (base[1] * np.float32(2.0 ** e)).view(np.uint32))
        assert np.array_equal(got[3].view(np.uint32), (base[3] * np.float32(2.0 ** e)).view(np.uint32))
        del xs, qs
    del x
    torch.cuda.empty_cache()
