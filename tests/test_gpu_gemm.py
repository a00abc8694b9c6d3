"""The matrix-core path (fir_gemm_*) must return exactly what the exact scan returns -- index and
distance bit-for-bit -- on easy data, on adversarial data (many near ties: more than the re-ranked
candidates), with duplicates, NaN rows and odd shapes; the certificate / fallback makes it so."""
import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu


def run_both(fir, rows, q, precision=2):
    dev = torch.device("cuda", 0)
    with fir.Gallery(rows, None, 0, 0) as g:
        g.set_large_batch_mfma(0)            # the reference answer is the exact scan's, whatever the batch size
        eidx, edist = g.search_top1(q)
        assert g.last_dispatch()["path"] == "scan"
        tq = torch.from_numpy(np.ascontiguousarray(q, np.float32)).to(dev)
        keys = torch.empty(q.shape[0], dtype=torch.int64, device=dev)
        with fir.GemmSearch(g, precision) as m:
            m.search_top1_keys_dev(tq.data_ptr(), q.shape[0], keys.data_ptr())
            torch.cuda.synchronize()
            st = m.stats()
            st["kernel"] = g.last_dispatch()["kernel"]
        idx, dist = fir.keys_unpack(keys.cpu().numpy().view(np.uint64))
    return (idx, dist), (eidx, edist), st


@pytest.fixture(params=["default", "sample-flow", "adaptive-always"])
def flow(request, monkeypatch):
    """the hard inputs below must reach BOTH threshold flows of the 16-row fp16 kernels, whatever the shape-based
    choice (adaptive_for, fir_gemm.hip) would have been: FIR_GEMM_ADAPTIVE=0 is the sample flow (k_gemm_proxy_f16x<2, *> +
    <1, *>), 2 forces the in-flight threshold (<3, *>) at any shape. Both flows are sound: the knob changes no answer."""
    if request.param == "sample-flow":
        monkeypatch.setenv("FIR_GEMM_ADAPTIVE", "0")
    elif request.param == "adaptive-always":
        monkeypatch.setenv("FIR_GEMM_ADAPTIVE", "2")
    return request.param


def check_flow(flow, precision, st):
    """the kernel that ran is the one the parametrisation asked for (fp16 form only: the other precisions have one flow)"""
    if precision != 2:
        if flow != "default":
            pytest.skip("the f32 / bf16 forms have one threshold flow")
        return
    if flow == "adaptive-always":
        assert "f16x<3," in st["kernel"], st["kernel"]
    elif flow == "sample-flow":
        assert "f16x<1," in st["kernel"], st["kernel"]


@pytest.mark.parametrize("seed,n,d,qb", [(1, 5000, 512, 70), (2, 40000, 512, 64), (3, 1000, 256, 5), (4, 333, 100, 130), (5, 7, 64, 3), (6, 70000, 128, 200), (7, 3000, 1280, 70), (8, 900, 1536, 9), (9, 2000, 520, 33)])
@pytest.mark.parametrize("precision", [0, 1, 2])
def test_gemm_equals_scan(fir, oracle, seed, n, d, qb, precision):
    rows = synth.make_gallery(seed, n, d, 0)
    q, _ = synth.make_queries(seed, rows, qb, 0)
    (idx, dist), (eidx, edist), st = run_both(fir, rows, q, precision)
    assert np.array_equal(idx, eidx)
    assert np.array_equal(dist.view(np.uint32), edist.view(np.uint32))
    for i in (0, qb - 1):
        assert (idx[i], dist[i]) == oracle.recognize_bf(rows, q[i], 0, d, 0)
    if precision != 2 or n >= 1000:
        assert st["fallback_queries"] <= qb // 4 + 2, st  # the certificate normally holds on random data (fp16: given enough rows to sample)


@pytest.mark.parametrize("n,d,qb", [(3000, 48, 128), (3000, 100, 129), (9000, 512, 257), (2500, 700, 192), (1200, 1280, 130),
                                    (5000, 264, 513), (70, 512, 640), (4000, 40, 65)])
def test_paired_passes_all_shapes(fir, n, d, qb):
    """The 128-queries-per-gallery-read kernel (pairs of 64-query passes): feature counts whose last LDS slab holds 4, 8
    or 12 k-blocks, an odd number of passes (the last one goes through the 64-query kernel), a ragged last pass, more
    than one super-batch, fewer rows than one workgroup covers."""
    rows = synth.make_gallery(100 + d, n, d, 0)
    q, _ = synth.make_queries(100 + d, rows, qb, 0)
    if n > 1000:
        rows[n - 1] = rows[17]                   # an exact tie across row blocks: the lower row wins
        q[3] = rows[17]
    for precision in (1, 2):
        (idx, dist), (eidx, edist), st = run_both(fir, rows, q, precision)
        assert np.array_equal(idx, eidx), precision
        assert np.array_equal(dist.view(np.uint32), edist.view(np.uint32)), precision


@pytest.mark.parametrize("scale", [1e-20, 1e-6, 30.0, 1e15, 1e19])
@pytest.mark.parametrize("precision", [0, 1, 2])
def test_extreme_magnitudes_fall_back_to_the_same_answers(fir, scale, precision, flow):
    """Inputs far from unit norm: products in the denormal range, distances beyond the 100000 cut-off, squares that
    overflow. Whatever the proxies become, the certificate or the fallback returns the exact scan's keys."""
    rows = (synth.make_gallery(55, 6000, 128, 0) * np.float32(scale)).astype(np.float32)
    q = (synth.make_queries(55, rows / np.float32(scale), 70, 0)[0] * np.float32(scale)).astype(np.float32)
    if precision != 2 and flow != "default":
        pytest.skip("the f32 / bf16 forms have one threshold flow")
    (idx, dist), (eidx, edist), st = run_both(fir, rows, q, precision)
    check_flow(flow, precision, st)
    assert np.array_equal(idx, eidx)
    assert np.array_equal(dist.view(np.uint32), edist.view(np.uint32))


@pytest.mark.parametrize("precision", [0, 1, 2])
@pytest.mark.parametrize("qb", [12, 200])
def test_gemm_adversarial_near_ties_and_duplicates(fir, oracle, precision, flow, qb):
    """20 rows within a few ulps of the best, exact duplicates of the best, a NaN row and unnormalised rows: every row
    inside the rounding window of the best proxy is re-ranked exactly, so the lowest row of a tie wins as in the scan."""
    n, d = 30000, 512
    if precision != 2 and flow != "default":
        pytest.skip("the f32 / bf16 forms have one threshold flow")
    rows = synth.make_gallery(9, n, d, 0)
    q, pick = synth.make_queries(9, rows, qb, 0)              # (200: a full pair of 64-query passes and a half-filled one)
    base = rows[777].copy()
    for j in range(20):                       # near-duplicates of one row, spread over the gallery
        v = base.copy()
        v[j] = np.nextafter(v[j], np.float32(2), dtype=np.float32)
        rows[1000 + 1400 * j] = v
    q[0] = base
    q[1] = base * np.float32(1.0000001)
    for r in (5, 29999, 12345):               # exact duplicates: the lowest row must win
        rows[r] = rows[4242]
    q[2] = rows[4242]
    rows[100] = np.nan
    rows[200] *= np.float32(50.0)             # a long row: |g|^2 enters the proxy and the error bound
    q[3] = rows[200]
    if qb > 128:
        q[130], q[131], q[199] = base, rows[4242], rows[200]     # the same hard queries in the half-filled pair
    (idx, dist), (eidx, edist), st = run_both(fir, rows, q, precision)
    check_flow(flow, precision, st)
    assert np.array_equal(idx, eidx)
    assert np.array_equal(dist.view(np.uint32), edist.view(np.uint32))
    assert idx[2] == 5 and dist[2] == 0
    for i in range(4):
        assert (idx[i], dist[i]) == oracle.recognize_bf(rows, q[i], 0, d, 0)


def test_gemm_nothing_found(fir):
    rows = np.full((300, 64), 1.0e4, np.float32)
    q = np.zeros((3, 64), np.float32)
    (idx, dist), (eidx, edist), _ = run_both(fir, rows, q)
    assert np.all(idx == -1) and np.array_equal(idx, eidx) and np.array_equal(dist, edist)


def test_host_call_auto_dispatch(fir):
    """fir_gallery_set_large_batch_mfma: the plain host-pointer search uses the matrix cores for big batches -- same answers."""
    rows = synth.make_gallery(21, 20000, 256, 0)
    q, _ = synth.make_queries(21, rows, 300, 0)
    with fir.Gallery(rows, None, 0, 0) as g:
        a = g.search_top1(q)
        g.set_large_batch_mfma(128)
        b = g.search_top1(q)             # 300 >= 128: GEMM path
        c = g.search_top1(q[:50])        # below the threshold: exact scan
        d = g.search_top1(q, 0, 64)      # a feature prefix of whole k-blocks: the matrix cores too (its own fp16 copy)
        g.set_row_offset(1000)           # the cached GEMM state follows the offset
        e = g.search_top1(q)
        g.set_large_batch_mfma(0)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
    assert np.array_equal(a[0][:50], c[0])
    assert d[0].shape == (300,)
    with fir.Gallery(rows, None, 0, 0) as g2:
        g2.set_large_batch_mfma(0)
        d_scan = g2.search_top1(q, 0, 64)
    assert np.array_equal(d[0], d_scan[0]) and np.array_equal(d[1].view(np.uint32), d_scan[1].view(np.uint32))
    assert np.array_equal(e[0], a[0] + 1000) and np.array_equal(e[1].view(np.uint32), a[1].view(np.uint32))


def test_fp16_term_wide_dynamic_range_and_mixed_scales(fir):
    """FIR_GEMM_F16 scales the gallery by one power of two and every query by its own: rows and queries whose values span
    many binades (elements far below the largest lose their low bits in fp16, some flush to zero), queries at very
    different magnitudes in one batch, an all-zero query and an infinite one. Same keys as the scan, whatever is certified."""
    rng = np.random.default_rng(5)
    n, d, qb = 20000, 320, 140                      # 320 features: 20 k-blocks, the ring does not run on into the next row block
    rows = synth.make_gallery(61, n, d, 0)
    rows *= np.exp2(rng.integers(-12, 1, size=(n, 1))).astype(np.float32)        # row norms over 12 binades
    rows[:, ::5] *= np.float32(2.0 ** -20)                                        # a fifth of the features 20 binades down
    q, _ = synth.make_queries(61, rows, qb, 0)
    q *= np.exp2(rng.integers(-30, 30, size=(qb, 1))).astype(np.float32)
    q[7] = 0
    q[8, 3] = np.inf
    q[9] = rows[123]
    (idx, dist), (eidx, edist), st = run_both(fir, rows, q, 2)
    assert np.array_equal(idx, eidx)
    assert np.array_equal(dist.view(np.uint32), edist.view(np.uint32), )
    assert idx[9] == 123


@pytest.mark.parametrize("precision", [0, 1, 2])
def test_more_ties_than_the_candidate_list_holds(fir, precision, flow):
    """5000 copies of one row: more entries below tau than a query's candidate list (4096) can take. The overflow is
    detected, the certificate refuses, the second pass (fp16 form) overflows as well -- the ties are all inside one window of
    the best -- and the exact device scan answers: the lowest copy."""
    if precision != 2 and flow != "default":
        pytest.skip("the f32 / bf16 forms have one threshold flow")
    n, d = 9000, 128
    rows = synth.make_gallery(77, n, d, 0)
    rows[2000:7000] = rows[10]
    q, _ = synth.make_queries(77, rows, 66, 0)
    q[0] = rows[10]
    q[1] = rows[10] * np.float32(1.001)
    (idx, dist), (eidx, edist), st = run_both(fir, rows, q, precision)
    check_flow(flow, precision, st)
    assert np.array_equal(idx, eidx)
    assert np.array_equal(dist.view(np.uint32), edist.view(np.uint32))
    assert idx[0] == 10 and dist[0] == 0
    assert st["fallback_queries"] >= 1 and st["second_pass_queries"] >= st["fallback_queries"]


@pytest.mark.parametrize("precision", [1, 2])
@pytest.mark.parametrize("spread", [0.3, 0.05, 0.005])
def test_clustered_gallery_like_identities(fir, precision, spread, flow):
    """A gallery of 600 identities x 40 images (each image = the identity's centre + noise of the given relative size), queries
    drawn the same way: dozens of rows sit within the proxy's rounding window of the best one. They are all re-ranked
    exactly, so the keys equal the scan's, and with a window re-rank only a handful of queries may need the exact scan."""
    rng = np.random.default_rng(17)
    ids, per, d, qb = 600, 40, 256, 200
    centres = rng.random((ids, d), dtype=np.float32)
    rows = np.repeat(centres, per, axis=0) * (1 + spread * (rng.random((ids * per, d), dtype=np.float32) - 0.5))
    rows = synth.normalise(rows, 0)
    who = rng.integers(0, ids, qb)
    q = synth.normalise(centres[who] * (1 + spread * (rng.random((qb, d), dtype=np.float32) - 0.5)), 0)
    if precision != 2 and flow != "default":
        pytest.skip("the f32 / bf16 forms have one threshold flow")
    (idx, dist), (eidx, edist), st = run_both(fir, rows, q, precision)
    check_flow(flow, precision, st)
    assert np.array_equal(idx, eidx)
    assert np.array_equal(dist.view(np.uint32), edist.view(np.uint32))
    assert np.all(idx // per == who)                      # the nearest image belongs to the query's identity
    assert st["fallback_queries"] <= qb // 10, st


@pytest.mark.parametrize("precision", [1, 2])
def test_long_rows(fir, precision):
    """2048- and 4100-feature rows: the re-rank's LDS tile (query + candidate rows) needs more than the default 64 KiB, and
    beyond ~4000 features fewer than eight candidate rows fit at a time."""
    for d, n, qb in ((2048, 1500, 66), (4100, 700, 40)):
        rows = synth.make_gallery(200 + d, n, d, 0)
        q, _ = synth.make_queries(200 + d, rows, qb, 0)
        rows[n - 1] = rows[3]
        q[0] = rows[3]
        (idx, dist), (eidx, edist), st = run_both(fir, rows, q, precision)
        assert np.array_equal(idx, eidx), d
        assert np.array_equal(dist.view(np.uint32), edist.view(np.uint32)), d
        assert idx[0] == 3


def test_default_dispatch_takes_the_matrix_cores_for_big_batches_over_big_galleries(fir, oracle):
    """No opt-in: >= 128 queries against >= 65536 rows (L2, whole range) go through fir_gemm_* by default, smaller batches,
    smaller galleries, sub-ranges and chi-square stay with the exact scan; fir_gallery_last_dispatch says which kernel ran;
    set_large_batch_mfma(0) is the opt-out. Keys are identical either way, and the oracle's on a sample."""
    n, d = 70000, 64
    rows = synth.make_gallery(23, n, d, 0)
    q, _ = synth.make_queries(23, rows, 300, 0)
    rows[n - 1] = rows[7]
    q[3] = rows[7]                                    # exact tie: the first row wins on either path
    with fir.Gallery(rows, None, 0, 0) as g:
        a = g.search_top1(q)
        da = g.last_dispatch()
        assert da["path"] == "mfma" and "k_gemm_proxy_f16" in da["kernel"] and da["queries_per_pass"] % 128 == 0
        assert da["lds_bytes"] >= 64 * 1024 and da["vgprs"] > 0 and da["flops_per_launch"] > 0 and da["bytes_per_launch"] > 0
        b = g.search_top1(q[:100])                    # below 128 queries
        assert g.last_dispatch()["path"] == "scan"
        c = g.search_top1(q, 0, 32)                   # a sub-range
        assert g.last_dispatch()["path"] == "scan"
        g.set_large_batch_mfma(0)                     # opt-out
        e = g.search_top1(q)
        de = g.last_dispatch()
        assert de["path"] == "scan" and "k_scan" in de["kernel"] and de["flops_per_launch"] == 0
        g.set_large_batch_mfma(-1)                    # back to the default
        f = g.search_top1(q)
        assert g.last_dispatch()["path"] == "mfma"
    assert np.array_equal(a[0], e[0]) and np.array_equal(a[1].view(np.uint32), e[1].view(np.uint32))
    assert np.array_equal(a[0], f[0]) and np.array_equal(a[0][:100], b[0]) and c[0].shape == (300,)
    assert a[0][3] == 7
    eidx, edist = oracle.top1_batch(rows, q[:40], 0, d, 0)
    assert np.array_equal(a[0][:40], eidx) and np.array_equal(a[1][:40].view(np.uint32), edist.view(np.uint32))
    small = synth.make_gallery(24, 20000, 64, 0)
    with fir.Gallery(small, None, 0, 0) as g:         # below 65536 rows: the scan, unless asked for
        g.search_top1(q)
        assert g.last_dispatch()["path"] == "scan"
        g.set_large_batch_mfma(128)
        g.search_top1(q)
        assert g.last_dispatch()["path"] == "mfma"
    with fir.Gallery(synth.make_gallery(25, 70000, 64, 1), None, 1, 0) as g:     # chi-square never
        g.search_top1(synth.make_queries(25, rows, 130, 1)[0])
        assert g.last_dispatch()["path"] == "scan"


def test_class_ordered_gallery_is_sampled_across_all_classes(fir):
    """The reference's galleries are ordered by class. The threshold comes from a row sample; a prefix sample would see the
    first classes only and a query of a late class would pass thousands of rows (list overflow -> exact scan). The sample
    is every n-th row block: no fallbacks, and the exact scan's keys."""
    rng = np.random.default_rng(3)
    ident, per, d = 700, 100, 64
    centres = rng.random((ident, d)).astype(np.float32)
    rows = np.repeat(centres, per, axis=0) * (1 + 0.02 * (rng.random((ident * per, d)).astype(np.float32) - 0.5))
    rows /= np.linalg.norm(rows, axis=1, keepdims=True)
    pick = np.arange(300) % 40 + ident * per - 40 * per + (np.arange(300) // 40) * per      # images of the last identities
    q = rows[pick] * (1 + 0.01 * (rng.random((300, d)).astype(np.float32) - 0.5))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    (idx, dist), (eidx, edist), st = run_both(fir, rows, q)
    assert np.array_equal(idx, eidx) and np.array_equal(dist.view(np.uint32), edist.view(np.uint32))
    assert st["fallback_queries"] == 0


@pytest.mark.parametrize("n,d,qb", [(70000, 64, 300), (66000, 512, 200), (80000, 200, 129)])
def test_topk_through_the_matrix_cores_equals_the_exact_scan(fir, oracle, n, d, qb):
    """fir_search_topk on >= 128 whole-range L2 queries over >= 65536 rows takes the fp16 nomination pass with the re-rank
    window hung on the K-th smallest proxy and the certificate against the K-th exact distance: the keys of the exact
    top-K scan (set_large_batch_mfma(0)) for every K, with ties inside the list and at the K-th place, and the oracle's."""
    rows = synth.make_gallery(n % 97, n, d, 0)
    q, _ = synth.make_queries(n % 97, rows, qb, 0)
    rows[200:205] = rows[n - 10]                      # six equal rows
    q[1] = rows[n - 10]
    for k in (2, 5, 8):
        with fir.Gallery(rows, None, 0, 0) as g:
            a = g.search_topk(q, k)
            g.set_large_batch_mfma(0)
            e = g.search_topk(q, k)
        assert np.array_equal(a[0], e[0]), k
        assert np.array_equal(a[1].view(np.uint32), e[1].view(np.uint32)), k
        assert list(a[0][1][: min(k, 6)]) == [200, 201, 202, 203, 204, n - 10][: min(k, 6)]
        for j in (0, 1, qb - 1):
            ei, ed = oracle.topk(rows, q[j], 0, d, k, 0)
            assert np.array_equal(a[0][j], ei) and np.array_equal(a[1][j].view(np.uint32), ed.view(np.uint32))


def test_topk_matrix_cores_on_a_clustered_gallery_and_by_handle(fir):
    """Near-duplicate rows (600 identities x ~117 images each): dozens of rows within the rounding window of the K-th best;
    what the certificate cannot prove goes to the exact scan -- same keys either way. Also through fir_gemm_search_topk_keys_dev
    directly, with its fallback count."""
    rng = np.random.default_rng(5)
    n, d, k = 70000, 128, 5
    centres = synth.make_gallery(3, 600, d, 0)
    rows = centres[np.arange(n) % 600] + rng.normal(0, 2e-4, (n, d)).astype(np.float32)
    rows = (rows / np.linalg.norm(rows, axis=1, keepdims=True)).astype(np.float32)
    q = rows[rng.integers(0, n, 256)] + rng.normal(0, 1e-4, (256, d)).astype(np.float32)
    q = np.ascontiguousarray(q, np.float32)
    dev = torch.device("cuda", 0)
    with fir.Gallery(rows, None, 0, 0) as g:
        a = g.search_topk(q, k)
        g.set_large_batch_mfma(0)
        e = g.search_topk(q, k)
        tq = torch.from_numpy(q).to(dev)
        keys = torch.empty(q.shape[0] * k, dtype=torch.int64, device=dev)
        with fir.GemmSearch(g, 2) as m:
            m.search_topk_keys_dev(tq.data_ptr(), q.shape[0], k, keys.data_ptr())
            torch.cuda.synchronize()
            st = m.stats()
        hidx, hdist = fir.keys_unpack(keys.cpu().numpy().view(np.uint64))
    assert np.array_equal(a[0], e[0]) and np.array_equal(a[1].view(np.uint32), e[1].view(np.uint32))
    assert np.array_equal(hidx.reshape(-1, k), e[0]) and np.array_equal(hdist.reshape(-1, k).view(np.uint32), e[1].view(np.uint32))
    assert st["fallback_queries"] <= 256


@pytest.mark.parametrize("n,d,end", [(70000, 512, 64), (70000, 512, 256), (66000, 200, 128), (70000, 1536, 64)])
def test_feature_prefixes_through_the_matrix_cores(fir, oracle, n, d, end):
    """The reference's "BF, 64" / "BF, 256" classifiers compare a prefix [0, end) of every row (ImageTesting.cpp:526-529). Large
    batches of such calls take the matrix cores too (own fp16 fragments and norms of the prefix): the exact scan's keys, for
    top-1 and top-K, ties and all; another prefix replaces the cached one; prefixes that are not whole k-blocks stay with the scan."""
    rows = synth.make_gallery(end, n, d, 0)
    q, _ = synth.make_queries(end, rows, 200, 0)
    rows[300, :end] = rows[n - 5, :end]               # equal on the prefix, different beyond it
    q[2] = rows[n - 5]
    with fir.Gallery(rows, None, 0, 0) as g:
        a = g.search_top1(q, 0, end)
        da = g.last_dispatch()
        a5 = g.search_topk(q, 5, 0, end)
        w = g.search_top1(q)                          # the whole row in between
        other = 128 if end != 128 else 64
        b = g.search_top1(q, 0, other)
        odd = g.search_top1(q, 0, end + 4)            # not a multiple of 16
        assert g.last_dispatch()["path"] == "scan"
        g.set_large_batch_mfma(0)
        e = g.search_top1(q, 0, end)
        e5 = g.search_topk(q, 5, 0, end)
        ew = g.search_top1(q)
        eb = g.search_top1(q, 0, other)
    assert da["path"] == "mfma"
    for x, y in ((a, e), (a5, e5), (w, ew), (b, eb)):
        assert np.array_equal(x[0], y[0]) and np.array_equal(x[1].view(np.uint32), y[1].view(np.uint32))
    assert a[0][2] == 300 and odd[0].shape == (200,)
    for j in (0, 2, 199):
        assert (a[0][j], a[1][j]) == oracle.recognize_bf(rows, q[j], 0, end, 0)


def test_topk_matrix_cores_with_hostile_rows_and_queries(fir):
    """Top-K through the matrix cores with a NaN row, a NaN query, an all-zero query, a query of huge values and more exact
    duplicates of the best row than K: whatever the certificate cannot prove goes to the exact top-K scan -- same keys."""
    n, d, k = 70000, 128, 5
    rows = synth.make_gallery(31, n, d, 0)
    q, _ = synth.make_queries(31, rows, 160, 0)
    rows[123] = np.nan
    for r in (9, 500, 501, 40000, 69999, 69998, 31):     # seven copies of one row
        rows[r] = rows[7777]
    q[0] = rows[7777]
    q[1] = np.nan
    q[2] = 0.0
    q[3] = q[3] * np.float32(3.0e18)
    q[4] = rows[123 + 1]
    with fir.Gallery(rows, None, 0, 0) as g:
        a = g.search_topk(q, k)
        assert g.last_dispatch()["path"] == "mfma"
        g.set_large_batch_mfma(0)
        e = g.search_topk(q, k)
    assert np.array_equal(a[0], e[0]) and np.array_equal(a[1].view(np.uint32), e[1].view(np.uint32))
    assert list(a[0][0]) == [9, 31, 500, 501, 7777]


def test_default_dispatch_takes_small_batches_on_large_galleries(fir):
    """The automatic threshold falls with the size of the gallery (a matrix-core call costs about the same from 2 to 128
    queries, the scan one gallery pass per power-of-two group of up to 8): 32 queries from 128 MB of compared rows on, fewer where
    the cost model says so (7 = 4 + 2 + 1 queries are three scan passes; since round 4 a call of up to 32 queries multiplies against its
    live query blocks only and 2 or 4 queries over 2 GB of rows go through the matrix cores as well: 265 against 340 us). Same keys."""
    dev = torch.device("cuda", 0)
    d = 512
    for n, qb, want in ((70_000, 32, "mfma"), (70_000, 16, "scan"), (400_000, 16, "mfma"), (400_000, 8, "scan"), (400_000, 7, "mfma"),
                        (1_000_000, 8, "mfma"), (1_000_000, 3, "mfma"), (1_000_000, 4, "mfma"), (1_000_000, 2, "mfma"),
                        # the few-block forms' edges: 16 queries fill one query block, 17 start the second, 32 fill it, 33 take the whole tile
                        (1_000_000, 16, "mfma"), (1_000_000, 17, "mfma"), (1_000_000, 32, "mfma"), (1_000_000, 33, "mfma")):
        x = torch.rand((n, d), device=dev)
        x = (x / x.norm(dim=1, keepdim=True)).contiguous()
        q = (x[:: n // qb][:qb] * 0.97 + x[1: qb + 1] * 0.03).contiguous()
        g = fir.Gallery(dev_ptr=x.data_ptr(), n=n, d=d, metric=0, device=0)
        k1 = torch.empty(qb, dtype=torch.int64, device=dev)
        k2 = torch.empty(qb, dtype=torch.int64, device=dev)
        g.search_top1_keys_dev(q.data_ptr(), qb, k1.data_ptr())
        g.sync()
        assert g.last_dispatch()["path"] == want, (n, qb)
        if n == 1_000_000 and want == "mfma":
            kern = g.last_dispatch()["kernel"]
            assert ("f16x<3, 0, 0, 0, 1>" in kern) == (qb <= 16) and ("f16x<3, 0, 0, 0, 2>" in kern) == (16 < qb <= 32), (qb, kern)
        g.set_large_batch_mfma(0)
        g.search_top1_keys_dev(q.data_ptr(), qb, k2.data_ptr())
        g.sync()
        assert torch.equal(k1, k2), (n, qb)
        g.close()
        del x, q
    torch.cuda.empty_cache()


def test_few_queries_on_a_large_gallery_take_the_fp16_nomination_scan(fir):
    """ONE L2 query against >= 1.5 GB of rows: one pass over the fp16 copy (k_gemm_scan_f16: v_dot2, no matrix cores), every row
    within one rounding window of the smallest proxy of ALL rows re-ranked exactly -- the exact scan's key, with an exact duplicate
    (lowest row wins), a near-duplicate one ulp away, a NaN query, through device and host pointers; and the same form by handle
    for up to 8 queries."""
    dev = torch.device("cuda", 0)
    n, d = 800_000, 512
    x = torch.rand((n, d), device=dev)
    x = (x / x.norm(dim=1, keepdim=True)).contiguous()
    x[650_000] = x[777]                                    # exact duplicate: row 777 wins
    x[123_456] = x[999]
    x[123_456, 3] = torch.nextafter(x[999, 3], torch.tensor(2.0, device=dev))
    q = (x[torch.tensor([777, 999, 5, 799_999, 64, 100_001, 7], device=dev)] * 0.99 + x[:7] * 0.01).contiguous()
    q[0] = x[777]
    q[1] = x[999]
    q = torch.cat([q, torch.full((1, d), float("nan"), device=dev)]).contiguous()
    g = fir.Gallery(dev_ptr=x.data_ptr(), n=n, d=d, metric=0, device=0)
    k1 = torch.empty(1, dtype=torch.int64, device=dev)
    k2 = torch.empty(1, dtype=torch.int64, device=dev)
    for i in range(8):
        qi = q[i:i + 1].contiguous()
        g.set_large_batch_mfma(-1)
        for call in range(17 if i == 0 else 1):          # the first 16 one-query calls of a gallery take the scan (the fp16 copy is not built for a handful)
            g.search_top1_keys_dev(qi.data_ptr(), 1, k1.data_ptr())
            g.sync()
            if i == 0 and call < 16:
                assert g.last_dispatch()["path"] == "scan"
        dsp = g.last_dispatch()
        assert dsp["path"] == "mfma" and "k_gemm_scan_f16" in dsp["kernel"], dsp
        hidx, hdist = g.search_top1(qi.cpu().numpy())               # host pointers: the same form
        assert "k_gemm_scan_f16" in g.last_dispatch()["kernel"]
        g.search_top1_keys_dev(q.data_ptr(), 2, torch.empty(2, dtype=torch.int64, device=dev).data_ptr())
        g.sync()
        assert g.last_dispatch()["path"] == "mfma" and "k_gemm_proxy_f16x" in g.last_dispatch()["kernel"]   # two queries over 1.6 GB: one query block of the matrix-core pass
        g.set_large_batch_mfma(0)
        g.search_top1_keys_dev(qi.data_ptr(), 1, k2.data_ptr())
        g.sync()
        assert g.last_dispatch()["path"] == "scan"
        assert torch.equal(k1, k2), i
        idx, dist = fir.keys_unpack(k2.cpu().numpy().view(np.uint64))
        assert np.array_equal(hidx, idx) and np.array_equal(hdist.view(np.uint32), dist.view(np.uint32))
        if i == 0:
            assert idx[0] == 777 and dist[0] == 0
        if i == 1:
            assert idx[0] == 999
    with fir.GemmSearch(g, 2) as m:
        for qb in (2, 3, 5, 8):
            kq = torch.empty(qb, dtype=torch.int64, device=dev)
            eq = torch.empty(qb, dtype=torch.int64, device=dev)
            m.search_few_keys_dev(q.data_ptr(), qb, kq.data_ptr())
            torch.cuda.synchronize()
            g.search_top1_keys_dev(q.data_ptr(), qb, eq.data_ptr())   # (the matrix-core path is off: the exact scan)
            g.sync()
            assert torch.equal(kq, eq), qb
        assert m.stats()["fallback_queries"] <= 1                     # the NaN query
    g.close()
    del x
    torch.cuda.empty_cache()


def test_default_dispatch_on_cache_resident_galleries(fir, oracle):
    """Below 65 536 rows a cost model of the two forms decides (the scan folds its 16-query passes into one launch): large batches take
    the matrix cores down to a few thousand rows -- from the fourth such call on: the state is not built for one-off calls --, small
    ones and the reference's own gallery size with a few hundred queries stay with the scan. Same keys either way, the oracle's on a
    sample."""
    for n, d, qb, want in ((40000, 512, 128, "mfma"), (40000, 512, 32, "scan"), (16384, 512, 1024, "mfma"), (16384, 512, 128, "mfma"), (8192, 512, 64, "scan"),
                           (3030, 1536, 256, "scan"), (3030, 1536, 2000, "mfma"), (1500, 512, 2000, "scan")):
        rows = synth.make_gallery(n % 89, n, d, 0)
        q, _ = synth.make_queries(n % 89, rows, qb, 0)
        with fir.Gallery(rows, None, 0, 0) as g:
            for call in range(4):                      # the matrix-core state is only built for a gallery that keeps getting such calls
                a = g.search_top1(q)
                assert g.last_dispatch()["path"] == ("scan" if call < 3 else want), (n, d, qb, call)
            g.set_large_batch_mfma(0)
            e = g.search_top1(q)
        assert np.array_equal(a[0], e[0]) and np.array_equal(a[1].view(np.uint32), e[1].view(np.uint32)), (n, d, qb)
        assert (a[0][0], a[1][0]) == oracle.recognize_bf(rows, q[0], 0, d, 0)


def test_memory_report_shadow_modes_and_the_two_prefix_states(fir):
    """fir_gallery_memory_bytes / fir_gallery_set_shadow_copies / fir_gallery_mfma_stats: the automatic dispatch's
    copies are reported and can be forbidden; alternating prefixes [0, 64) and [0, 256) -- the reference's "BF, 64" / "BF, 256"
    classifiers, ImageTesting.cpp:526-529 -- keep both matrix-core states instead of rebuilding one slot on every call; the
    candidate lists are sized by the batches actually seen. Keys stay the exact scan's in every mode."""
    n, d, qb = 70000, 320, 192
    rows = synth.make_gallery(71, n, d, 0)
    q, _ = synth.make_queries(71, rows, qb, 0)
    with fir.Gallery(rows, None, 0, 0) as g:
        g.set_large_batch_mfma(0)
        ref = {end: g.search_top1(q, 0, end) for end in (64, 256)}
        g.set_large_batch_mfma(-1)
        mem0 = g.memory_bytes()
        assert mem0["tiled"] >= n * d * 4 and mem0["fp16_fragments"] == 0 and mem0["rowmajor_shadow"] == 0
        for mode, want_shadow in ((fir.SHADOW_ALL, True), (fir.SHADOW_FP16, False)):
            g.set_shadow_copies(mode)
            for rep in range(3):
                for end in (64, 256):
                    idx, dist = g.search_top1(q, 0, end)
                    assert g.last_dispatch()["path"] == "mfma"
                    assert np.array_equal(idx, ref[end][0]) and np.array_equal(dist.view(np.uint32), ref[end][1].view(np.uint32))
            mem = g.memory_bytes()
            # both prefix states are alive: two fragment copies, padded to 128-feature units ([0, 64) -> 128, [0, 256) -> 256)
            assert n * (256 + 128) * 2 <= mem["fp16_fragments"] <= n * (256 + 128) * 2 * 1.01 + 4096, mem
            assert (mem["rowmajor_shadow"] >= n * (256 + 64) * 4) == want_shadow, mem
            assert mem["scratch"] < 96 << 20, mem            # two states with lists for 256 queries each (8 MiB), not for 8192 (512 MiB each)
            st = g.mfma_stats()
            assert st["passes"] > 0 and st["fallback_queries"] <= 8
        passes_before = g.mfma_stats()["passes"]
        g.set_shadow_copies(fir.SHADOW_NONE)
        idx, dist = g.search_top1(q, 0, 256)
        assert g.last_dispatch()["path"] == "scan" and np.array_equal(idx, ref[256][0])
        assert np.array_equal(dist.view(np.uint32), ref[256][1].view(np.uint32))
        assert g.memory_bytes()["fp16_fragments"] == 0 and g.mfma_stats()["passes"] == 0 and passes_before > 0


def _coherent_rounding_fixture(n_random=6000, d=512, seed=91):
    """A near-tie the fp16 proxy orders the WRONG way by almost the whole error bound: every component of row A (and of the query)
    is 1 + 0.499 * 2^-10 -- rounds DOWN to 1 in fp16 -- and every component of row B is 1 + 0.501 * 2^-10 -- rounds UP to 1 + 2^-10.
    A is the query itself (distance exactly 0), B lies 2e-9 / d away; but B's proxy is LOWER than A's by 2 * 2^-10 * |q||g| * 0.998
    (all 2 d rounding errors have the same sign), 0.77 of the rounding window 2 E d the re-rank allows for. Random rows far away
    fill the rest of the gallery (values in [0, 1): the power-of-two scale is the same as for A and B)."""
    rng = np.random.default_rng(seed)
    rows = rng.random((n_random, d), dtype=np.float32)
    a = np.full(d, np.float32(1.0) + np.float32(0.499 * 2.0 ** -10), np.float32)
    b = np.full(d, np.float32(1.0) + np.float32(0.501 * 2.0 ** -10), np.float32)
    ia, ib = n_random // 3, 2 * n_random // 3 + 5
    rows[ia], rows[ib] = a, b
    q = np.vstack([a] + [rows[i] * np.float32(0.999) for i in (7, 99, 1234)] + [rng.random(d, dtype=np.float32) for _ in range(60)]).astype(np.float32)
    return rows, q, ia, ib


def test_the_certificate_bound_is_needed_and_a_shrunken_one_is_caught(fir, fir_audit, oracle, monkeypatch):
    """the suite must be able to SEE an unsound error bound. With the bound as derived the
    coherent-rounding near-tie is answered exactly (both rows are inside the rounding window, both are re-ranked, the reference's
    first minimum wins); with FIR_GEMM_EREL_SCALE=0.25 -- an audit knob that multiplies the certificate's E -- row A falls out of
    the window, the certificate 'proves' B and the call returns the wrong row WITHOUT falling back: exactly the failure the bound
    exists to prevent, and this test goes red if it ever stops being detected."""
    rows, q, ia, ib = _coherent_rounding_fixture()
    d = rows.shape[1]
    dev = torch.device("cuda", 0)
    tq = torch.from_numpy(q).to(dev)
    keys = torch.empty(q.shape[0], dtype=torch.int64, device=dev)
    got = {}
    # the audit build (libfir_amd_audit.so) honours the knob; the shipped library must not -- a stray variable cannot break the keys
    for lib_name, pkg in (("audit", fir_audit), ("shipped", fir)):
        with pkg.Gallery(rows, None, 0, 0) as g:
            g.set_large_batch_mfma(0)
            eidx, edist = g.search_top1(q)
            assert eidx[0] == ia and edist[0] == 0.0 and (eidx[0], edist[0]) == oracle.recognize_bf(rows, q[0], 0, d, 0)
            for scale in ("1", "0.25"):
                monkeypatch.setenv("FIR_GEMM_EREL_SCALE", scale)
                with pkg.GemmSearch(g, 2) as m:
                    m.search_top1_keys_dev(tq.data_ptr(), q.shape[0], keys.data_ptr())
                    torch.cuda.synchronize()
                    got[lib_name, scale] = (pkg.keys_unpack(keys.cpu().numpy().view(np.uint64)), m.stats()["fallback_queries"])
                    knobs = g.last_dispatch().get("knobs", "")
                    assert ("FIR_GEMM_EREL_SCALE" in knobs) == (lib_name == "audit"), knobs     # whatever is honoured is reported
            monkeypatch.delenv("FIR_GEMM_EREL_SCALE")
    for key in (("audit", "1"), ("shipped", "1"), ("shipped", "0.25")):
        (idx, dist), fb = got[key]
        assert np.array_equal(idx, eidx) and np.array_equal(dist.view(np.uint32), edist.view(np.uint32)) and fb <= 2, key
    (idx_s, dist_s), fb_s = got["audit", "0.25"]
    # the wrong row, certified (no fall-back to the exact scan for it): the shrunken bound is unsound and it shows
    assert idx_s[0] == ib and dist_s[0] > 0.0 and fb_s <= 2, (idx_s[0], dist_s[0], fb_s)


def _identity_gallery(n_ids, per, d, seed, dev):
    """class-ordered rows = identity centre x (1 +- 2.5 %), unit length: what the reference's galleries look like"""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    centres = torch.rand((n_ids, d), generator=g, device=dev)
    rows = centres.repeat_interleave(per, dim=0) * (1 + 0.05 * (torch.rand((n_ids * per, d), generator=g, device=dev) - 0.5))
    return (rows / rows.norm(dim=1, keepdim=True)).contiguous(), centres


@pytest.mark.parametrize("k", [1, 5])
def test_a_loose_first_bound_costs_a_second_pass_not_the_exact_scan(fir, monkeypatch, k):
    """an overflowing candidate list must be cheap. 25 000 identities x 40 class-ordered rows and a row sample
    cut down to 8 192 rows (FIR_GEMM_SAMPLE_DIV) with the sample flow (FIR_GEMM_ADAPTIVE=0; the K-nearest form always samples): the
    strided sample misses the query's identity, every row of every identity nearer than the nearest SAMPLED one passes the first
    bound and many lists overflow their 4 096 entries. Those queries take the second matrix-core pass with
    min(first bound, smallest stored proxy + one window) -- on the device, in stream order -- and come back certified: nothing
    reaches the exact device scan, and the keys are the exact scan's."""
    dev = torch.device("cuda", 0)
    n_ids, per, d, qb = 25000, 40, 128, 384
    rows, centres = _identity_gallery(n_ids, per, d, 77, dev)
    gq = torch.Generator(device=dev)
    gq.manual_seed(78)
    who = torch.randint(0, n_ids, (qb,), generator=gq, device=dev)
    q = centres[who] * (1 + 0.05 * (torch.rand((qb, d), generator=gq, device=dev) - 0.5))
    q = (q / q.norm(dim=1, keepdim=True)).contiguous()
    torch.cuda.synchronize()                               # (the library's streams do not wait for torch's)
    monkeypatch.setenv("FIR_GEMM_ADAPTIVE", "0")
    monkeypatch.setenv("FIR_GEMM_SAMPLE_DIV", "1000000")
    ke = torch.empty(qb * k, dtype=torch.int64, device=dev)
    km = torch.empty(qb * k, dtype=torch.int64, device=dev)
    with fir.Gallery(dev_ptr=rows.data_ptr(), n=n_ids * per, d=d, metric=0, device=0) as g:
        g.set_large_batch_mfma(0)
        if k == 1:
            g.search_top1_keys_dev(q.data_ptr(), qb, ke.data_ptr())
        else:
            g.search_topk_keys_dev(q.data_ptr(), qb, k, ke.data_ptr())
        torch.cuda.synchronize()
        with fir.GemmSearch(g, 2) as m:
            if k == 1:
                m.search_top1_keys_dev(q.data_ptr(), qb, km.data_ptr())
            else:
                m.search_topk_keys_dev(q.data_ptr(), qb, k, km.data_ptr())
            torch.cuda.synchronize()
            st = m.stats()
            assert "FIR_GEMM_SAMPLE_DIV" in g.last_dispatch()["knobs"]
    assert torch.equal(ke, km)
    idx, _ = fir.keys_unpack(km.cpu().numpy().view(np.uint64))
    assert np.all(idx.reshape(qb, k)[:, 0] // per == who.cpu().numpy())
    assert st["second_pass_queries"] >= 8, st             # the first bound WAS loose ...
    assert st["fallback_queries"] == 0, st                # ... and the second pass answered all of them
    if k == 1:
        # the same through the HOST-pointer entry point (queries staged super-batch by super-batch) and the gallery's own dispatch
        qh = q.cpu().numpy()
        with fir.Gallery(dev_ptr=rows.data_ptr(), n=n_ids * per, d=d, metric=0, device=0) as g:
            g.set_large_batch_mfma(1)
            hidx, hdist = g.search_top1(qh)
            sth = g.mfma_stats()
            notes = g.uncertified_notes()
        eidx, edist = fir.keys_unpack(ke.cpu().numpy().view(np.uint64))
        assert np.array_equal(hidx, eidx) and np.array_equal(hdist.view(np.uint32), edist.view(np.uint32))
        assert sth["second_pass_queries"] >= 8 and sth["fallback_queries"] == 0, sth
        assert len(notes) >= 1 and notes[0][0] > 4096 and np.isfinite(notes[0][1]), notes     # the diagnostics say: list overflow under a finite bound


def test_the_device_pointer_call_returns_before_its_kernels_have_run(fir):
    """no host synchronisation inside fir_search_top1_keys_dev on the matrix-core path -- the certified /
    uncertified split, the second pass and the exact scan of the rest are all queued on the caller's stream. Two calls are
    enqueued back to back; when the second one returns the stream still has work (the first call's 8 192-query pass alone is
    tens of milliseconds at 300 000 x 512), and a NaN query -- which no certificate holds for -- is answered in stream order."""
    dev = torch.device("cuda", 0)
    n, d, qb = 300_000, 512, 8192
    g0 = torch.Generator(device=dev)
    g0.manual_seed(5)
    rows = torch.rand((n, d), generator=g0, device=dev)
    rows = (rows / rows.norm(dim=1, keepdim=True)).contiguous()
    q = torch.rand((qb, d), generator=g0, device=dev)
    q = (q / q.norm(dim=1, keepdim=True)).contiguous()
    q[5, 7] = float("nan")
    q[6] = rows[1234]
    torch.cuda.synchronize()                               # (the library's streams do not wait for torch's)
    k1 = torch.empty(qb, dtype=torch.int64, device=dev)
    k2 = torch.empty(qb, dtype=torch.int64, device=dev)
    ke = torch.empty(qb, dtype=torch.int64, device=dev)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        with fir.Gallery(dev_ptr=rows.data_ptr(), n=n, d=d, metric=0, device=0, stream=st.cuda_stream) as g:
            g.search_top1_keys_dev(q.data_ptr(), qb, k1.data_ptr(), stream=st.cuda_stream)      # (builds the fp16 copy: synchronises once)
            st.synchronize()
            assert g.last_dispatch()["path"] == "mfma"
            g.search_top1_keys_dev(q.data_ptr(), qb, k1.data_ptr(), stream=st.cuda_stream)
            g.search_top1_keys_dev(q.data_ptr(), qb, k2.data_ptr(), stream=st.cuda_stream)
            still_running = not st.query()
            st.synchronize()
            stats = g.mfma_stats()
            g.set_large_batch_mfma(0)
            g.search_top1_keys_dev(q.data_ptr(), qb, ke.data_ptr(), stream=st.cuda_stream)
            st.synchronize()
    assert still_running, "the calls returned only after their work was done: something synchronised"
    assert torch.equal(k1, ke) and torch.equal(k2, ke)
    idx, _ = fir.keys_unpack(k1.cpu().numpy().view(np.uint64))
    assert idx[5] == -1 and idx[6] == 1234
    assert 3 <= stats["fallback_queries"] <= 6, stats     # the NaN query of each of the three calls went to the exact device scan


@pytest.mark.parametrize("qb", [5, 16, 24])
def test_small_calls_with_hard_queries_at_1m(fir, qb):
    """The few-block forms of the matrix-core pass (k_gemm_proxy_f16x<3, 0, 0, 0, 1 | 2>: a call of at most 16 / 32 queries multiplies
    against its live query blocks only, its first row block is summed once, and what it cannot certify goes straight to the exact
    device scan) with the queries that make trouble: a NaN component, an infinite one, an all-zero query, an exact copy of a row that
    has a duplicate further down (the lower row wins), a query one ulp from a row, a query midway between two rows. Keys = the exact
    scan's, several calls in a row (state carried between calls would show)."""
    dev = torch.device("cuda", 0)
    n, d = 1_000_000, 512
    g0 = torch.Generator(device=dev)
    g0.manual_seed(77 + qb)
    x = torch.rand((n, d), generator=g0, device=dev)
    x = (x / x.norm(dim=1, keepdim=True)).contiguous()
    x[900_001] = x[4321]                                       # duplicate further down
    q = torch.rand((qb, d), generator=g0, device=dev)
    q = (q / q.norm(dim=1, keepdim=True)).contiguous()
    q[0, 3] = float("nan")
    q[1, 5] = float("inf")
    q[2] = 0.0
    q[3] = x[4321]
    q[4] = x[777_777]
    q[4, 9] = torch.nextafter(q[4, 9], torch.tensor(2.0, device=dev))
    if qb > 5:
        q[5] = 0.5 * (x[10] + x[999_990])
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    ka = torch.empty(qb, dtype=torch.int64, device=dev)
    ke = torch.empty(qb, dtype=torch.int64, device=dev)
    with torch.cuda.stream(st):
        with fir.Gallery(dev_ptr=x.data_ptr(), n=n, d=d, metric=0, device=0, stream=st.cuda_stream) as g:
            g.set_large_batch_mfma(0)
            g.search_top1_keys_dev(q.data_ptr(), qb, ke.data_ptr(), stream=st.cuda_stream)
            st.synchronize()
            g.set_large_batch_mfma(-1)
            for call in range(4):
                ka.fill_(-7)
                g.search_top1_keys_dev(q.data_ptr(), qb, ka.data_ptr(), stream=st.cuda_stream)
                st.synchronize()
                kern = g.last_dispatch()["kernel"]
                assert ("f16x<3, 0, 0, 0, 1>" in kern) if qb <= 16 else ("f16x<3, 0, 0, 0, 2>" in kern), kern
                assert torch.equal(ka, ke), (call, qb)
            stats = g.mfma_stats()
    idx, dist = fir.keys_unpack(ke.cpu().numpy().view(np.uint64))
    assert idx[0] == -1 and idx[3] == 4321 and dist[3] == 0 and idx[4] == 777_777
    assert stats["second_pass_queries"] >= 4 and stats["fallback_queries"] >= 4, stats      # the NaN query of every call: no second-chance round below 33 queries, the exact device scan
