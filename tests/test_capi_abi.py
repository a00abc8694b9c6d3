"""The C-ABI library loads without a GPU and exports exactly what include/fir_amd.h declares;
the host-only entry points (key packing) behave; device entry points fail loudly without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "fir_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fir_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(fir):
    L = ctypes.CDLL(fir.lib_path())
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/fir_amd.h but not exported"
    bound = {s[0] for s in fir.capi.SYMBOLS}
    assert bound == set(names), (bound ^ set(names))


def test_the_audit_build_is_the_same_abi_and_the_shipped_library_has_no_audit_knobs(fir, fir_audit):
    """libfir_amd_audit.so (-DFIR_AUDIT) exports the same symbols; the strings of the knobs that can change answers exist in the audit
    library only -- a stray environment variable cannot reach them in the product."""
    A = ctypes.CDLL(fir_audit.lib_path())
    for n in declared_symbols():
        assert hasattr(A, n), n
    assert fir_audit.lib_path() != fir.lib_path() and fir_audit.lib_path().endswith("libfir_amd_audit.so")
    shipped = open(fir.lib_path(), "rb").read()
    audit = open(fir_audit.lib_path(), "rb").read()
    for knob in (b"FIR_GEMM_EREL_SCALE", b"FIR_GEMM_DBG_SKIP", b"FIR_GEMM_ADAPT_DBG", b"FIR_GEMM_DEBUG_COUNTS"):
        assert knob in audit and knob not in shipped, knob


def test_no_torch_or_oracle_in_the_product_library(fir):
    out = os.popen(f"ldd {fir.lib_path()}").read()
    assert "libamdhip64" in out
    assert "librccl" in out        # the row-sharded gallery reduces its keys with RCCL inside the library
    assert "torch" not in out and "oracle" not in out and "libref" not in out


def test_key_pack_orders_like_distance_then_index(fir):
    rng = np.random.default_rng(0)
    d = np.concatenate([rng.random(200).astype(np.float32), np.array([0.0, -0.0, 1e-38, 99999.99, 3.4e38, -1.5, -1e-30], np.float32)])
    i = rng.integers(0, 2**31 - 1, d.size).astype(np.int32)
    keys = np.array([fir.key_pack(a, b) for a, b in zip(d, i)], np.uint64)
    order = np.argsort(keys, kind="stable")
    exp = np.lexsort((i, d + np.float32(0)))
    assert np.array_equal(d[order], d[exp])
    ui, ud = fir.keys_unpack(keys)
    assert np.array_equal(ui, i)
    assert np.array_equal((ud + np.float32(0)).view(np.uint32), (d + np.float32(0)).view(np.uint32))
    assert fir.key_pack(1.0, -1) == 0xFFFFFFFFFFFFFFFF
    ni, nd = fir.keys_unpack(np.array([0xFFFFFFFFFFFFFFFF], np.uint64))
    assert ni[0] == -1 and nd[0] == np.float32(100000.0)
    # same distance: the lower index wins the integer minimum (first-minimum rule across shards)
    assert fir.key_pack(0.5, 7) < fir.key_pack(0.5, 8) < fir.key_pack(np.nextafter(np.float32(0.5), np.float32(1)), 0)


def test_device_calls_fail_loudly_without_a_gpu(fir):
    if fir.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(fir.FirError):
        fir.Gallery(np.zeros((4, 8), np.float32), None, 0, 0)
    with pytest.raises(fir.FirError):
        fir.feature_distance(np.zeros(8, np.float32), np.zeros(8, np.float32))
    with pytest.raises(fir.FirError):
        fir.ShardedGallery(np.zeros((4, 8), np.float32), None, 0, devices=[0])
