"""ctypes binding of include/fir_amd.h (plumbing only: argument marshalling, error mapping)."""
import ctypes as C
import os

import numpy as np

METRIC_L2, METRIC_CHI2, METRIC_KL = 0, 1, 2
KEY_NONE = 0xFFFFFFFFFFFFFFFF

_HERE = os.path.dirname(os.path.abspath(__file__))


_LIB_FILE = "libfir_amd.so"       # __graft_entry__.load_package(audit=True) points its own copy of this module at libfir_amd_audit.so


def lib_path():
    # FIR_AMD_LIB: an alternative build of the same library (kernel-variant experiments in tools/)
    return os.environ.get("FIR_AMD_LIB") or os.path.join(_HERE, _LIB_FILE)


class FirError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"fir_amd error {code}: {msg}")
        self.code = code


# Every symbol include/fir_amd.h declares: (name, restype, argtypes).
_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_u64p = C.POINTER(C.c_uint64)
_vp = C.c_void_p
SYMBOLS = [
    ("fir_last_error", C.c_char_p, []),
    ("fir_version", C.c_int, []),
    ("fir_device_count", C.c_int, []),
    ("fir_device_info", C.c_int, [C.c_int32, C.c_char_p, C.c_int32, _i32p, _i64p]),
    ("fir_device_peak_hbm_gbs", C.c_int, [C.c_int32, C.POINTER(C.c_double)]),
    ("fir_gallery_create", C.c_int, [_vp, C.c_int64, C.c_int32, _vp, C.c_int32, C.c_int32, C.POINTER(_vp)]),
    ("fir_gallery_create_dev", C.c_int, [_vp, C.c_int64, C.c_int32, _vp, C.c_int32, C.c_int32, _vp, C.POINTER(_vp)]),
    ("fir_gallery_destroy", C.c_int, [_vp]),
    ("fir_gallery_info", C.c_int, [_vp, _i64p, _i32p, _i32p, _i32p]),
    ("fir_gallery_set_metric", C.c_int, [_vp, C.c_int32]),
    ("fir_gallery_set_row_offset", C.c_int, [_vp, C.c_int64]),
    ("fir_gallery_set_large_batch_mfma", C.c_int, [_vp, C.c_int32]),
    ("fir_feature_distance", C.c_int, [_vp, _vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _f32p]),
    ("fir_search_top1", C.c_int, [_vp, _vp, C.c_int32, C.c_int32, C.c_int32, _vp, _vp]),
    ("fir_search_top1_keys_dev", C.c_int, [_vp, _vp, C.c_int32, C.c_int32, C.c_int32, _vp, _vp]),
    ("fir_keys_unpack", C.c_int, [_vp, C.c_int32, _vp, _vp]),
    ("fir_key_pack", C.c_uint64, [C.c_float, C.c_int32]),
    ("fir_gallery_classes_of", C.c_int, [_vp, _vp, C.c_int32, _vp]),
    ("fir_search_topk", C.c_int, [_vp, _vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _vp, _vp]),
    ("fir_search_topk_keys_dev", C.c_int, [_vp, _vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _vp, _vp]),
    ("fir_range_distances", C.c_int, [_vp, _vp, C.c_int32, C.c_int32, C.c_int32, _vp]),
    ("fir_range_distances_dev", C.c_int, [_vp, _vp, C.c_int32, C.c_int32, C.c_int32, _vp, _vp]),
    ("fir_twd_conventional", C.c_int, [_vp, _vp, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_int32, _vp, _vp]),
    ("fir_twd_proposed", C.c_int, [_vp, _vp, C.c_int32, C.c_int32, C.c_double, _vp, _vp, _vp]),
    ("fir_cls_create", C.c_int, [_vp, C.c_int64, C.c_int32, _vp, C.c_int32, _vp, C.c_int32, C.POINTER(_vp)]),
    ("fir_cls_create_dev", C.c_int, [_vp, C.c_int64, C.c_int32, _vp, C.c_int32, _vp, C.c_int32, C.POINTER(_vp)]),
    ("fir_cls_destroy", C.c_int, [_vp]),
    ("fir_cls_create_sharded", C.c_int, [_vp, C.c_int64, C.c_int32, _vp, C.c_int32, _vp, _vp, C.c_int32, _vp, C.POINTER(_vp)]),
    ("fir_cls_sharded_destroy", C.c_int, [_vp]),
    ("fir_cls_sharded_pnn_predict", C.c_int, [_vp, _vp, C.c_int32, C.c_double, _vp, _vp]),
    ("fir_cls_sharded_knn_predict", C.c_int, [_vp, _vp, C.c_int32, C.c_int32, _vp]),
    ("fir_cls_set_total_training_size", C.c_int, [_vp, C.c_int64]),
    ("fir_cls_distance_sums", C.c_int, [_vp, _vp, C.c_int32, _vp]),
    ("fir_cls_pnn_predict", C.c_int, [_vp, _vp, C.c_int32, C.c_double, _vp, _vp]),
    ("fir_cls_pnn_predict_seq", C.c_int, [_vp, _vp, C.c_int32, C.c_double, _vp, _vp]),
    ("fir_cls_knn_predict", C.c_int, [_vp, _vp, C.c_int32, C.c_int32, _vp]),
    ("fir_cls_set_knn_mfma", C.c_int, [_vp, C.c_int32]),
    ("fir_cls_knn_stats", C.c_int, [_vp, _i64p, _i64p]),
    ("fir_cls_last_dispatch", C.c_int, [_vp, C.c_char_p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("fir_cls_knn_class_nearest", C.c_int, [_vp, _vp, C.c_int32, C.c_int32, _vp]),
    ("fir_gemm_create", C.c_int, [_vp, C.POINTER(_vp)]),
    ("fir_gemm_create_ex", C.c_int, [_vp, C.c_int32, C.POINTER(_vp)]),
    ("fir_gemm_destroy", C.c_int, [_vp]),
    ("fir_gemm_create_range", C.c_int, [_vp, C.c_int32, C.c_int32, C.POINTER(_vp)]),
    ("fir_gemm_search_top1_keys_dev", C.c_int, [_vp, _vp, C.c_int32, _vp, _vp]),
    ("fir_gemm_search_topk_keys_dev", C.c_int, [_vp, _vp, C.c_int32, C.c_int32, _vp, _vp]),
    ("fir_gemm_search_few_keys_dev", C.c_int, [_vp, _vp, C.c_int32, _vp, _vp]),
    ("fir_gemm_stats", C.c_int, [_vp, _i64p, _i64p]),
    ("fir_gemm_stats_ex", C.c_int, [_vp, _i64p]),
    ("fir_gemm_uncertified_notes", C.c_int, [_vp, _f32p, _i32p]),
    ("fir_gallery_mfma_uncertified_notes", C.c_int, [_vp, _f32p, _i32p]),
    ("fir_dem_pivot_table", C.c_int, [_vp, C.c_int32, C.c_int32, _vp, _vp, _vp, _i32p]),
    ("fir_dem_create", C.c_int, [_vp, C.c_int32, C.c_int32, C.POINTER(_vp)]),
    ("fir_dem_destroy", C.c_int, [_vp]),
    ("fir_dem_info", C.c_int, [_vp, _i32p, _i32p, _i32p, _i64p]),
    ("fir_dem_get", C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    ("fir_dem_likelihoods", C.c_int, [_vp, _vp, C.c_int32, _vp, _vp]),
    ("fir_rows_distances", C.c_int, [_vp, _vp, C.c_int32, _vp, C.c_int32, C.c_int32, C.c_int32, _vp]),
    ("fir_fpnn_train", C.c_int, [_vp, C.c_int64, C.c_int32, _vp, C.c_int32, _vp, _vp, C.c_double, C.c_int32, C.POINTER(_vp)]),
    ("fir_fpnn_destroy", C.c_int, [_vp]),
    ("fir_fpnn_info", C.c_int, [_vp, _i32p, _i32p, _i32p]),
    ("fir_fpnn_get_model", C.c_int, [_vp, _vp]),
    ("fir_fpnn_predict", C.c_int, [_vp, _vp, C.c_int32, _vp, _vp]),
    ("fir_fpnn_predict_seq", C.c_int, [_vp, _vp, C.c_int32, C.c_float, _vp, _vp]),
    ("fir_comm_unique_id", C.c_int, [_vp]),
    ("fir_gallery_create_sharded", C.c_int, [_vp, C.c_int64, C.c_int32, _vp, C.c_int32, _vp, C.c_int32, C.POINTER(_vp)]),
    ("fir_gallery_create_sharded_ex", C.c_int, [_vp, C.c_int64, C.c_int32, _vp, C.c_int32, _vp, C.c_int32, _vp, C.POINTER(_vp)]),
    ("fir_sharded_destroy", C.c_int, [_vp]),
    ("fir_sharded_info", C.c_int, [_vp, _i64p, _i32p, _i32p, _i32p, _i32p, _i32p]),
    ("fir_sharded_shard", C.c_int, [_vp, C.c_int32, C.POINTER(_vp), _i64p, _i64p]),
    ("fir_sharded_set_metric", C.c_int, [_vp, C.c_int32]),
    ("fir_sharded_search_top1", C.c_int, [_vp, _vp, C.c_int32, C.c_int32, C.c_int32, _vp, _vp]),
    ("fir_sharded_search_topk", C.c_int, [_vp, _vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _vp, _vp]),
    ("fir_sharded_classify_top1", C.c_int, [_vp, _vp, C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp]),
    ("fir_sharded_search_top1_keys_dev", C.c_int, [_vp, _vp, C.c_int32, C.c_int32, C.c_int32, _vp, _vp]),
    ("fir_sharded_sync", C.c_int, [_vp]),
    ("fir_sharded_profile_enable", C.c_int, [_vp, C.c_int32]),
    ("fir_sharded_profile_read", C.c_int, [_vp, _vp, C.c_int32, _i32p]),
    ("fir_profile_enable", C.c_int, [_vp, C.c_int32]),
    ("fir_profile_read", C.c_int, [_vp, _vp, C.c_int32, _i32p, C.POINTER(C.c_double)]),
    ("fir_gallery_last_dispatch", C.c_int, [_vp, _vp]),
    ("fir_gallery_sync", C.c_int, [_vp]),
    ("fir_gallery_set_tuning", C.c_int, [_vp, C.c_int32, C.c_int32]),
    ("fir_gallery_value_range", C.c_int, [_vp, _i32p, _i32p]),
    ("fir_gallery_get_tuning", C.c_int, [_vp, _i32p, _i32p, _i32p]),
    ("fir_cls_profile_enable", C.c_int, [_vp, C.c_int32]),
    ("fir_cls_profile_read", C.c_int, [_vp, _vp, C.c_int32, _i32p, C.POINTER(C.c_double), C.c_char_p, C.c_int32]),
    ("fir_gallery_mfma_stats", C.c_int, [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("fir_gallery_mfma_stats_ex", C.c_int, [_vp, C.POINTER(C.c_int64)]),
    ("fir_gallery_memory_bytes", C.c_int, [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("fir_gallery_set_shadow_copies", C.c_int, [_vp, C.c_int32]),
]

_lib = None


def lib():
    """Load libfir_amd.so (once). Raises if it has not been built: there is no fallback."""
    global _lib
    if _lib is None:
        p = lib_path()
        if not os.path.exists(p):
            raise FirError(-100, f"{p} not built (run __graft_entry__.build())")
        L = C.CDLL(p)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise FirError(rc, lib().fir_last_error().decode(errors="replace"))


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_vp)


def device_count():
    return lib().fir_device_count()


def device_info(device=0):
    name = C.create_string_buffer(64)
    cus = C.c_int32()
    hbm = C.c_int64()
    _check(lib().fir_device_info(device, name, 64, C.byref(cus), C.byref(hbm)))
    return {"arch": name.value.decode(), "cus": cus.value, "hbm_bytes": hbm.value}


def device_peak_hbm_gbs(device=0):
    v = C.c_double()
    _check(lib().fir_device_peak_hbm_gbs(device, C.byref(v)))
    return v.value


def feature_distance(lhs, rhs, start=0, end=None, metric=METRIC_L2, device=0):
    lhs, pl = _f32(lhs)
    rhs, pr = _f32(rhs)
    if end is None:
        end = lhs.size
    out = C.c_float()
    _check(lib().fir_feature_distance(pl, pr, lhs.size, start, end, metric, device, C.byref(out)))
    return np.float32(out.value)


def key_pack(dist, idx):
    return int(lib().fir_key_pack(float(dist), int(idx)))


def keys_unpack(keys):
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    idx = np.empty(keys.shape, np.int32)
    dist = np.empty(keys.shape, np.float32)
    _check(lib().fir_keys_unpack(keys.ctypes.data_as(_vp), keys.size, idx.ctypes.data_as(_vp), dist.ctypes.data_as(_vp)))
    return idx, dist


class DispatchInfo(C.Structure):
    _fields_ = [("struct_bytes", C.c_int32), ("path", C.c_int32), ("kernel", C.c_char * 160), ("launches", C.c_int32), ("grid_x", C.c_int32),
                ("grid_y", C.c_int32), ("block", C.c_int32), ("lds_bytes", C.c_int32), ("vgprs", C.c_int32), ("queries_per_pass", C.c_int32),
                ("bytes_per_launch", C.c_double), ("flops_per_launch", C.c_double), ("warmup_calls_left", C.c_int32), ("reserved", C.c_int32),
                ("knobs", C.c_char * 160)]


SHADOW_NONE, SHADOW_FP16, SHADOW_ALL = 0, 1, 2


class Gallery:
    """Owns one fir_gallery handle. Host arrays in, host arrays out, unless the *_dev methods
    are used with raw device pointers (ints, e.g. torch.Tensor.data_ptr())."""

    def __init__(self, rows=None, class_no=None, metric=METRIC_L2, device=0, *, dev_ptr=None, n=None, d=None,
                 dev_class_ptr=None, stream=None):
        self._h = _vp()
        if dev_ptr is not None:
            _check(lib().fir_gallery_create_dev(_vp(dev_ptr), n, d, _vp(dev_class_ptr) if dev_class_ptr else None, metric,
                                                device, _vp(stream) if stream else None, C.byref(self._h)))
            self.n, self.d = int(n), int(d)
        else:
            rows = np.ascontiguousarray(rows, dtype=np.float32)
            if rows.ndim != 2:
                raise ValueError("rows must be [n, d]")
            cls_p = None
            if class_no is not None:
                class_no = np.ascontiguousarray(class_no, dtype=np.int32)
                cls_p = class_no.ctypes.data_as(_vp)
            _check(lib().fir_gallery_create(rows.ctypes.data_as(_vp), rows.shape[0], rows.shape[1], cls_p, metric, device,
                                            C.byref(self._h)))
            self.n, self.d = rows.shape
        self.device = device

    def close(self):
        if self._h:
            lib().fir_gallery_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_metric(self, metric):
        _check(lib().fir_gallery_set_metric(self._h, metric))

    def set_row_offset(self, off):
        _check(lib().fir_gallery_set_row_offset(self._h, off))

    def set_large_batch_mfma(self, min_queries):
        _check(lib().fir_gallery_set_large_batch_mfma(self._h, min_queries))

    def set_shadow_copies(self, mode):
        _check(lib().fir_gallery_set_shadow_copies(self._h, mode))

    def mfma_stats(self):
        """passes queued; queries whose first certificate did not hold (second matrix-core pass); queries the exact device scan answered"""
        o = (C.c_int64 * 3)()
        _check(lib().fir_gallery_mfma_stats_ex(self._h, o))
        return {"passes": o[0], "second_pass_queries": o[1], "fallback_queries": o[2]}

    def uncertified_notes(self):
        """the first (up to 8) queries whose first certificate did not hold: [list entries asked for, bound, smallest stored proxy, |q|^2]"""
        o, c = (C.c_float * 32)(), C.c_int32()
        _check(lib().fir_gallery_mfma_uncertified_notes(self._h, o, C.byref(c)))
        return [[float(o[4 * i + j]) for j in range(4)] for i in range(c.value)]

    def memory_bytes(self):
        t, f, r, sc = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        _check(lib().fir_gallery_memory_bytes(self._h, C.byref(t), C.byref(f), C.byref(r), C.byref(sc)))
        return {"tiled": t.value, "fp16_fragments": f.value, "rowmajor_shadow": r.value, "scratch": sc.value}

    def value_range(self):
        """(every gallery value in the plain range, every query value of the last search too) -- chi-square / KL scans
        use the short division sequence when both hold (include/fir_amd.h)."""
        a, b = C.c_int32(), C.c_int32()
        _check(lib().fir_gallery_value_range(self._h, C.byref(a), C.byref(b)))
        return bool(a.value), bool(b.value)

    def set_tuning(self, queries_per_pass=0, waves=0):
        _check(lib().fir_gallery_set_tuning(self._h, queries_per_pass, waves))

    def get_tuning(self):
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        _check(lib().fir_gallery_get_tuning(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {"queries_per_pass": a.value, "waves": b.value, "max_waves": c.value}

    def search_top1(self, queries, start=0, end=0):
        q, pq = _f32(queries)
        q = q.reshape(-1, self.d)
        idx = np.empty(q.shape[0], np.int32)
        dist = np.empty(q.shape[0], np.float32)
        _check(lib().fir_search_top1(self._h, pq, q.shape[0], start, end, idx.ctypes.data_as(_vp), dist.ctypes.data_as(_vp)))
        return idx, dist

    def search_top1_keys_dev(self, q_ptr, qb, keys_ptr, start=0, end=0, stream=None):
        _check(lib().fir_search_top1_keys_dev(self._h, _vp(q_ptr), qb, start, end, _vp(keys_ptr), _vp(stream) if stream else None))

    def search_topk(self, queries, k, start=0, end=0):
        q, pq = _f32(queries)
        q = q.reshape(-1, self.d)
        idx = np.empty((q.shape[0], k), np.int32)
        dist = np.empty((q.shape[0], k), np.float32)
        _check(lib().fir_search_topk(self._h, pq, q.shape[0], start, end, k, idx.ctypes.data_as(_vp), dist.ctypes.data_as(_vp)))
        return idx, dist

    def search_topk_keys_dev(self, q_ptr, qb, k, keys_ptr, start=0, end=0, stream=None):
        _check(lib().fir_search_topk_keys_dev(self._h, _vp(q_ptr), qb, start, end, k, _vp(keys_ptr), _vp(stream) if stream else None))

    def range_distances(self, queries, start=0, end=0):
        q, pq = _f32(queries)
        q = q.reshape(-1, self.d)
        out = np.empty((q.shape[0], self.n), np.float32)
        _check(lib().fir_range_distances(self._h, pq, q.shape[0], start, end, out.ctypes.data_as(_vp)))
        return out

    def rows_distances(self, queries, rows, start=0, end=0):
        """out[q][k] = distance(query q, gallery row rows[q][k]) (fir_rows_distances)."""
        q, pq = _f32(queries)
        q = q.reshape(-1, self.d)
        rows = np.ascontiguousarray(rows, np.int32).reshape(q.shape[0], -1)
        out = np.empty(rows.shape, np.float32)
        _check(lib().fir_rows_distances(self._h, pq, q.shape[0], rows.ctypes.data_as(_vp), rows.shape[1], start, end, out.ctypes.data_as(_vp)))
        return out

    def dem_pivot_table(self, first_pivot, n_pivots, want_table=True):
        """DirectedEnumeration's PIVOT build (ann.cpp:302-331): (pivots, table or None, min_other, n_built)."""
        piv = np.empty(n_pivots, np.int32)
        mo = np.empty(n_pivots, np.float32)
        table = np.empty((n_pivots, self.n), np.float32) if want_table else None
        built = C.c_int32()
        _check(lib().fir_dem_pivot_table(self._h, first_pivot, n_pivots, piv.ctypes.data_as(_vp),
                                         table.ctypes.data_as(_vp) if want_table else None, mo.ctypes.data_as(_vp), C.byref(built)))
        return piv, table, mo, built.value

    def range_distances_dev(self, q_ptr, qb, out_ptr, start=0, end=0, stream=None):
        _check(lib().fir_range_distances_dev(self._h, _vp(q_ptr), qb, start, end, _vp(out_ptr), _vp(stream) if stream else None))

    def twd_conventional(self, queries, num_classes, typ, threshold, reduced_features_count=64):
        q, pq = _f32(queries)
        q = q.reshape(-1, self.d)
        cls = np.empty(q.shape[0], np.int32)
        unrel = np.empty(q.shape[0], np.int32)
        _check(lib().fir_twd_conventional(self._h, pq, q.shape[0], num_classes, typ, threshold, reduced_features_count,
                                          cls.ctypes.data_as(_vp), unrel.ctypes.data_as(_vp)))
        return cls, unrel

    def twd_proposed(self, queries, reduced_features_count, threshold):
        q, pq = _f32(queries)
        q = q.reshape(-1, self.d)
        cls = np.empty(q.shape[0], np.int32)
        unrel = np.empty(q.shape[0], np.int32)
        chunks = np.empty(q.shape[0], np.int32)
        _check(lib().fir_twd_proposed(self._h, pq, q.shape[0], reduced_features_count, threshold, cls.ctypes.data_as(_vp),
                                      unrel.ctypes.data_as(_vp), chunks.ctypes.data_as(_vp)))
        return cls, unrel, chunks

    def classes_of(self, idx):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        out = np.empty(idx.shape, np.int32)
        _check(lib().fir_gallery_classes_of(self._h, idx.ctypes.data_as(_vp), idx.size, out.ctypes.data_as(_vp)))
        return out

    def profile_enable(self, on=True):
        _check(lib().fir_profile_enable(self._h, 1 if on else 0))

    def profile_read(self, cap=65536):
        ms = np.empty(cap, np.float32)
        cnt = C.c_int32()
        nbytes = C.c_double()
        _check(lib().fir_profile_read(self._h, ms.ctypes.data_as(_vp), cap, C.byref(cnt), C.byref(nbytes)))
        return ms[: min(cnt.value, cap)].copy(), nbytes.value

    def sync(self):
        _check(lib().fir_gallery_sync(self._h))

    def last_dispatch(self):
        """The dominant kernel of the most recent top-1 search as the library launched it (fir_gallery_last_dispatch)."""
        o = DispatchInfo()
        o.struct_bytes = C.sizeof(DispatchInfo)
        _check(lib().fir_gallery_last_dispatch(self._h, C.byref(o)))
        return {"path": "mfma" if o.path == 1 else "scan", "kernel": o.kernel.decode(), "launches": o.launches, "grid": [o.grid_x, o.grid_y],
                "block": o.block, "lds_bytes": o.lds_bytes, "vgprs": o.vgprs, "queries_per_pass": o.queries_per_pass,
                "bytes_per_launch": o.bytes_per_launch, "flops_per_launch": o.flops_per_launch, "warmup_calls_left": o.warmup_calls_left,
                "knobs": o.knobs.decode()}


COMM_ID_BYTES = 128


def comm_unique_id():
    """RCCL unique id (bytes) made by process 0 of a multi-process sharded gallery; hand it to the other processes."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _check(lib().fir_comm_unique_id(buf))
    return buf.raw


class ShardOpts(C.Structure):
    _fields_ = [("struct_bytes", C.c_int32), ("shards_per_device", C.c_int32), ("first_global_row", C.c_int64), ("comm_id", _vp),
                ("proc_rank", C.c_int32), ("nprocs", C.c_int32), ("rows_on_device", C.c_int32), ("reserved", C.c_int32), ("total_rows", C.c_int64),
                ("timeout_ms", C.c_int32), ("fail_shard", C.c_int32), ("fail_step", C.c_int32), ("reserved2", C.c_int32)]


class _BorrowedGallery(Gallery):
    """A shard's fir_gallery handle, owned by the ShardedGallery it came from."""

    def __init__(self, handle, n, d, device):  # noqa: super().__init__ not called on purpose
        self._h = handle
        self.n, self.d, self.device = n, d, device

    def close(self):
        self._h = _vp()


class ShardedGallery:
    """Owns one fir_sharded handle: a gallery split by rows over `devices` (x shards_per_device logical shards each);
    the ranks' keys are reduced by RCCL inside the library (include/fir_amd.h)."""

    def __init__(self, rows=None, class_no=None, metric=METRIC_L2, devices=(0,), shards_per_device=1, *, first_global_row=0,
                 comm_id=None, proc_rank=0, nprocs=1, dev_ptr=None, n=None, d=None, dev_class_ptr=None, timeout_ms=0, fail_shard=0,
                 fail_step=0):
        self._h = _vp()
        devs = np.ascontiguousarray(list(devices), dtype=np.int32)
        o = ShardOpts()
        o.struct_bytes = C.sizeof(ShardOpts)
        o.shards_per_device = shards_per_device
        o.first_global_row = first_global_row
        self._id = C.create_string_buffer(comm_id, COMM_ID_BYTES) if comm_id is not None else None
        o.comm_id = C.cast(self._id, _vp) if self._id is not None else None
        o.proc_rank, o.nprocs = proc_rank, nprocs
        o.timeout_ms, o.fail_shard, o.fail_step = timeout_ms, fail_shard, fail_step
        if dev_ptr is not None:
            o.rows_on_device = 1
            rp, cp = _vp(dev_ptr), (_vp(dev_class_ptr) if dev_class_ptr else None)
            self.n, self.d = int(n), int(d)
        else:
            rows = np.ascontiguousarray(rows, dtype=np.float32)
            if rows.ndim != 2:
                raise ValueError("rows must be [n, d]")
            self.n, self.d = rows.shape
            rp = rows.ctypes.data_as(_vp)
            cp = None
            if class_no is not None:
                class_no = np.ascontiguousarray(class_no, dtype=np.int32)
                cp = class_no.ctypes.data_as(_vp)
        _check(lib().fir_gallery_create_sharded_ex(rp, self.n, self.d, cp, metric, devs.ctypes.data_as(_vp), devs.size, C.byref(o),
                                                   C.byref(self._h)))
        self.devices = [int(v) for v in devs]

    def close(self):
        if self._h:
            lib().fir_sharded_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def info(self):
        n, d, nd, ns, nr, fr = C.c_int64(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        _check(lib().fir_sharded_info(self._h, C.byref(n), C.byref(d), C.byref(nd), C.byref(ns), C.byref(nr), C.byref(fr)))
        return {"n_local": n.value, "d": d.value, "ndev": nd.value, "nshards": ns.value, "nranks": nr.value, "first_rank": fr.value}

    def shard(self, i):
        """(borrowed Gallery or None, first global row, rows) of local shard i."""
        g, lo, rows = _vp(), C.c_int64(), C.c_int64()
        _check(lib().fir_sharded_shard(self._h, i, C.byref(g), C.byref(lo), C.byref(rows)))
        return (_BorrowedGallery(g, rows.value, self.d, None) if g else None), lo.value, rows.value

    def set_metric(self, metric):
        _check(lib().fir_sharded_set_metric(self._h, metric))

    def search_top1(self, queries, start=0, end=0):
        q, pq = _f32(queries)
        q = q.reshape(-1, self.d)
        idx = np.empty(q.shape[0], np.int32)
        dist = np.empty(q.shape[0], np.float32)
        _check(lib().fir_sharded_search_top1(self._h, pq, q.shape[0], start, end, idx.ctypes.data_as(_vp), dist.ctypes.data_as(_vp)))
        return idx, dist

    def search_topk(self, queries, k, start=0, end=0):
        q, pq = _f32(queries)
        q = q.reshape(-1, self.d)
        idx = np.empty((q.shape[0], k), np.int32)
        dist = np.empty((q.shape[0], k), np.float32)
        _check(lib().fir_sharded_search_topk(self._h, pq, q.shape[0], start, end, k, idx.ctypes.data_as(_vp), dist.ctypes.data_as(_vp)))
        return idx, dist

    def classify_top1(self, queries, start=0, end=0):
        q, pq = _f32(queries)
        q = q.reshape(-1, self.d)
        cls = np.empty(q.shape[0], np.int32)
        idx = np.empty(q.shape[0], np.int32)
        dist = np.empty(q.shape[0], np.float32)
        _check(lib().fir_sharded_classify_top1(self._h, pq, q.shape[0], start, end, cls.ctypes.data_as(_vp), idx.ctypes.data_as(_vp),
                                               dist.ctypes.data_as(_vp)))
        return cls, idx, dist

    def search_top1_keys_dev(self, q_ptr, qb, keys_ptr, start=0, end=0, stream=None):
        _check(lib().fir_sharded_search_top1_keys_dev(self._h, _vp(q_ptr), qb, start, end, _vp(keys_ptr), _vp(stream) if stream else None))

    def sync(self):
        _check(lib().fir_sharded_sync(self._h))

    def profile_enable(self, on=True):
        _check(lib().fir_sharded_profile_enable(self._h, 1 if on else 0))

    def profile_read(self, cap=65536):
        ms = np.empty(cap, np.float32)
        cnt = C.c_int32()
        _check(lib().fir_sharded_profile_read(self._h, ms.ctypes.data_as(_vp), cap, C.byref(cnt)))
        return ms[: min(cnt.value, cap)].copy()


class GemmSearch:
    """Large-batch L2 top-1 through the matrix cores (fir_gemm_*): same answers as Gallery.search_top1."""

    F32, BF16_SPLIT, F16 = 0, 1, 2

    def __init__(self, gallery, precision=2, end=0):
        self._g = gallery           # keeps the gallery alive
        self._h = _vp()
        _check(lib().fir_gemm_create_range(gallery._h, precision, end, C.byref(self._h)))   # end: a feature prefix [0, end), 0 = whole rows

    def close(self):
        if self._h:
            lib().fir_gemm_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def search_top1_keys_dev(self, q_ptr, qb, keys_ptr, stream=None):
        _check(lib().fir_gemm_search_top1_keys_dev(self._h, _vp(q_ptr), qb, _vp(keys_ptr), _vp(stream) if stream else None))

    def search_few_keys_dev(self, q_ptr, qb, keys_ptr, stream=None):
        _check(lib().fir_gemm_search_few_keys_dev(self._h, _vp(q_ptr), qb, _vp(keys_ptr), _vp(stream) if stream else None))

    def search_topk_keys_dev(self, q_ptr, qb, k, keys_ptr, stream=None):
        _check(lib().fir_gemm_search_topk_keys_dev(self._h, _vp(q_ptr), qb, k, _vp(keys_ptr), _vp(stream) if stream else None))

    def stats(self):
        o = (C.c_int64 * 3)()
        _check(lib().fir_gemm_stats_ex(self._h, o))
        return {"passes": o[0], "second_pass_queries": o[1], "fallback_queries": o[2]}


class Dem:
    """DirectedEnumeration's device state (fir_dem_*): pivot table + kept pivots, likelihoods at query time."""

    def __init__(self, gallery, first_pivot, n_pivots):
        self._g = gallery
        self._h = _vp()
        _check(lib().fir_dem_create(gallery._h, first_pivot, n_pivots, C.byref(self._h)))
        a, b, c, n = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64()
        _check(lib().fir_dem_info(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(n)))
        self.n_pivots, self.n_built, self.n_used, self.n = a.value, b.value, c.value, n.value

    def close(self):
        if self._h:
            lib().fir_dem_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def get(self, want_table=True):
        piv = np.empty(self.n_pivots, np.int32)
        mo = np.empty(self.n_pivots, np.float32)
        table = np.empty((self.n_used, self.n), np.float32) if want_table else None
        order = np.empty(self.n, np.int32)
        _check(lib().fir_dem_get(self._h, piv.ctypes.data_as(_vp), mo.ctypes.data_as(_vp),
                                 table.ctypes.data_as(_vp) if want_table else None, order.ctypes.data_as(_vp)))
        return piv, mo, table, order

    def likelihoods(self, queries, want_lik=True):
        q, pq = _f32(queries)
        q = q.reshape(-1, self._g.d)
        pd = np.empty((q.shape[0], self.n_used), np.float32)
        lik = np.empty((q.shape[0], self.n), np.float32) if want_lik else None
        _check(lib().fir_dem_likelihoods(self._h, pq, q.shape[0], pd.ctypes.data_as(_vp), lik.ctypes.data_as(_vp) if want_lik else None))
        return pd, lik


class Fpnn:
    """FPNNClassifier (classification.cpp:618-791): trained trigonometric-series model on the device."""

    def __init__(self, train_rows, train_class, num_classes, avg, sd, scale=1.0, device=0):
        rows = np.ascontiguousarray(train_rows, np.float64)
        cls = np.ascontiguousarray(train_class, np.int32)
        avg = np.ascontiguousarray(avg, np.float64)
        sd = np.ascontiguousarray(sd, np.float64)
        self.d, self.num_classes = rows.shape[1], num_classes
        self._h = _vp()
        _check(lib().fir_fpnn_train(rows.ctypes.data_as(_vp), rows.shape[0], self.d, cls.ctypes.data_as(_vp), num_classes,
                                    avg.ctypes.data_as(_vp), sd.ctypes.data_as(_vp), scale, device, C.byref(self._h)))
        j = C.c_int32()
        _check(lib().fir_fpnn_info(self._h, C.byref(j), None, None))
        self.J = j.value

    def close(self):
        if self._h:
            lib().fir_fpnn_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def model(self):
        a = np.empty(self.d * self.num_classes * (2 * self.J + 1), np.float64)
        _check(lib().fir_fpnn_get_model(self._h, a.ctypes.data_as(_vp)))
        return a

    def predict(self, queries):
        q = np.ascontiguousarray(queries, np.float64).reshape(-1, self.d)
        best = np.empty(q.shape[0], np.int32)
        outs = np.empty((q.shape[0], self.num_classes), np.float32)
        _check(lib().fir_fpnn_predict(self._h, q.ctypes.data_as(_vp), q.shape[0], best.ctypes.data_as(_vp), outs.ctypes.data_as(_vp)))
        return best, outs

    def predict_seq(self, queries, output_ratio=0.9):
        q = np.ascontiguousarray(queries, np.float64).reshape(-1, self.d)
        best = np.empty(q.shape[0], np.int32)
        chunks = np.empty(q.shape[0], np.int32)
        _check(lib().fir_fpnn_predict_seq(self._h, q.ctypes.data_as(_vp), q.shape[0], output_ratio, best.ctypes.data_as(_vp),
                                          chunks.ctypes.data_as(_vp)))
        return best, chunks


class ClsModel:
    """Owns one fir_cls handle: the training set of the double-precision kNN / PNN classifiers
    (qt_cpp/classification.cpp:116-226) in the reference's class-major scan order."""

    def __init__(self, train_rows, train_class, num_classes, avg, device=0, *, dev_ptr=None, nt=None, d=None):
        tc = np.ascontiguousarray(train_class, dtype=np.int32)
        av = np.ascontiguousarray(avg, dtype=np.float64)
        self._h = _vp()
        self.num_classes = int(num_classes)
        if dev_ptr is not None:          # rows already in the device's memory (float64, row-major)
            self.nt, self.d = int(nt), int(d)
            _check(lib().fir_cls_create_dev(_vp(dev_ptr), self.nt, self.d, tc.ctypes.data_as(_vp), num_classes, av.ctypes.data_as(_vp), device,
                                            C.byref(self._h)))
            return
        tr = np.ascontiguousarray(train_rows, dtype=np.float64)
        self.nt, self.d = tr.shape
        _check(lib().fir_cls_create(tr.ctypes.data_as(_vp), tr.shape[0], tr.shape[1], tc.ctypes.data_as(_vp), num_classes,
                                    av.ctypes.data_as(_vp), device, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().fir_cls_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _q(self, queries):
        q = np.ascontiguousarray(queries, dtype=np.float64).reshape(-1, self.d)
        return q, q.ctypes.data_as(_vp)

    def profile_enable(self, on=True):
        _check(lib().fir_cls_profile_enable(self._h, 1 if on else 0))

    def profile_read(self, cap=4096):
        ms = np.empty(cap, np.float32)
        cnt, nb = C.c_int32(), C.c_double()
        name = C.create_string_buffer(64)
        _check(lib().fir_cls_profile_read(self._h, ms.ctypes.data_as(_vp), cap, C.byref(cnt), C.byref(nb), name, 64))
        return ms[: min(cnt.value, cap)].copy(), nb.value, name.value.decode()

    def set_total_training_size(self, total):
        _check(lib().fir_cls_set_total_training_size(self._h, int(total)))

    def distance_sums(self, queries):
        q, pq = self._q(queries)
        out = np.empty((q.shape[0], self.nt), np.float64)
        _check(lib().fir_cls_distance_sums(self._h, pq, q.shape[0], out.ctypes.data_as(_vp)))
        return out

    def pnn_predict(self, queries, var=0.0):
        q, pq = self._q(queries)
        scores = np.empty((q.shape[0], self.num_classes), np.float64)
        best = np.empty(q.shape[0], np.int32)
        _check(lib().fir_cls_pnn_predict(self._h, pq, q.shape[0], var, scores.ctypes.data_as(_vp), best.ctypes.data_as(_vp)))
        return best, scores

    def pnn_predict_seq(self, queries, var=0.0):
        q, pq = self._q(queries)
        best = np.empty(q.shape[0], np.int32)
        chunks = np.empty(q.shape[0], np.int32)
        _check(lib().fir_cls_pnn_predict_seq(self._h, pq, q.shape[0], var, best.ctypes.data_as(_vp), chunks.ctypes.data_as(_vp)))
        return best, chunks

    def set_knn_mfma(self, min_queries):
        _check(lib().fir_cls_set_knn_mfma(self._h, min_queries))

    def knn_stats(self):
        a, b = C.c_int64(), C.c_int64()
        _check(lib().fir_cls_knn_stats(self._h, C.byref(a), C.byref(b)))
        return {"matrix_core_queries": a.value, "exact_scan_queries_of_them": b.value}

    def last_dispatch(self):
        buf = C.create_string_buffer(96)
        nb, fl = C.c_double(), C.c_double()
        _check(lib().fir_cls_last_dispatch(self._h, buf, 96, C.byref(nb), C.byref(fl)))
        return {"kernel": buf.value.decode(), "bytes_per_launch": nb.value, "flops_per_launch": fl.value}

    def knn_class_nearest(self, queries, k):
        """[qb, num_classes, k] smallest mean distances per class among the rows held (sharded kNN vote)."""
        q, pq = self._q(queries)
        out = np.empty((q.shape[0], self.num_classes, k), np.float64)
        _check(lib().fir_cls_knn_class_nearest(self._h, pq, q.shape[0], k, out.ctypes.data_as(_vp)))
        return out

    def knn_predict(self, queries, k):
        q, pq = self._q(queries)
        best = np.empty(q.shape[0], np.int32)
        _check(lib().fir_cls_knn_predict(self._h, pq, q.shape[0], k, best.ctypes.data_as(_vp)))
        return best


class ShardedClsModel:
    """Owns one fir_cls_sharded handle: the PNN training set split by rows over `devices` (x shards_per_device), class
    scores added across shards on the device and across ranks by RCCL (ncclAllReduce(ncclSum, ncclDouble))."""

    def __init__(self, train_rows, train_class, num_classes, avg, devices=(0,), shards_per_device=1, *, timeout_ms=0, fail_shard=0, fail_step=0):
        tr = np.ascontiguousarray(train_rows, dtype=np.float64)
        tc = np.ascontiguousarray(train_class, dtype=np.int32)
        av = np.ascontiguousarray(avg, dtype=np.float64)
        devs = np.ascontiguousarray(list(devices), dtype=np.int32)
        o = ShardOpts()
        o.struct_bytes = C.sizeof(ShardOpts)
        o.shards_per_device = shards_per_device
        o.timeout_ms, o.fail_shard, o.fail_step = timeout_ms, fail_shard, fail_step
        self._h = _vp()
        self.nt, self.d = tr.shape
        self.num_classes = int(num_classes)
        _check(lib().fir_cls_create_sharded(tr.ctypes.data_as(_vp), tr.shape[0], tr.shape[1], tc.ctypes.data_as(_vp), num_classes, av.ctypes.data_as(_vp),
                                            devs.ctypes.data_as(_vp), devs.size, C.byref(o), C.byref(self._h)))

    def close(self):
        if self._h:
            lib().fir_cls_sharded_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def pnn_predict(self, queries, var=0.0):
        q = np.ascontiguousarray(queries, dtype=np.float64).reshape(-1, self.d)
        scores = np.empty((q.shape[0], self.num_classes), np.float64)
        best = np.empty(q.shape[0], np.int32)
        _check(lib().fir_cls_sharded_pnn_predict(self._h, q.ctypes.data_as(_vp), q.shape[0], var, scores.ctypes.data_as(_vp), best.ctypes.data_as(_vp)))
        return best, scores

    def knn_predict(self, queries, k):
        q = np.ascontiguousarray(queries, dtype=np.float64).reshape(-1, self.d)
        best = np.empty(q.shape[0], np.int32)
        _check(lib().fir_cls_sharded_knn_predict(self._h, q.ctypes.data_as(_vp), q.shape[0], k, best.ctypes.data_as(_vp)))
        return best
