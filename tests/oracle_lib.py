"""ctypes bindings of the CHECKERS: oracle/liboracle.so (C restatement) and oracle/_ref/*.so
(the real reference slices). Test infrastructure only -- the product never imports this."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_vp = C.c_void_p
_f32p = C.POINTER(C.c_float)


def _p(a):
    return a.ctypes.data_as(_vp)


def _load_videos(call, d):
    names = C.create_string_buffer(1 << 16)
    nv, nf = C.c_int(), C.c_int()
    npersons = call(names, len(names), None, None, None, C.byref(nv), C.byref(nf))
    vpp = np.empty(npersons, np.int32)
    fpv = np.empty(nv.value, np.int32)
    rows = np.empty((nf.value, d), np.float32)
    call(names, len(names), _p(vpp), _p(fpv), _p(rows), C.byref(nv), C.byref(nf))
    return names.value.decode().split("\n")[:-1], vpp, fpv, rows


class Oracle:
    """oracle/oracle.c"""

    def __init__(self, path):
        L = C.CDLL(path)
        self.L = L
        L.orc_feature_distance.restype = C.c_float
        L.orc_feature_distance.argtypes = [_vp, _vp, C.c_int, C.c_int, C.c_int]
        L.orc_all_distances.restype = None
        L.orc_all_distances.argtypes = [_vp, C.c_int64, C.c_int, _vp, C.c_int, C.c_int, C.c_int, _vp]
        L.orc_recognize_bf.restype = C.c_int64
        L.orc_recognize_bf.argtypes = [_vp, C.c_int64, C.c_int, _vp, C.c_int, C.c_int, C.c_int, _f32p]
        L.orc_topk.restype = None
        L.orc_topk.argtypes = [_vp, C.c_int64, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]
        L.orc_bf_classifier.restype = C.c_int
        L.orc_bf_classifier.argtypes = [_vp, C.c_int64, C.c_int, _vp, _vp, C.c_int, C.c_int]
        L.orc_twd_conventional.restype = C.c_int
        L.orc_twd_conventional.argtypes = [_vp, C.c_int64, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int,
                                           C.POINTER(C.c_int)]
        L.orc_twd_proposed.restype = C.c_int
        L.orc_twd_proposed.argtypes = [_vp, C.c_int64, C.c_int, _vp, _vp, C.c_int, C.c_double, C.c_int, C.POINTER(C.c_int),
                                       C.POINTER(C.c_int)]
        L.orc_knn_predict.restype = C.c_int
        L.orc_knn_predict.argtypes = [_vp, _vp, C.c_int64, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp]
        L.orc_pnn_predict.restype = C.c_int
        L.orc_pnn_predict.argtypes = [_vp, _vp, C.c_int64, C.c_int, _vp, C.c_int, _vp, _vp]
        L.orc_pnn_predict_seq.restype = C.c_int
        L.orc_pnn_predict_seq.argtypes = [_vp, _vp, C.c_int64, C.c_int, _vp, C.c_int, _vp, C.POINTER(C.c_int)]
        L.orc_pnn_cluster_class.restype = C.c_int
        L.orc_pnn_cluster_class.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp]
        L.orc_pnn_predict_den.restype = C.c_int
        L.orc_pnn_predict_den.argtypes = [_vp, _vp, C.c_int64, C.c_int, _vp, C.c_int, _vp, C.c_double, _vp]
        L.orc_train_stats.restype = None
        L.orc_train_stats.argtypes = [_vp, C.c_int64, C.c_int, _vp, _vp, _vp, _vp]
        L.orc_load_images.restype = C.c_int64
        L.orc_load_images.argtypes = [C.c_char_p, C.c_int, C.c_int, _vp, _vp, C.c_int64, C.POINTER(C.c_int)]
        L.orc_load_dataset_f64.restype = C.c_int64
        L.orc_load_dataset_f64.argtypes = [C.c_char_p, C.c_int, _vp, _vp, C.c_int64, C.POINTER(C.c_int)]
        L.orc_split.restype = C.c_int64
        L.orc_split.argtypes = [_vp, C.c_int, _vp, C.c_int, C.c_double, _vp, _vp, _vp, _vp, C.POINTER(C.c_int64)]
        L.orc_dem_pivot_table.restype = None
        L.orc_dem_pivot_table.argtypes = [_vp, C.c_int64, C.c_int, _vp, C.c_int, C.c_int, _vp, _vp, _vp]
        L.orc_dem_recognize.restype = C.c_int
        L.orc_dem_recognize.argtypes = [_vp, C.c_int64, C.c_int, C.c_int, _vp, C.c_int, _vp, C.c_float, C.c_int, _vp,
                                        _f32p, C.POINTER(C.c_int), C.POINTER(C.c_int), _vp]
        L.orc_fastlog.restype = C.c_float
        L.orc_fastlog.argtypes = [C.c_float]
        L.orc_fpnn_J.restype = C.c_int
        L.orc_fpnn_J.argtypes = [C.c_int64, C.c_int]
        L.orc_fpnn_train.restype = None
        L.orc_fpnn_train.argtypes = [_vp, _vp, C.c_int64, C.c_int, C.c_int, _vp, _vp, C.c_double, C.c_int, _vp]
        L.orc_fpnn_predict.restype = C.c_int
        L.orc_fpnn_predict.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_double, _vp, C.c_int, C.c_float, _vp, C.POINTER(C.c_int)]
        L.orc_load_videos.restype = C.c_int
        L.orc_load_videos.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_int, _vp, _vp, _vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_get_threshold.restype = C.c_float
        L.orc_get_threshold.argtypes = [_vp, C.c_int, C.c_float]

    @staticmethod
    def _rows(rows):
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        return rows, rows.shape[0], rows.shape[1]

    def feature_distance(self, a, b, start, end, metric=0):
        a = np.ascontiguousarray(a, np.float32)
        b = np.ascontiguousarray(b, np.float32)
        return np.float32(self.L.orc_feature_distance(_p(a), _p(b), start, end, metric))

    def all_distances(self, rows, q, start, end, metric=0):
        rows, n, d = self._rows(rows)
        q = np.ascontiguousarray(q, np.float32)
        out = np.empty(n, np.float32)
        self.L.orc_all_distances(_p(rows), n, d, _p(q), start, end, metric, _p(out))
        return out

    def recognize_bf(self, rows, q, start, end, metric=0):
        rows, n, d = self._rows(rows)
        q = np.ascontiguousarray(q, np.float32)
        bd = C.c_float()
        i = self.L.orc_recognize_bf(_p(rows), n, d, _p(q), start, end, metric, C.byref(bd))
        return int(i), np.float32(bd.value)

    def top1_batch_omp(self, rows, queries, start, end, metric=0, threads=0):
        """recognize_image_bf for every query, queries spread over all host cores (OpenMP); returns (idx, dist, threads)."""
        rows, n, d = self._rows(rows)
        queries = np.ascontiguousarray(queries, np.float32).reshape(-1, d)
        idx = np.empty(queries.shape[0], np.int64)
        dist = np.empty(queries.shape[0], np.float32)
        fn = self.L.orc_recognize_bf_batch_omp
        fn.restype = C.c_int
        fn.argtypes = [_vp, C.c_int64, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]
        threads = fn(_p(rows), n, d, _p(queries), queries.shape[0], start, end, metric, threads, _p(idx), _p(dist))
        return idx.astype(np.int32), dist, int(threads)

    def top1_batch(self, rows, queries, start, end, metric=0):
        queries = np.ascontiguousarray(queries, np.float32).reshape(-1, rows.shape[1])
        idx = np.empty(queries.shape[0], np.int32)
        dist = np.empty(queries.shape[0], np.float32)
        for i, q in enumerate(queries):
            idx[i], dist[i] = self.recognize_bf(rows, q, start, end, metric)
        return idx, dist

    def topk(self, rows, q, start, end, k, metric=0):
        rows, n, d = self._rows(rows)
        q = np.ascontiguousarray(q, np.float32)
        idx = np.empty(k, np.int64)
        dist = np.empty(k, np.float32)
        self.L.orc_topk(_p(rows), n, d, _p(q), start, end, metric, k, _p(idx), _p(dist))
        return idx.astype(np.int32), dist

    def bf_classifier(self, rows, cls, q, max_features, metric=0):
        rows, n, d = self._rows(rows)
        cls = np.ascontiguousarray(cls, np.int32)
        q = np.ascontiguousarray(q, np.float32)
        return int(self.L.orc_bf_classifier(_p(rows), n, d, _p(cls), _p(q), max_features, metric))

    def twd_conventional(self, rows, cls, q, num_classes, typ, threshold, feat_count=64, metric=0):
        rows, n, d = self._rows(rows)
        cls = np.ascontiguousarray(cls, np.int32)
        q = np.ascontiguousarray(q, np.float32)
        unrel = C.c_int()
        r = self.L.orc_twd_conventional(_p(rows), n, d, _p(cls), _p(q), num_classes, typ, threshold, feat_count, metric, C.byref(unrel))
        return int(r), int(unrel.value)

    def twd_proposed(self, rows, cls, q, feat_count, th, metric=0):
        rows, n, d = self._rows(rows)
        cls = np.ascontiguousarray(cls, np.int32)
        q = np.ascontiguousarray(q, np.float32)
        unrel = C.c_int()
        chunks = C.c_int()
        r = self.L.orc_twd_proposed(_p(rows), n, d, _p(cls), _p(q), feat_count, th, metric, C.byref(unrel), C.byref(chunks))
        return int(r), int(unrel.value), int(chunks.value)

    def knn_predict(self, train_rows, train_class, avg, num_classes, q, K):
        tr = np.ascontiguousarray(train_rows, np.float64)
        tc = np.ascontiguousarray(train_class, np.int32)
        avg = np.ascontiguousarray(avg, np.float64)
        q = np.ascontiguousarray(q, np.float64)
        dist = np.empty(tr.shape[0], np.float64)
        r = self.L.orc_knn_predict(_p(tr), _p(tc), tr.shape[0], tr.shape[1], _p(avg), num_classes, _p(q), K, _p(dist))
        return int(r), dist

    def pnn_predict(self, train_rows, train_class, avg, num_classes, q):
        tr = np.ascontiguousarray(train_rows, np.float64)
        tc = np.ascontiguousarray(train_class, np.int32)
        avg = np.ascontiguousarray(avg, np.float64)
        q = np.ascontiguousarray(q, np.float64)
        scores = np.empty(num_classes, np.float64)
        r = self.L.orc_pnn_predict(_p(tr), _p(tc), tr.shape[0], tr.shape[1], _p(avg), num_classes, _p(q), _p(scores))
        return int(r), scores

    def pnn_predict_seq(self, train_rows, train_class, avg, num_classes, q):
        tr = np.ascontiguousarray(train_rows, np.float64)
        tc = np.ascontiguousarray(train_class, np.int32)
        avg = np.ascontiguousarray(avg, np.float64)
        q = np.ascontiguousarray(q, np.float64)
        chunks = C.c_int()
        r = self.L.orc_pnn_predict_seq(_p(tr), _p(tc), tr.shape[0], tr.shape[1], _p(avg), num_classes, _p(q), C.byref(chunks))
        return int(r), int(chunks.value)

    def pnn_cluster_class(self, rows, num_clusters):
        r = np.ascontiguousarray(rows, np.float64)
        out = np.full(max(num_clusters, r.shape[0]), -1, np.int32)
        m = self.L.orc_pnn_cluster_class(_p(r), r.shape[0], r.shape[1], num_clusters, _p(out))
        return out[:m].copy()

    def pnn_cluster_train(self, train_rows, train_class, num_classes, num_clusters):
        """medoid rows (positions in train_rows) of every class, class-major."""
        tc = np.asarray(train_class)
        keep = []
        for c in range(num_classes):
            pos = np.nonzero(tc == c)[0]
            if pos.size:
                keep.extend(pos[self.pnn_cluster_class(train_rows[pos], num_clusters)])
        return np.array(keep, np.int64)

    def pnn_predict_den(self, train_rows, train_class, avg, num_classes, q, total):
        tr = np.ascontiguousarray(train_rows, np.float64)
        tc = np.ascontiguousarray(train_class, np.int32)
        avg = np.ascontiguousarray(avg, np.float64)
        q = np.ascontiguousarray(q, np.float64)
        scores = np.empty(num_classes, np.float64)
        r = self.L.orc_pnn_predict_den(_p(tr), _p(tc), tr.shape[0], tr.shape[1], _p(avg), num_classes, _p(q), float(total), _p(scores))
        return int(r), scores

    def train_stats(self, train_rows):
        tr = np.ascontiguousarray(train_rows, np.float64)
        d = tr.shape[1]
        mn, mx, avg, sd = (np.empty(d, np.float64) for _ in range(4))
        self.L.orc_train_stats(_p(tr), tr.shape[0], d, _p(mn), _p(mx), _p(avg), _p(sd))
        return mn, mx, avg, sd

    def load_images(self, path, d, metric=0):
        ncls = C.c_int()
        n = self.L.orc_load_images(path.encode(), d, metric, None, None, 0, C.byref(ncls))
        rows = np.empty((n, d), np.float32)
        cls = np.empty(n, np.int32)
        n2 = self.L.orc_load_images(path.encode(), d, metric, _p(rows), _p(cls), n, C.byref(ncls))
        assert n2 == n
        return rows, cls, int(ncls.value)

    def load_videos(self, path, d, metric=0):
        """loadVideos (video.cpp:35-96) -> (names, videos_per_person, frames_per_video, rows)"""
        return _load_videos(lambda *a: self.L.orc_load_videos(path.encode(), d, metric, *a), d)

    def load_dataset_f64(self, path, d):
        ncls = C.c_int()
        n = self.L.orc_load_dataset_f64(path.encode(), d, None, None, 0, C.byref(ncls))
        rows = np.empty((n, d), np.float64)
        lab = np.empty(n, np.int32)
        n2 = self.L.orc_load_dataset_f64(path.encode(), d, _p(rows), _p(lab), n, C.byref(ncls))
        assert n2 == n
        return rows, lab, int(ncls.value)

    def split(self, class_counts, perm=None, caltech_rule=True, fraction=0.03):
        cc = np.ascontiguousarray(class_counts, np.int32)
        if perm is None:
            perm = np.arange(400, dtype=np.int32)
        perm = np.ascontiguousarray(perm, np.int32)
        tot = int(cc.sum())
        dbi, dbc, ti, tc = (np.empty(tot, np.int32) for _ in range(4))
        nt = C.c_int64()
        ndb = self.L.orc_split(_p(cc), cc.size, _p(perm), 1 if caltech_rule else 0, fraction, _p(dbi), _p(dbc), _p(ti), _p(tc), C.byref(nt))
        return dbi[:ndb].copy(), dbc[:ndb].copy(), ti[: nt.value].copy(), tc[: nt.value].copy()

    def get_threshold(self, dists, rate):
        d = np.ascontiguousarray(dists, np.float32)
        return np.float32(self.L.orc_get_threshold(_p(d), d.size, rate))

    def dem_pivot_table(self, rows, cls, first_pivot, n_pivots, metric=0):
        rows, n, d = self._rows(rows)
        cls = np.ascontiguousarray(cls, np.int32)
        piv = np.full(n_pivots, -1, np.int32)
        piv[0] = first_pivot
        table = np.empty((n_pivots, n), np.float32)
        mo = np.empty(n_pivots, np.float32)
        self.L.orc_dem_pivot_table(_p(rows), n, d, _p(cls), metric, n_pivots, _p(piv), _p(table), _p(mo))
        return piv, table, mo


    def fastlog(self, x):
        return np.float32(self.L.orc_fastlog(float(x)))

    def fpnn_train(self, train_rows, train_class, num_classes, avg, sd, scale):
        """FPNNClassifier::train (classification.cpp:661-696) -> (J, a)"""
        rows = np.ascontiguousarray(train_rows, np.float64)
        cls = np.ascontiguousarray(train_class, np.int32)
        avg = np.ascontiguousarray(avg, np.float64)
        sd = np.ascontiguousarray(sd, np.float64)
        nt, d = rows.shape
        J = self.L.orc_fpnn_J(nt, num_classes)
        a = np.empty(d * num_classes * (2 * J + 1), np.float64)
        self.L.orc_fpnn_train(_p(rows), _p(cls), nt, d, num_classes, _p(avg), _p(sd), scale, J, _p(a))
        return J, a

    def fpnn_predict(self, a, J, num_classes, avg, sd, scale, q, seq=False, output_ratio=0.9):
        """predict_bf / predict_sequentional (:698-791) -> (class, outputs[C], chunks)"""
        a = np.ascontiguousarray(a, np.float64)
        avg = np.ascontiguousarray(avg, np.float64)
        sd = np.ascontiguousarray(sd, np.float64)
        q = np.ascontiguousarray(q, np.float64)
        outs = np.empty(num_classes, np.float32)
        ch = C.c_int()
        r = self.L.orc_fpnn_predict(_p(a), J, q.size, num_classes, _p(avg), _p(sd), scale, _p(q), 1 if seq else 0, output_ratio, _p(outs), C.byref(ch))
        return r, outs, ch.value

    def dem_recognize(self, rows, pivots, table, threshold, image_count, query, metric=0, want_lik=False):
        """-> (row, best_dist, found, calc_count[, likelihoods])"""
        rows, n, d = self._rows(rows)
        pivots = np.ascontiguousarray(pivots, np.int32)
        table = np.ascontiguousarray(table, np.float32)
        q = np.ascontiguousarray(query, np.float32)
        bd, fo, cc = C.c_float(), C.c_int(), C.c_int()
        lik = np.zeros(n, np.float32) if want_lik else None
        r = self.L.orc_dem_recognize(_p(rows), n, d, metric, _p(pivots), pivots.size, _p(table), float(threshold), image_count, _p(q),
                                     C.byref(bd), C.byref(fo), C.byref(cc), _p(lik) if want_lik else None)
        out = (r, np.float32(bd.value), fo.value, cc.value)
        return out + (lik,) if want_lik else out


class RefMatch:
    """oracle/_ref/libref_{l2,chi2,kl}.so -- the reference's own code (oracle/ref_wrap_match.inc)."""

    def __init__(self, path):
        L = C.CDLL(path)
        self.L = L
        L.ref_feature_distance.restype = C.c_float
        L.ref_feature_distance.argtypes = [_vp, _vp, C.c_int, C.c_int, C.c_int]
        L.ref_db_create.restype = _vp
        L.ref_db_create.argtypes = [_vp, C.c_int64, C.c_int, _vp, C.c_int]
        L.ref_db_destroy.argtypes = [_vp]
        L.ref_db_recognize_image_bf.restype = C.c_int
        L.ref_db_recognize_image_bf.argtypes = [_vp, _vp, C.c_int]
        L.ref_db_distance.restype = C.c_float
        L.ref_db_distance.argtypes = [_vp, _vp, C.c_int64, C.c_int, C.c_int]
        L.ref_db_all_distances.argtypes = [_vp, _vp, C.c_int, C.c_int, _vp]
        L.ref_db_bf_classifier.restype = C.c_int
        L.ref_db_bf_classifier.argtypes = [_vp, _vp, C.c_int, C.c_char_p, C.c_int]
        L.ref_db_twd_conventional.restype = C.c_int
        L.ref_db_twd_conventional.argtypes = [_vp, _vp, C.c_int, C.c_int, C.c_double, C.c_int, C.POINTER(C.c_int)]
        L.ref_db_twd_proposed.restype = C.c_int
        L.ref_db_twd_proposed.argtypes = [_vp, _vp, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int)]
        L.ref_db_ann_bruteforce.restype = C.c_int
        L.ref_db_ann_bruteforce.argtypes = [_vp, _vp]
        L.ref_get_threshold.restype = C.c_float
        L.ref_get_threshold.argtypes = [_vp, C.c_int, C.c_float]
        L.ref_dem_create.restype = _vp
        L.ref_dem_create.argtypes = [_vp, C.c_float, C.c_float, C.c_int, C.c_uint]
        L.ref_dem_destroy.argtypes = [_vp]
        L.ref_dem_get.restype = C.c_int
        L.ref_dem_get.argtypes = [_vp, _vp, _vp, _f32p]
        L.ref_dem_set_image_count.argtypes = [_vp, C.c_int]
        L.ref_dem_recognize.restype = C.c_int
        L.ref_dem_recognize.argtypes = [_vp, _vp, _vp, _f32p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.ref_video_features_file.restype = C.c_char_p
        L.ref_load_videos_cwd.restype = C.c_int
        L.ref_load_videos_cwd.argtypes = [C.c_char_p, C.c_int, _vp, _vp, _vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.ref_load_images.restype = C.c_int
        L.ref_load_images.argtypes = [C.c_char_p, _vp, _vp, C.c_int64, C.POINTER(C.c_int)]
        L.ref_split_noshuffle.restype = C.c_int
        L.ref_split_noshuffle.argtypes = [_vp, C.c_int, _vp, _vp, _vp, _vp, C.POINTER(C.c_int)]
        self.features_count = L.ref_features_count()
        self.metric = L.ref_metric()

    def feature_distance(self, a, b, start, end):
        a = np.ascontiguousarray(a, np.float32)
        b = np.ascontiguousarray(b, np.float32)
        return np.float32(self.L.ref_feature_distance(_p(a), _p(b), a.size, start, end))

    def db(self, rows, class_no=None, pad_to=0):
        return RefDb(self, rows, class_no, pad_to)

    def get_threshold(self, dists, rate):
        d = np.array(dists, np.float32)
        return np.float32(self.L.ref_get_threshold(_p(d), d.size, rate))

    def load_images(self, path):
        ncls = C.c_int()
        n = self.L.ref_load_images(path.encode(), None, None, 0, C.byref(ncls))
        rows = np.empty((n, self.features_count), np.float32)
        cls = np.empty(n, np.int32)
        n2 = self.L.ref_load_images(path.encode(), _p(rows), _p(cls), n, C.byref(ncls))
        assert n2 == n
        return rows, cls, int(ncls.value)

    def load_videos_cwd(self):
        """loadVideos over VIDEO_FEATURES_FILE in the current directory."""
        return _load_videos(self.L.ref_load_videos_cwd, self.features_count)

    def video_features_file(self):
        return self.L.ref_video_features_file().decode()

    def split_noshuffle(self, class_counts):
        cc = np.ascontiguousarray(class_counts, np.int32)
        tot = int(cc.sum())
        dbi, dbc, ti, tc = (np.empty(tot, np.int32) for _ in range(4))
        nt = C.c_int()
        ndb = self.L.ref_split_noshuffle(_p(cc), cc.size, _p(dbi), _p(dbc), _p(ti), _p(tc), C.byref(nt))
        return dbi[:ndb].copy(), dbc[:ndb].copy(), ti[: nt.value].copy(), tc[: nt.value].copy()


class RefDb:
    def __init__(self, ref, rows, class_no, pad_to):
        self.ref = ref
        rows = np.ascontiguousarray(rows, np.float32)
        self.n, self.d = rows.shape
        cls = None if class_no is None else np.ascontiguousarray(class_no, np.int32)
        self.h = ref.L.ref_db_create(_p(rows), self.n, self.d, None if cls is None else _p(cls), pad_to)

    def close(self):
        if self.h:
            self.ref.L.ref_db_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def recognize_image_bf(self, q, max_features):
        q = np.ascontiguousarray(q, np.float32)
        return int(self.ref.L.ref_db_recognize_image_bf(self.h, _p(q), max_features))

    def distance(self, q, row, start, end):
        q = np.ascontiguousarray(q, np.float32)
        return np.float32(self.ref.L.ref_db_distance(self.h, _p(q), row, start, end))

    def all_distances(self, q, start, end):
        q = np.ascontiguousarray(q, np.float32)
        out = np.empty(self.n, np.float32)
        self.ref.L.ref_db_all_distances(self.h, _p(q), start, end, _p(out))
        return out

    def bf_classifier(self, q, max_feats):
        q = np.ascontiguousarray(q, np.float32)
        name = C.create_string_buffer(64)
        r = self.ref.L.ref_db_bf_classifier(self.h, _p(q), max_feats, name, 64)
        return int(r), name.value.decode()

    def twd_conventional(self, q, num_classes, typ, threshold, feat_count=64):
        q = np.ascontiguousarray(q, np.float32)
        u = C.c_int()
        r = self.ref.L.ref_db_twd_conventional(self.h, _p(q), num_classes, typ, threshold, feat_count, C.byref(u))
        return int(r), int(u.value)

    def twd_proposed(self, q, num_classes, feat_count, th):
        q = np.ascontiguousarray(q, np.float32)
        u = C.c_int()
        r = self.ref.L.ref_db_twd_proposed(self.h, _p(q), num_classes, feat_count, th, C.byref(u))
        return int(r), int(u.value)

    def dem(self, false_accept_rate=0.01, threshold=0.0, image_count=0, seed=13):
        """The reference's DirectedEnumeration over this gallery (ann.cpp:270-507)."""
        return RefDem(self, false_accept_rate, threshold, image_count, seed)

    def dem_build(self, false_accept_rate=0.01, seed=13):
        """DirectedEnumeration's constructor (ann.cpp:270-348): (pivots, table[np][n], threshold)."""
        dem = self.dem(false_accept_rate, seed=seed)
        out = dem.get()
        dem.close()
        return out

    def ann_bruteforce(self, q):
        q = np.ascontiguousarray(q, np.float32)
        return int(self.ref.L.ref_db_ann_bruteforce(self.h, _p(q)))


class RefDem:
    def __init__(self, db, false_accept_rate, threshold, image_count, seed):
        self.db, self.L = db, db.ref.L
        self.h = self.L.ref_dem_create(db.h, false_accept_rate, threshold, image_count, seed)
        assert self.h, "gallery must hold FEATURES_COUNT features"

    def get(self):
        npiv = min(32, max(5, int(self.db.n * 0.015)))
        piv = np.empty(npiv, np.int32)
        table = np.empty((npiv, self.db.n), np.float32)
        th = C.c_float()
        got = self.L.ref_dem_get(self.h, _p(piv), _p(table), C.byref(th))
        assert got == npiv, got
        return piv, table, np.float32(th.value)

    def set_image_count(self, m):
        self.L.ref_dem_set_image_count(self.h, m)

    def recognize(self, query):
        q = np.ascontiguousarray(query, np.float32)
        bd, fo, cc = C.c_float(), C.c_int(), C.c_int()
        r = self.L.ref_dem_recognize(self.h, self.db.h, _p(q), C.byref(bd), C.byref(fo), C.byref(cc))
        return r, np.float32(bd.value), fo.value, cc.value

    def close(self):
        if self.h:
            self.L.ref_dem_destroy(self.h)
            self.h = None


class RefCls:
    """oracle/_ref/libref_cls.so -- classification.cpp slices (oracle/ref_wrap_cls.inc)."""

    def __init__(self, path):
        L = C.CDLL(path)
        self.L = L
        L.ref_cls_set_dataset.argtypes = [_vp, C.c_int64, C.c_int, _vp, C.c_int]
        L.ref_cls_features_file_name.restype = C.c_char_p
        L.ref_cls_load_dataset_cwd.restype = C.c_int64
        L.ref_cls_get_dataset.argtypes = [_vp, _vp]
        L.ref_cls_split.argtypes = [C.c_double, C.c_uint]
        L.ref_cls_train_size.restype = C.c_int64
        L.ref_cls_train_size.argtypes = [C.c_int]
        L.ref_cls_test_size.restype = C.c_int64
        L.ref_cls_get_split.argtypes = [_vp, _vp]
        L.ref_cls_get_stats.argtypes = [_vp, _vp, _vp, _vp]
        L.ref_cls_predict_row.restype = C.c_int
        L.ref_cls_predict_row.argtypes = [C.c_int, C.c_int, C.c_int64]
        L.ref_cls_predict_vec.restype = C.c_int
        L.ref_cls_predict_vec.argtypes = [C.c_int, C.c_int, _vp]
        L.ref_cls_fpnn_predict.restype = C.c_int
        L.ref_cls_fpnn_predict.argtypes = [C.c_double, C.c_int, C.c_float, _vp, C.c_int64]
        L.ref_cls_fpnn_model.restype = C.c_int
        L.ref_cls_fpnn_model.argtypes = [C.c_double, _vp]
        L.ref_cls_fastlog.restype = C.c_float
        L.ref_cls_fastlog.argtypes = [C.c_float]

    def set_dataset(self, rows, labels, n_classes):
        rows = np.ascontiguousarray(rows, np.float64)
        labels = np.ascontiguousarray(labels, np.int32)
        self.L.ref_cls_set_dataset(_p(rows), rows.shape[0], rows.shape[1], _p(labels), n_classes)

    def split(self, fraction, seed=13):
        self.L.ref_cls_split(fraction, seed)
        ncls = self.L.ref_cls_num_classes()
        sizes = [self.L.ref_cls_train_size(c) for c in range(ncls)]
        train = np.empty(sum(sizes), np.int64)
        test = np.empty(self.L.ref_cls_test_size(), np.int64)
        self.L.ref_cls_get_split(_p(train), _p(test))
        tcls = np.repeat(np.arange(ncls, dtype=np.int32), sizes)
        return train, tcls, test

    def stats(self):
        d = self.L.ref_cls_num_features()
        mn, mx, avg, sd = (np.empty(d, np.float64) for _ in range(4))
        self.L.ref_cls_get_stats(_p(mn), _p(mx), _p(avg), _p(sd))
        return mn, mx, avg, sd

    def predict_row(self, kind, param, row):
        return int(self.L.ref_cls_predict_row(kind, param, row))

    def fpnn_predict(self, scale, bf, output_ratio, q=None, row=-1):
        if q is not None:
            q = np.ascontiguousarray(q, np.float64)
        return int(self.L.ref_cls_fpnn_predict(scale, 1 if bf else 0, output_ratio, _p(q) if q is not None else None, row))

    def fpnn_model(self, scale):
        J = self.L.ref_cls_fpnn_model(scale, None)
        a = np.empty(self.L.ref_cls_num_features() * self.L.ref_cls_num_classes() * (2 * J + 1), np.float64)
        self.L.ref_cls_fpnn_model(scale, _p(a))
        return J, a

    def fastlog(self, x):
        return np.float32(self.L.ref_cls_fastlog(float(x)))

    def predict_vec(self, kind, param, q):
        q = np.ascontiguousarray(q, np.float64)
        return int(self.L.ref_cls_predict_vec(kind, param, _p(q)))


def oracle_path():
    return os.path.join(ROOT, "oracle", "liboracle.so")


def ref_path(name):
    return os.path.join(ROOT, "oracle", "_ref", f"libref_{name}.so")


def load_oracle():
    # test infrastructure: build the C restatement on demand (gcc only) when __graft_entry__.build() has not run here
    if not os.path.exists(oracle_path()):
        import subprocess

        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"], check=True, stdout=subprocess.DEVNULL)
    return Oracle(oracle_path())


def have_ref():
    return all(os.path.exists(ref_path(n)) for n in ("l2", "chi2", "kl", "cls"))


def load_ref(name):
    if name == "cls":
        return RefCls(ref_path("cls"))
    return RefMatch(ref_path(name))
