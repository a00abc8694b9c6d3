"""Row sharding of one gallery over the ranks of a node.

Rank r owns rows [lo_r, hi_r); every rank scans its shard for ALL queries; the global nearest
neighbour of each query is the integer minimum over ranks of the packed (distance, index) keys,
which orders exactly like the reference's first-minimum rule (qt_cpp/db_features.cpp:329-332)
because the low word is the GLOBAL row index. One all-reduce(MIN) of qb x 8 bytes is the only
exchange step of the path; RCCL carries it over xGMI (torch.distributed backend "nccl"), gloo in
the CPU tests. The K nearest rows are an all-gather of every rank's K packed keys plus an integer
K-way merge; the PNN class scores (qt_cpp/classification.cpp:188-226, partial sums over the rank's
training rows divided by the GLOBAL training-set size) are one all-reduce(SUM) of qb x C doubles; the kNN vote
(classification.cpp:116-170) is an all-gather of every rank's K nearest mean distances per class.
Plumbing only: no distance is computed here.
"""
import torch


def shard_bounds(n_rows, world, rank, granule=1):
    """Contiguous row block of `rank`: whole granules, remainder rows to the last non-empty rank."""
    units = (n_rows + granule - 1) // granule
    per = (units + world - 1) // world
    lo = min(rank * per, units) * granule
    hi = min((rank + 1) * per, units) * granule
    return min(lo, n_rows), min(hi, n_rows)


_SIGN = torch.iinfo(torch.int64).min   # 0x8000000000000000


def keys_as_int64(keys_u64_tensor):
    """uint64 keys -> int64 values with the same order (x ^ 2^63): RCCL/gloo MIN on int64 then
    equals the unsigned minimum. FIR_KEY_NONE (all ones) becomes INT64_MAX and loses to any row."""
    return keys_u64_tensor.view(torch.int64) ^ _SIGN


def allreduce_min_keys(keys_i64, group=None, async_op=False):
    """In-place MIN all-reduce of int64-viewed packed keys (see keys_as_int64)."""
    import torch.distributed as dist

    return dist.all_reduce(keys_i64, op=dist.ReduceOp.MIN, group=group, async_op=async_op)


def keys_from_int64(keys_i64):
    """Inverse of keys_as_int64 (still an int64 tensor; view it as uint64 on the host)."""
    return keys_i64 ^ _SIGN


def merge_topk_keys(parts_i64, k):
    """parts_i64[P, qb, k] (order-preserving int64 views of each shard's ascending packed keys) -> [qb, k]: the k
    smallest of the P*k candidates per query, ascending. Keys are unique (the low word is the global row index) and
    FIR_KEY_NONE sorts last, so this is exactly fir_search_topk over the whole gallery."""
    p, qb, kk = parts_i64.shape
    cat = parts_i64.permute(1, 0, 2).reshape(qb, p * kk)
    return torch.sort(cat, dim=1).values[:, :k].contiguous()


def allgather_merge_topk(keys_i64, k, group=None):
    """Every rank contributes its [qb, k] keys (keys_as_int64 view); every rank gets the merged [qb, k]."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    buf = [torch.empty_like(keys_i64) for _ in range(world)]
    dist.all_gather(buf, keys_i64.contiguous(), group=group)
    return merge_topk_keys(torch.stack(buf), k)


def allreduce_sum_scores(scores_f64, group=None, async_op=False):
    """In-place SUM all-reduce of the PNN class scores [qb, C] (float64). Each rank computes them over its own
    training rows with the global training-set size as the divisor (fir_cls_set_total_training_size)."""
    import torch.distributed as dist

    return dist.all_reduce(scores_f64, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def first_max_class(scores_f64):
    """The reference's arg-max (strict '<' from -DBL_MAX in class order, classification.cpp:217-224): first maximum."""
    mx = scores_f64.max(dim=1, keepdim=True).values
    return (scores_f64 == mx).to(torch.int8).argmax(dim=1).to(torch.int32)


_DBL_MAX = torch.finfo(torch.float64).max


def merge_knn_class_nearest(parts_f64, k):
    """parts_f64[P, qb, C, k] (fir_cls_knn_class_nearest of every training-row shard: the k smallest mean distances per
    class, ascending, DBL_MAX-padded) -> kth[qb, C]: the k-th smallest of each class over all shards. The reference's
    vote (classification.cpp:151-160: rows in distance order vote until one class has k) is won by the class whose
    k-th nearest member is nearest, which is the first minimum of this table."""
    p, qb, c, kk = parts_f64.shape
    cat = parts_f64.permute(1, 2, 0, 3).reshape(qb, c, p * kk)
    return torch.sort(cat, dim=2).values[:, :, k - 1].contiguous()


def allgather_knn_class_nearest(nearest_f64, k, group=None):
    """Every rank contributes its [qb, C, k] table; every rank gets the merged kth[qb, C]."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    buf = [torch.empty_like(nearest_f64) for _ in range(world)]
    dist.all_gather(buf, nearest_f64.contiguous(), group=group)
    return merge_knn_class_nearest(torch.stack(buf), k)


def knn_class_of(kth_f64, class_sizes):
    """First minimum of kth[qb, C] in class order; when no class has k rows anywhere (all DBL_MAX) the reference's loop
    ends without a winner and its arg-max of the vote counts picks the first largest class (classification.cpp:161-168).
    class_sizes[C] = GLOBAL training rows per class."""
    mn = kth_f64.min(dim=1, keepdim=True).values
    best = (kth_f64 == mn).to(torch.int8).argmax(dim=1).to(torch.int32)
    none = mn.squeeze(1) >= _DBL_MAX
    if bool(none.any()):
        sizes = torch.as_tensor(class_sizes, dtype=torch.int64)
        largest = int((sizes == sizes.max()).to(torch.int8).argmax())
        best = torch.where(none, torch.full_like(best, largest), best)
    return best
