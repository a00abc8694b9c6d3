"""Row sharding of one gallery over the ranks of a node.

Rank r owns rows [lo_r, hi_r); every rank scans its shard for ALL queries; the global nearest
neighbour of each query is the integer minimum over ranks of the packed (distance, index) keys,
which orders exactly like the reference's first-minimum rule (qt_cpp/db_features.cpp:329-332)
because the low word is the GLOBAL row index. One all-reduce(MIN) of qb x 8 bytes is the only
exchange step of the path; RCCL carries it over xGMI (torch.distributed backend "nccl"), gloo in
the CPU tests. Plumbing only: no distance is computed here.
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
