"""Multi-GPU sharding of the registration path.

The path shards over (source, target) pairs with no data-path collective: rank r of W owns the
contiguous block of pairs [lo, hi).  The only exchange is the final all-gather of the fixed 96-byte
kss_icp_result record per pair (RCCL over xGMI on GPUs; gloo in the CPU tests).  Message sizes are
KB-scale, i.e. latency bound: one collective per batch, never per pair.
"""
import ctypes as C

import numpy as np

RECORD_BYTES = 96


def shard_range(npairs, world, rank):
    """Static contiguous partition; the first (npairs % world) ranks take one extra pair."""
    base, rem = divmod(int(npairs), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def max_shard(npairs, world):
    return -(-int(npairs) // int(world))


def records_to_array(records, lo):
    """ctypes record array -> uint8 [n, sizeof(record)].  kss_icp_result records (96 bytes) get pair_id rewritten to the
    GLOBAL pair index; other fixed-size records (kss_register_result: s, R, t, ...) are copied as they are."""
    n = len(records)
    size = C.sizeof(records[0]) if n else RECORD_BYTES
    if n and isinstance(records, C.Array) and hasattr(records[0], "pair_id"):
        # one contiguous ctypes array (what kss_icp_batch_dev fills): no per-record marshalling -- a 1024-record shard is 96 KB
        out = np.frombuffer(records, dtype=np.uint8).reshape(n, size).copy()
        off = type(records[0]).pair_id.offset
        out[:, off:off + 4] = (lo + np.arange(n, dtype=np.int32)).view(np.uint8).reshape(n, 4)
        return out
    out = np.zeros((n, size), np.uint8)
    for i in range(n):
        if hasattr(records[i], "pair_id"):
            records[i].pair_id = lo + i
        out[i] = np.frombuffer(bytes(records[i]), dtype=np.uint8)
    return out


def array_to_records(arr, record_type):
    arr = np.ascontiguousarray(arr, dtype=np.uint8).reshape(-1, C.sizeof(record_type))
    return [record_type.from_buffer_copy(arr[i].tobytes()) for i in range(len(arr))]


def gather_records(local, npairs, world, rank, device=None, record_bytes=RECORD_BYTES):
    """All-gather the per-pair records of every rank (torch.distributed; backend nccl == RCCL on ROCm).

    local: uint8 [n_local, B] (B = 96 for kss_icp_result; any fixed record size works).  Returns uint8 [npairs, B]
    ordered by global pair id on every rank.  Ranks may own different counts: records are padded to the largest shard
    for ONE fixed-size collective.  `record_bytes` is needed only by a rank that owns no record."""
    import torch
    import torch.distributed as dist
    m = max_shard(npairs, world)
    RECORD_BYTES = int(local.shape[1]) if len(local) else int(record_bytes)
    buf = torch.zeros((m, RECORD_BYTES), dtype=torch.uint8, device=device)
    if len(local):
        buf[:len(local)] = torch.from_numpy(np.ascontiguousarray(local)).to(buf.device)
    allbuf = torch.zeros((world * m, RECORD_BYTES), dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(allbuf, buf)
    allbuf = allbuf.cpu().numpy().reshape(world, m, RECORD_BYTES)
    out = np.zeros((npairs, RECORD_BYTES), np.uint8)
    for r in range(world):
        lo, hi = shard_range(npairs, world, r)
        out[lo:hi] = allbuf[r, :hi - lo]
    return out


# ---- the single-pair exchange step: source rows split over ranks, target replicated ----------------
def make_allreduce(binding, device=None, group=None):
    """A kss_allreduce_fn (ctypes callback) that sums n host doubles over the ranks of torch.distributed
    (backend nccl == RCCL over xGMI on GPUs, gloo in the CPU tests): the ONE collective per ICP iteration of a
    registration whose source rows are split over ranks.  Keep the returned object alive while it is in use.

    160 bytes per call: latency bound.  With `device` set the tensor lives on the GPU (what RCCL needs)."""
    import torch
    import torch.distributed as dist

    def _cb(_user, values, n):
        try:
            arr = np.ctypeslib.as_array(values, shape=(int(n),))
            t = torch.from_numpy(arr.copy())
            if device is not None:
                t = t.to(device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            arr[:] = t.cpu().numpy()
            return 0
        except Exception:      # never unwind through the C caller
            return -1

    return binding.ALLREDUCE_FN(_cb)


def icp_split_source(ctx, binding, src_rows, tgt, allreduce, **params):
    """ICP of ONE pair with this rank's rows of the source (shard_range over the rows) and the whole target.
    Every rank gets the same T / iterations / fitness (fitness = mean over ALL source rows)."""
    p = ctx.icp_params(**params)
    p.allreduce = allreduce
    return ctx.icp(src_rows, tgt, p)


def _load_rccl():
    import torch  # noqa: F401  (its librccl is the one dlopen("librccl.so") inside libkssicp.so resolves to)
    for name in ("librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"):
        try:
            return C.CDLL(name)
        except OSError:
            continue
    import os
    import torch as _t
    return C.CDLL(os.path.join(os.path.dirname(_t.__file__), "lib", "librccl.so"))


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def rccl_comm(rank, world, device=None):
    """An ncclComm_t of our own over the ranks of the initialised torch.distributed job (torch does not hand out its
    communicator): rank 0 draws the unique id, one broadcast carries its 128 bytes, every rank joins.  Returns
    (librccl handle, comm as c_void_p); call lib.ncclCommDestroy(comm) when done.  The HIP device must be current."""
    import torch
    import torch.distributed as dist
    lib = _load_rccl()
    uid = _UniqueId()
    if rank == 0 and lib.ncclGetUniqueId(C.byref(uid)) != 0:
        raise RuntimeError("ncclGetUniqueId failed")
    t = torch.from_numpy(np.frombuffer(bytes(uid), dtype=np.uint8).copy())
    if world > 1:
        if device is not None:
            t = t.to(device)
        dist.broadcast(t, src=0)
        t = t.cpu()
    C.memmove(C.byref(uid), t.numpy().tobytes(), 128)
    comm = C.c_void_p()
    lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
    if lib.ncclCommInitRank(C.byref(comm), int(world), uid, int(rank)) != 0:
        raise RuntimeError("ncclCommInitRank failed")
    lib.ncclCommDestroy.argtypes = [C.c_void_p]
    return lib, comm


def rccl_allreduce_params(ctx, binding, library, comm, **params):
    """kss_icp_params whose exchange step is the library's own RCCL callback (ncclAllReduce on the context's stream,
    no Python in the loop).  Returns (params, keepalive): keep `keepalive` referenced while the params are in use."""
    link = binding.RcclLink(ctx.h, comm)
    p = ctx.icp_params(**params)
    p.allreduce = C.cast(library.kss_rccl_allreduce_sum, binding.ALLREDUCE_FN)
    p.allreduce_user = C.cast(C.pointer(link), C.c_void_p)
    return p, link
