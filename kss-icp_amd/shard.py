"""Multi-GPU sharding of independent registrations.

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
    """ctypes IcpResult array -> uint8 [n, 96] with pair_id rewritten to the GLOBAL pair index."""
    n = len(records)
    out = np.zeros((n, RECORD_BYTES), np.uint8)
    for i in range(n):
        records[i].pair_id = lo + i
        out[i] = np.frombuffer(bytes(records[i]), dtype=np.uint8)
    return out


def array_to_records(arr, record_type):
    arr = np.ascontiguousarray(arr, dtype=np.uint8).reshape(-1, RECORD_BYTES)
    return [record_type.from_buffer_copy(arr[i].tobytes()) for i in range(len(arr))]


def gather_records(local, npairs, world, rank, device=None):
    """All-gather the per-pair records of every rank (torch.distributed; backend nccl == RCCL on ROCm).

    local: uint8 [n_local, 96].  Returns uint8 [npairs, 96] ordered by global pair id on every rank.
    Ranks may own different counts: records are padded to the largest shard for ONE fixed-size collective."""
    import torch
    import torch.distributed as dist
    m = max_shard(npairs, world)
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
