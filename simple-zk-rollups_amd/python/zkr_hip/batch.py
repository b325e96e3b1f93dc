"""Multi-GPU batch driver: independent proofs of a rollup batch shard across ranks (one process per
GPU).  The only collective is the one-time broadcast of the proving-key arena
(RCCL over xGMI when the backend is nccl; gloo in the CPU tests); proofs never communicate.
"""
import torch


def shard_indices(count, rank, world):
    """proof i -> rank i mod world."""
    return list(range(rank, count, world))


def broadcast_key(key, rank, world, device, dist=None):
    """rank 0 holds `key`; every other rank receives the arena bytes and adopts them.
    Returns a ProvingKey on every rank (rank 0: the same object)."""
    from .binding import ProvingKey
    if world == 1:
        return key
    import torch.distributed as d
    dist = dist or d
    dev = torch.device("cuda", device)
    n = torch.zeros(1, dtype=torch.int64, device=dev)
    if rank == 0:
        ptr, length = key.arena()
        n[0] = length
    dist.broadcast(n, src=0)
    length = int(n.item())
    if rank == 0:
        # view the existing arena as a tensor without copying
        arena = _tensor_from_ptr(ptr, length, device)
        dist.broadcast(arena, src=0)
        return key
    buf = torch.empty(length, dtype=torch.uint8, device=dev)
    dist.broadcast(buf, src=0)
    torch.cuda.synchronize(dev)
    return ProvingKey.adopt_arena(buf.data_ptr(), length, device, keepalive=buf)


class _CudaArray:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def _tensor_from_ptr(ptr, nbytes, device):
    with torch.cuda.device(device):
        return torch.as_tensor(_CudaArray(ptr, nbytes), device=torch.device("cuda", device))


def prove_batch(key, witnesses, blinding, rank=0, world=1):
    """Prove the proofs of this rank's shard.  witnesses: list of bytes; blinding: list of (r, s).
    Returns {index: proof_bytes} for the local shard."""
    out = {}
    for i in shard_indices(len(witnesses), rank, world):
        r, s = blinding[i]
        out[i] = key.prove(witnesses[i], r, s)
    return out
