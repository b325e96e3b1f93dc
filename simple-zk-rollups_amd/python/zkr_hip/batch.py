"""Multi-GPU batch driver: independent proofs of a rollup batch shard across ranks (one process per
GPU).  The only collective is the one-time broadcast of the proving-key arena
(RCCL over xGMI when the backend is nccl; gloo in the CPU tests); proofs never communicate.
"""
import torch


def shard_indices(count, rank, world):
    """proof i -> rank i mod world."""
    return list(range(rank, count, world))


ARENA_CHUNK = 1 << 30


def broadcast_arena(arena, rank, dist, device, chunk=None, group=None):
    """Broadcast a uint8 tensor (the key arena) from rank 0; other ranks pass None and get a new tensor
    on `device`.  The length, then the bytes in 1 GiB pieces (arenas reach 78 GB at 2^24: every collective stays far
    below any 32-bit element count).  group: the process group that carries the bytes (default: the default group)."""
    chunk = chunk or ARENA_CHUNK
    n = torch.zeros(1, dtype=torch.int64, device=device)
    if rank == 0:
        n[0] = arena.numel()
    dist.broadcast(n, src=0, group=group)
    if rank != 0:
        arena = torch.empty(int(n.item()), dtype=torch.uint8, device=device)
    for off in range(0, arena.numel(), chunk):
        dist.broadcast(arena[off:off + chunk], src=0, group=group)
    return arena


class _CudaArray:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def _tensor_from_ptr(ptr, nbytes, device):
    """Zero-copy uint8 view of memory owned by the key: device memory of the C library (device = HIP ordinal), or host
    memory (device = None: the stand-in keys of the CPU tests, tests/test_dist.py)."""
    if device is None:
        import ctypes
        return torch.frombuffer((ctypes.c_ubyte * nbytes).from_address(ptr), dtype=torch.uint8)
    with torch.cuda.device(device):
        return torch.as_tensor(_CudaArray(ptr, nbytes), device=torch.device("cuda", device))


BCAST_MODES = ("full", "base")
last_timing = {}   # of the last broadcast_key call on this rank: {"mode", "bytes", "wire_s" (the collective alone), "adopt_s"}


def broadcast_key(key, rank, world, device, dist=None, mode=None, key_cls=None, group=None):
    """rank 0 holds `key`; every other rank receives it over ONE broadcast and returns its own ProvingKey (rank 0: the
    same object).  mode (default: env ZKR_BCAST_MODE or "full"):
      "full"  the whole arena, window tables included (4.47 GB at 2^20): adopted in place (zkr_key_adopt_arena), nothing
              is recomputed -- the right choice over xGMI (153 GB/s per link: ~30 ms);
      "base"  the compact arena (base points + QAP rows, 0.45 GB): the receiver rebuilds the window levels and twiddles
              (zkr_key_adopt_base_arena, ~0.10 s at 2^20) -- for links slower than ~40 GB/s (PCIe peer copies, a
              host-staged backend) where ten times fewer bytes outweigh the rebuild.
    key_cls: the class whose adopt_arena / adopt_base_arena build the receiver's key (default: ProvingKey); device None:
    the arenas are host memory (CPU tests drive this very function with a stand-in key class, no device needed)."""
    import os
    if key_cls is None:
        from .binding import ProvingKey as key_cls
    if world == 1:
        return key
    mode = mode or os.environ.get("ZKR_BCAST_MODE", "full")
    if mode not in BCAST_MODES:
        raise ValueError("unknown key broadcast mode %r (use one of %r)" % (mode, BCAST_MODES))
    if dist is None:
        import torch.distributed as dist
    import time
    dev = torch.device("cpu") if device is None else torch.device("cuda", device)
    t0 = time.perf_counter()
    if rank == 0:
        ptr, length = key.arena() if mode == "full" else key.base_arena()
        broadcast_arena(_tensor_from_ptr(ptr, length, device), 0, dist, dev, group=group)
        if device is not None:
            torch.cuda.synchronize(dev)
        last_timing.update(mode=mode, bytes=length, wire_s=time.perf_counter() - t0, adopt_s=0.0)
        return key
    buf = broadcast_arena(None, rank, dist, dev, group=group)
    if device is not None:
        torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    if mode == "full":
        got = key_cls.adopt_arena(buf.data_ptr(), buf.numel(), device, keepalive=buf)
    else:
        got = key_cls.adopt_base_arena(buf.data_ptr(), buf.numel(), device)
    last_timing.update(mode=mode, bytes=buf.numel(), wire_s=t1 - t0, adopt_s=time.perf_counter() - t1)
    return got


def replicate_key(key, rank, world, device, build_local, dist=None, side=None, mode=None, key_cls=None, data_group=None):
    """broadcast_key with the fallback of SURVEY.md 8(e) row 3 ("replicas only ... if RCCL is unavailable"): when the
    broadcast raises (RCCL cannot set up its transport: IPC handles refused, no peer access, a backend that is not there),
    every rank builds the key itself with build_local() -- the same key from the same source: a packed key file or the
    provingKeyBin uploaded over each GPU's own PCIe link, the seeded generator in bench.py -- and the batch goes on without
    any collective on the key.  data_group: the group that carries the key bytes (RCCL; default: the default group) -- made
    lazily, so an RCCL that cannot even initialise fails HERE, inside the try, not at start-up.  The ranks AGREE on the
    outcome over `side` (a gloo group: host TCP, independent of the GPU transport; default: the default group): one rank
    falling back while another adopted the broadcast would still be a correct batch, but the reported replication must be
    one fact.  A rank that hangs inside the collective is ended by the process group's timeout
    (init_process_group(timeout=...)): the job then fails with a non-zero exit instead of waiting for ever.
    ZKR_FORCE_BCAST_FAIL=1 raises in place of the broadcast (tests).  Returns (key, how): how = "single" (world 1), the
    data backend's name ("nccl" = RCCL, "gloo") or "per-rank"."""
    import os
    if world == 1:
        return key, "single"
    if dist is None:
        import torch.distributed as dist
    ok, err, got = 1, None, None
    try:
        if os.environ.get("ZKR_FORCE_BCAST_FAIL") == "1":
            raise RuntimeError("ZKR_FORCE_BCAST_FAIL=1: the key broadcast is made to fail")
        got = broadcast_key(key, rank, world, device, dist, mode=mode, key_cls=key_cls, group=data_group)
    except Exception as e:  # noqa: BLE001 -- whatever the transport raises: the fallback does not depend on it
        ok, err = 0, e
    flag = torch.tensor([ok], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=side)   # host tensor over the side group (the default group when it is gloo itself)
    if int(flag.item()) == 1:
        return got, dist.get_backend(data_group)
    if err is not None:
        import sys
        sys.stderr.write("zkr_hip: rank %d: key broadcast failed (%s: %s); every rank builds its own replica\n" % (rank, type(err).__name__, err))
    got = None                                                # a replica adopted from a broadcast that failed elsewhere is dropped
    if rank == 0 and key is not None:
        return key, "per-rank"
    return build_local(), "per-rank"


def prove_batch(key, witnesses, blinding, rank=0, world=1):
    """Prove this rank's shard.  witnesses: list of bytes (or None for proofs of other ranks);
    blinding: list of (r, s).  Returns {index: proof_bytes} for the local shard."""
    mine = shard_indices(len(witnesses), rank, world)
    if hasattr(key, "prove_batch_device") and torch.cuda.is_available():
        # witnesses to HBM, then two proofs in flight (zkr_prove_submit / zkr_prove_collect)
        dev = torch.device("cuda", key.device)
        dw = [torch.frombuffer(bytearray(witnesses[i]), dtype=torch.uint8).to(dev) for i in mine]
        torch.cuda.synchronize(dev)
        proofs = key.prove_batch_device([t.data_ptr() for t in dw], [blinding[i][0] for i in mine], [blinding[i][1] for i in mine])
        return dict(zip(mine, proofs))
    out = {}
    for i in mine:
        r, s = blinding[i]
        out[i] = key.prove(witnesses[i], r, s)
    return out


def gather_proofs(local, count, dist=None):
    """Collect {index: proof} dicts on every rank (256 B per proof; not on the data path)."""
    if dist is None:
        return local
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, local)
    merged = {}
    for p in parts:
        merged.update(p)
    assert sorted(merged) == list(range(count)), "every proof index must be produced exactly once"
    return merged
