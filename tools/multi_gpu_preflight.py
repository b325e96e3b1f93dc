#!/usr/bin/env python3
"""tools/multi_gpu_preflight.py -- what a multi-GPU node can do for this library, in under a minute, BEFORE a bench is spent on
it: every step that has only ever run on one GPU exercised once, small, with its own timeout, and a
diagnosis instead of a hang.

    python tools/multi_gpu_preflight.py                 all visible devices
    python tools/multi_gpu_preflight.py --devices 0,0   rehearsal on a one-GPU box (the same code paths, "peer" = the device itself)

Steps (each prints one JSON line, the last line is the summary):
  1 peers      hipDeviceCanAccessPeer matrix of the devices (torch.cuda.can_device_access_peer)
  2 peer_copy  one --copy-mib (default 1024) device-to-device copy per ordered pair, GB/s (hipMemcpyPeerAsync underneath)
  3 replicate  a 2^16 key copied to every device with zkr_key_replicate in both forms: chosen mode, peer-direct, GB/s, the
               replica's proof == the source's
  4 sharded    ONE 2^16 proof over the first 2 (and all, if 4 or 8) devices with zkr_prove_sharded_device: first call (proves both
               ways across devices), the form that ran and why, proof == the whole key's and accepted by the native verifier
  5 rccl       one rank per device (child torchrun, gloo default group + lazily made nccl group, the shape of
               zkr_hip.replicate_key): a 64 MB broadcast over RCCL with a --rccl-timeout bound; which path the key replication
               would take (rccl / per-rank fallback) and the GB/s
Nothing here imports the oracle; acceptance is the product's own verifier against the setup's vk.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-zk-rollups_amd", "python"))


def emit(step, **kw):
    print(json.dumps(dict(step=step, **kw)), flush=True)
    return kw


def rank_main():
    """One rank of step 5 (started by torchrun)."""
    import datetime
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    devices = [int(x) for x in os.environ["ZKR_PREFLIGHT_DEVICES"].split(",")]
    dev = devices[rank]
    torch.cuda.set_device(dev)
    tmo = datetime.timedelta(seconds=int(os.environ.get("ZKR_PREFLIGHT_RCCL_TIMEOUT", "25")))
    dist.init_process_group("gloo", timeout=tmo)
    ok, err, gbps = 1, None, None
    nbytes = 64 << 20
    try:
        grp = dist.new_group(backend="nccl", timeout=tmo)
        buf = torch.full((nbytes,), rank + 1, dtype=torch.uint8, device=torch.device("cuda", dev))
        dist.broadcast(buf, src=0, group=grp)            # the communicator is made here: an RCCL that cannot start fails inside the try
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(4):
            dist.broadcast(buf, src=0, group=grp)
        torch.cuda.synchronize(dev)
        gbps = 4 * nbytes / (time.perf_counter() - t0) / 1e9
        if int(buf[::1 << 20].max().item()) != 1:
            raise RuntimeError("the broadcast did not deliver rank 0's bytes")
    except Exception as e:  # noqa: BLE001
        ok, err = 0, "%s: %s" % (type(e).__name__, str(e)[:300])
    flag = torch.tensor([ok], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)          # over gloo: the ranks agree, as replicate_key does
    rows = [None] * world
    dist.all_gather_object(rows, {"rank": rank, "device": dev, "ok": ok, "error": err, "GBps": gbps})
    if rank == 0:
        path = "rccl" if int(flag.item()) == 1 else "per-rank (every rank builds its own replica: zkr_hip.replicate_key's fallback)"
        print(json.dumps({"step": "rccl", "ranks": rows, "key_replication_would_use": path, "bytes": nbytes, "xgmi_link_GBps": 153.0}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    if os.environ.get("ZKR_PREFLIGHT_DEVICES") and "RANK" in os.environ:
        return rank_main()
    ap = argparse.ArgumentParser()
    ap.add_argument("--devices", default=None, help="comma-separated HIP ordinals (a device may repeat: rehearsal); default: all")
    ap.add_argument("--copy-mib", type=int, default=1024)
    ap.add_argument("--log-m", type=int, default=16)
    ap.add_argument("--rccl-timeout", type=int, default=25)
    ap.add_argument("--no-rccl", action="store_true")
    args = ap.parse_args()
    t_all = time.time()
    import torch
    import zkr_hip
    have = zkr_hip.device_count()
    devices = [int(x) for x in args.devices.split(",")] if args.devices else list(range(have))
    if not devices or max(devices) >= have:
        raise SystemExit("devices %s requested, %d present" % (devices, have))
    uniq = sorted(set(devices))
    summary = {"devices": devices, "rehearsal": len(uniq) < len(devices)}

    # 1 -- who can address whom
    matrix = [[True if a == b else bool(torch.cuda.can_device_access_peer(a, b)) for b in uniq] for a in uniq]
    emit("peers", devices=uniq, can_access=matrix)
    summary["all_pairs_peer_access"] = all(all(r) for r in matrix)

    # 2 -- one large copy per ordered pair
    n = args.copy_mib << 20
    rows = []
    bufs = {d: torch.empty(n, dtype=torch.uint8, device=torch.device("cuda", d)) for d in uniq}
    extra = torch.empty(n, dtype=torch.uint8, device=torch.device("cuda", uniq[0])) if len(uniq) == 1 else None
    pairs = [(a, b) for a in uniq for b in uniq if a != b] or [(uniq[0], uniq[0])]
    for a, b in pairs:
        src, dst = bufs[a], (bufs[b] if a != b else extra)
        src.fill_(a + 1)
        torch.cuda.synchronize(a)
        dst.copy_(src)                                    # warm (peer mapping)
        torch.cuda.synchronize(b)
        t0 = time.perf_counter()
        dst.copy_(src)
        torch.cuda.synchronize(b)
        torch.cuda.synchronize(a)
        dt = time.perf_counter() - t0
        rows.append({"src": a, "dst": b, "GBps": n / dt / 1e9, "intact": int(dst[:: 1 << 20].min().item()) == a + 1})
    emit("peer_copy", bytes=n, pairs=rows)
    summary["peer_copy_GBps_min"] = min(r["GBps"] for r in rows)
    summary["peer_copies_intact"] = all(r["intact"] for r in rows)
    del bufs, extra

    # 3 -- the key, device to device
    key, wb, aux = zkr_hip.ProvingKey.synth(args.log_m, 73, 0x5A4B0001, 0x5A4B00FF, device=devices[0])
    vk_bin = key.synth_vk(aux)
    pub = [int.from_bytes(wb[32 * j:32 * j + 32], "little") for j in range(1, 74)]
    want = key.prove(wb, 11, 13)
    if not zkr_hip.verify(vk_bin, want, pub):
        raise SystemExit("the whole key's proof does not verify")
    arena = key.arena()[1]
    rows = []
    for d in devices[1:] or [devices[0]]:
        for mode in ("auto", "base"):
            t0 = time.perf_counter()
            rep = key.replicate(d, mode)
            dt = time.perf_counter() - t0
            how = rep.replication()
            rows.append({"dst": d, "asked": mode, "mode": how["mode"], "peer_direct": how["peer_direct"], "seconds": dt,
                         "arena_GBps": arena / dt / 1e9 if how["mode"] == "full" else None, "proof_identical": rep.prove(wb, 11, 13) == want})
            rep.close()
    emit("replicate", arena_bytes=arena, replicas=rows)
    summary["replicas_identical"] = all(r["proof_identical"] for r in rows)

    # 4 -- one proof over several devices
    rows = []
    counts = [2] + ([len(devices)] if len(devices) in (4, 8) else [])
    for parts in counts:
        devs = devices[:parts]
        shards = [key.shard(i, parts, device=d) for i, d in enumerate(devs)]
        dws = [torch.frombuffer(bytearray(wb), dtype=torch.uint8).to(torch.device("cuda", d)) for d in devs]
        for d in set(devs):
            torch.cuda.synchronize(d)
        ptrs = [t.data_ptr() for t in dws]
        t0 = time.perf_counter()
        first = zkr_hip.prove_sharded_device(shards, ptrs, 11, 13)
        first_ms = 1e3 * (time.perf_counter() - t0)
        first_form = zkr_hip.sharded_last_form()
        t0 = time.perf_counter()
        again = zkr_hip.prove_sharded_device(shards, ptrs, 11, 13)
        ms = 1e3 * (time.perf_counter() - t0)
        form = zkr_hip.sharded_last_form()
        rows.append({"parts": parts, "devices": devs, "first_call_ms": first_ms, "first_call": first_form, "ms": ms, "form": form["form"], "reason": form["reason"],
                     "identical_to_whole_key": first == want and again == want, "verifies": zkr_hip.verify(vk_bin, again, pub)})
        for sh in shards:
            sh.close()
    emit("sharded", log_m=args.log_m, runs=rows)
    summary["sharded_ok"] = all(r["identical_to_whole_key"] and r["verifies"] for r in rows)
    summary["sharded_forms"] = [r["form"] for r in rows]
    key.close()

    # 5 -- RCCL across processes (child torchrun: this process has initialised the GPU and must not exec)
    if not args.no_rccl and len(devices) > 1:
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        env = dict(os.environ, ZKR_PREFLIGHT_DEVICES=",".join(map(str, devices)), ZKR_PREFLIGHT_RCCL_TIMEOUT=str(args.rccl_timeout))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(len(devices)), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)]
        try:
            out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=4 * args.rccl_timeout + 60)
            line = next((ln for ln in out.stdout.splitlines() if ln.startswith('{"step": "rccl"')), None)
            if line:
                print(line, flush=True)
                summary["key_replication_would_use"] = json.loads(line)["key_replication_would_use"]
            else:
                emit("rccl", error="no result from the ranks (exit %d)" % out.returncode, stderr_tail=out.stderr[-600:])
                summary["key_replication_would_use"] = "unknown (the ranks produced no result)"
        except subprocess.TimeoutExpired:
            emit("rccl", error="the ranks did not finish in time: RCCL hangs on this node -- bench.py would end in its own timeout")
            summary["key_replication_would_use"] = "unknown (timeout)"
    summary["seconds"] = time.time() - t_all
    summary["ok"] = bool(summary["peer_copies_intact"] and summary["replicas_identical"] and summary["sharded_ok"])
    emit("summary", **summary)
    return 0 if summary["ok"] else 1


if __name__ == "__main__":
    sys.exit(main())
