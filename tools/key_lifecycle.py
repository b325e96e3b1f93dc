"""Keys come and go: replicas, shards (with their rebuilt window levels), packed-file reloads and whole keys are created, used and
closed in a loop; device memory must return to where it was.  python tools/key_lifecycle.py [log_m=16] [rounds=12]"""
import os, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "simple-zk-rollups_amd", "python"))
import torch
import zkr_hip

log_m = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 12
key, wb, _ = zkr_hip.ProvingKey.synth(log_m, device=0, want_aux=False)
want = key.prove(wb, 3, 4)
path = os.path.join(tempfile.mkdtemp(), "k.zkrkey")
key.save(path)


def one_round():
    rep = key.replicate(0, "base")
    assert rep.prove(wb, 3, 4) == want
    shards = [key.shard(i, 4) for i in range(4)]
    assert zkr_hip.prove_sharded(shards, wb, 3, 4) == want
    again = zkr_hip.ProvingKey.load_file(path)
    assert again.prove_batch([wb, wb], [3, 3], [4, 4]) == [want, want]
    fresh, w2, _ = zkr_hip.ProvingKey.synth(log_m - 2, device=0, want_aux=False)
    fresh.prove(w2)
    for k in [rep, again, fresh] + shards:
        k.close()


one_round()                                   # whatever is built once per process or per key (streams, tables of delta) is in place
torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info(0)[0]
for r in range(rounds):
    one_round()
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info(0)[0]
print("2^%d: %d rounds of replica + 4 shards + file reload + a fresh key, each used and closed: device memory in use changed by %.1f MiB" % (log_m, rounds, (free0 - free1) / 2 ** 20))
sys.exit(1 if free0 - free1 > (32 << 20) else 0)
