import sys, os, ctypes
sys.path.insert(0, "/root/repo/simple-zk-rollups_amd/python"); sys.path.insert(0, "/root/repo/oracle"); sys.path.insert(0, "/root/repo/tests")
import torch, zkr_hip
from zkr_hip import rollup as n
from zkr_hip.binding import lib
from test_rollup import scenario, as_inputs
c = n.RollupCircuit(2, 6)
txs, _, _ = scenario(2, 6, 41, False, n_accounts=5)
flat = c.flatten_inputs(as_inputs(txs))
buf = b"".join(int(v).to_bytes(32, "little") for v in flat)
out = torch.zeros((1, c.n_vars * 32), dtype=torch.uint8, device="cuda")
rc = lib().zkr_rollup_witness_batch_device(2, 6, buf, len(flat), 1, out.data_ptr(), 0)
print("rc", rc, lib().zkr_last_error().decode() if rc else "")
dev = bytes(out[0].cpu().numpy().tobytes())
host = c.calculate_witness(flat)
diff = [i for i in range(c.n_vars) if dev[32*i:32*i+32] != host[32*i:32*i+32]]
print("n_vars", c.n_vars, "n_public", c.n_public, "differing signals:", len(diff), diff[:20])
K = (c.n_vars - c.n_public - 1) // 2
print("K", K, "first diff relative to tx0 private start:", [d - c.n_public - 1 for d in diff[:10]])
