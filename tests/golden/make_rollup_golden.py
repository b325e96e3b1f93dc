"""Writes tests/golden/rollup_tx.json with the ORACLE (oracle/rollup.py, pinned on the reference's vectors): a seeded
BatchProcessTx(2, 3) batch -- accounts, signed transactions, circuit inputs, the public signals and the hash of the
leaves -- so that the product alone can be checked against committed data (tests/test_rollup.py).
python tests/golden/make_rollup_golden.py"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import rollup as o  # noqa: E402

rnd = random.Random(0x5A4B0F30)
depth, batch = 3, 2
privs = [rnd.randrange(o.R) for _ in range(3)]
accounts = {i: [*o.gen_public_key(privs[i]), 7 * 10 ** 18 + i, i] for i in range(3)}
tree = o.Tree(depth)
for i in range(3):
    tree.update(i, o.leaf_hash(accounts[i][:2], accounts[i][2], accounts[i][3]))
root0 = tree.root
txs = [o.process_tx_inputs(tree, accounts, 0, 1, 3 * 10 ** 17, 10 ** 15, privs[0]),
       o.process_tx_inputs(tree, accounts, 2, 2, 5 * 10 ** 17, 2 * 10 ** 15, privs[2])]       # the second one to oneself
st = lambda x: [st(v) for v in x] if isinstance(x, (list, tuple)) else str(x)
out = dict(depth=depth, batch=batch, privs=st(privs), pubs=st([o.gen_public_key(p) for p in privs]),
           formatted=st([o.format_priv_key(p) for p in privs]), root_before=str(root0), root_after=str(tree.root),
           inputs={f: st([t[f] for t in txs]) for f in o.TX_FIELDS}, public_signals=st(o.batch_public_signals(txs)),
           hash_1=str(o.multi_hash([32767])), hash_lr=str(o.multi_hash([12345, 45678])))
json.dump(out, open(os.path.join(HERE, "rollup_tx.json"), "w"), indent=1)
print("wrote rollup_tx.json; new root", tree.root)
