"""CPU: SURVEY 8(f-3) -- the reference's rollup circuit without circom / snarkjs.

Oracle (oracle/rollup.py) pinned against the reference's own vectors, the native host crypto (csrc/rollup.cpp through
the C ABI) against the oracle, and the native constraint system + witness builder against the equalities the
reference's circuit tests assert (prover/__tests__/hasher.test.ts:24-26,45-47, eddsa.test.ts:22-37,
merkletree.test.ts:88,118-129, processtx.test.ts:132-138,240-246, batchprocesstx.test.ts:247-253)."""
import json
import os
import random

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _kat():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "rollup_kat.json")))


def parse_r1cs(b):
    nv, npub, nc = (int.from_bytes(b[i:i + 4], "little") for i in (0, 4, 8))
    off, cons = 12, []
    for _ in range(nc):
        row = []
        for _s in range(3):
            k = int.from_bytes(b[off:off + 4], "little")
            off += 4
            row.append([(int.from_bytes(b[off + 36 * e:off + 36 * e + 4], "little"), int.from_bytes(b[off + 36 * e + 4:off + 36 * e + 36], "little")) for e in range(k)])
            off += 36 * k
        cons.append(row)
    assert off == len(b)
    return nv, npub, cons


def first_violated(cons, w, R):
    ev = lambda lc: sum(c * w[s] for s, c in lc) % R
    for i, (a, b, c) in enumerate(cons):
        if ev(a) * ev(b) % R != ev(c):
            return i
    return -1


def scenario(batch, depth, seed, self_send=False, n_accounts=4):
    """Accounts + `batch` chained transactions against an oracle balance tree: the flow of processtx.test.ts:24-130."""
    import rollup as o
    rnd = random.Random(seed)
    privs = [rnd.randrange(o.R) for _ in range(n_accounts)]
    accounts = {i: [*o.gen_public_key(privs[i]), 50 * 10 ** 18 + i, i] for i in range(n_accounts)}
    tree = o.Tree(depth)
    for i in range(n_accounts):
        tree.update(i, o.leaf_hash(accounts[i][:2], accounts[i][2], accounts[i][3]))
    txs = []
    for t in range(batch):
        f = rnd.randrange(n_accounts)
        to = f if self_send else (f + 1 + rnd.randrange(n_accounts - 1)) % n_accounts
        txs.append(o.process_tx_inputs(tree, accounts, f, to, 10 ** 18 * (t + 1), 5 * 10 ** 17, privs[f]))
    return txs, tree, privs


def as_inputs(txs):
    from zkr_hip import rollup as n
    return {f: [t[f] for t in txs] for f in n.TX_INPUT_FIELDS}


def ints(b):
    return [int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)]


# ------------------------------------------------------------------ oracle vs the reference's own vectors
def test_oracle_round_constants_are_the_deployed_contract_constants():
    """CircomLib.json bytecode (generated MiMCSponge-220): PUSH32 #0 is r, then one unreduced constant per round 1..218."""
    import rollup as o
    push = [int(x, 16) for x in _kat()["mimcsponge_push32"]]
    cts = o.mimc_constants()
    assert push[0] == o.R and len(push) == 219
    assert [p % o.R for p in push[1:]] == cts[1:-1] and cts[0] == 0 and cts[-1] == 0


def test_oracle_reproduces_the_reference_key_pairs():
    """scripts/index.js:108-118: pins the hash, formatPrivKeyForBabyJub and BabyJub scalar multiplication end to end."""
    import rollup as o
    for kp in _kat()["keypairs"]:
        assert o.gen_public_key(int(kp["priv"])) == tuple(int(v) for v in kp["pub"])
        assert o.bj_on_curve(tuple(int(v) for v in kp["pub"]))
    assert o.bj_on_curve(o.BASE8) and o.bj_mul(o.BASE8, o.SUBORDER) == (0, 1)
    assert o.keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"


# ------------------------------------------------------------------ native host crypto vs oracle
def test_native_hash_keys_and_signatures_match_the_oracle():
    import rollup as o
    from zkr_hip import rollup as n
    rnd = random.Random(7)
    for k in (1, 2, 3, 5, 8):
        v = [rnd.randrange(o.R) for _ in range(k)]
        assert n.multi_hash(v) == o.multi_hash(v)
    assert n.multi_hash([32767]) == o.multi_hash([32767])                      # hasher.test.ts:15-26
    assert n.hash_left_right(12345, 45678) == o.multi_hash([12345, 45678])     # hasher.test.ts:36-47
    assert n.multi_hash([o.R + 5, (1 << 256) - 1]) == o.multi_hash([5, ((1 << 256) - 1) % o.R])   # operands mod r
    for kp in _kat()["keypairs"]:
        assert n.gen_public_key(int(kp["priv"])) == tuple(int(v) for v in kp["pub"])
    for t in range(6):
        priv = rnd.randrange(o.R) if t else 0
        msg = [rnd.randrange(o.R) for _ in range(1 + t)]
        so, sn = o.sign(priv, msg), n.sign(priv, msg)
        assert (sn["R8"][0], sn["R8"][1], sn["S"]) == so
        pub = n.gen_public_key(priv)
        assert pub == o.gen_public_key(priv)
        assert n.verify(msg, sn, pub) and o.verify(msg, so, pub)               # eddsa.test.ts:22-37
        bad = list(msg)
        bad[0] = (bad[0] + 1) % o.R
        assert not n.verify(bad, sn, pub)
        assert not n.verify(msg, {"R8": sn["R8"], "S": sn["S"] + o.SUBORDER}, pub)      # S >= subgroup order
        assert not n.verify(msg, {"R8": (sn["R8"][0], sn["R8"][1] + 1), "S": sn["S"]}, pub)  # R8 off the curve
        assert not n.verify(msg, sn, (pub[0], o.R))                                    # not a field element


def test_native_crypto_rejects_bad_arguments():
    import zkr_hip
    from zkr_hip import rollup as n
    with pytest.raises(zkr_hip.ZkrError):
        n.gen_public_key(n.SNARK_FIELD_SIZE)        # crypto.ts:79 assert(privKey < SNARK_FIELD_SIZE)
    with pytest.raises(zkr_hip.ZkrError):
        n.sign(5, [n.SNARK_FIELD_SIZE])
    with pytest.raises(zkr_hip.ZkrError):
        n.RollupCircuit(0, 6)
    with pytest.raises(zkr_hip.ZkrError):
        n.RollupCircuit(2, 33)


# ------------------------------------------------------------------ constraint system + witness builder
def test_tx_circuit_geometry():
    """tx.circom:3 = BatchProcessTx(2, 6): 73 public signals (TxVerifier.sol:281 takes uint[73]) and a 2^17 domain."""
    from zkr_hip import rollup as n
    c = n.RollupCircuit()
    assert (c.batch, c.depth, c.n_public) == (2, 6, 73)
    assert 1 << 16 < c.n_constraints + c.n_public + 1 <= 1 << 17
    b = c.r1cs()
    assert [int.from_bytes(b[i:i + 4], "little") for i in (0, 4, 8)] == [c.n_vars, 73, c.n_constraints]


@pytest.fixture(scope="module")
def circuit_2x4():
    from zkr_hip import rollup as n
    c = n.RollupCircuit(2, 4)
    nv, npub, cons = parse_r1cs(c.r1cs())
    assert (nv, npub, len(cons)) == (c.n_vars, c.n_public, c.n_constraints)
    return c, cons


def test_witness_satisfies_the_system_and_outputs_the_new_root(circuit_2x4):
    """processtx.test.ts:132-138 / batchprocesstx.test.ts:247-253: main.newBalanceTreeRoot == the root of the tree the
    operator computed; plus: every constraint holds and the public signals are the circuit inputs in circom's order."""
    import rollup as o
    c, cons = circuit_2x4
    for seed, self_send in ((1, False), (2, False), (3, True)):     # self-send: processtx.test.ts:141-246
        txs, tree, _ = scenario(2, 4, seed, self_send)
        wb = c.calculate_witness(as_inputs(txs))
        w = ints(wb)
        assert len(w) == c.n_vars and w[0] == 1
        assert first_violated(cons, w, o.R) == -1
        assert w[1] == tree.root == txs[-1]["newBalanceTreeRoot"]
        assert c.public_signals(wb) == o.batch_public_signals(txs)
        assert c.calculate_witness(c.flatten_inputs(as_inputs(txs))) == wb


def test_every_private_signal_is_constrained(circuit_2x4):
    """No free signals: changing any one witness value breaks some constraint (sampled)."""
    import rollup as o
    c, cons = circuit_2x4
    txs, _, _ = scenario(2, 4, 5)
    w = ints(c.calculate_witness(as_inputs(txs)))
    used = {}
    for i, row in enumerate(cons):
        for lc in row:
            for s, _c in lc:
                used.setdefault(s, []).append(i)
    assert set(range(1, c.n_vars)) <= set(used)
    rnd = random.Random(11)
    for s in [1, 2, c.n_public] + [rnd.randrange(1, c.n_vars) for _ in range(200)]:
        w2 = list(w)
        w2[s] = (w2[s] + 1 + rnd.randrange(o.R - 1)) % o.R
        ev = lambda lc: sum(cf * w2[x] for x, cf in lc) % o.R
        assert any(ev(cons[i][0]) * ev(cons[i][1]) % o.R != ev(cons[i][2]) for i in used[s]), s


@pytest.mark.parametrize("what,needle", [
    ("signature", "signature"), ("message", "signature"), ("nonce", "nonce"), ("balance", "balance"), ("zero_amount", "amount"),
    ("sender_path", "sender leaf"), ("recipient_leaf", "recipient leaf"), ("intermediate_root", "intermediate root"),
    ("chain", "previous one"), ("huge_amount", "250 bits"), ("big_s", "subgroup order"), ("index", "fits the tree")])
def test_inputs_that_violate_the_circuit_are_refused(what, needle):
    """Where Circuit.calculateWitness throws (merkletree.test.ts:118-129; the `===` lines of processtx.circom:83-101,
    127-136,185, batchprocesstx.circom:67-69): ZKR_ERR_UNSATISFIED naming the statement."""
    import rollup as o
    import zkr_hip
    from zkr_hip import rollup as n
    c = n.RollupCircuit(2, 3)
    txs, _, privs = scenario(2, 3, 21)
    t = txs[1]
    if what == "signature":
        t["txData"][7] = (t["txData"][7] + 1) % o.SUBORDER
    elif what == "message":
        t["txData"][3] += 1                                  # fee changed after signing
    elif what == "nonce":
        t["txSenderNonce"] += 1
    elif what == "balance":                                  # amount + fee == balance is not enough (strictly greater)
        bal = t["txSenderBalance"]
        txs, _, _ = scenario_with_amount(bal)
        t = txs[0]
        c = n.RollupCircuit(1, 3)
    elif what == "zero_amount":
        txs, _, _ = scenario_with_amount(None, amount=0)
        c = n.RollupCircuit(1, 3)
    elif what == "sender_path":
        t["txSenderPathElements"][1] += 1
    elif what == "recipient_leaf":
        t["txRecipientBalance"] += 1
    elif what == "intermediate_root":
        t["intermediateBalanceTreeRoot"] += 1
    elif what == "chain":
        txs = [txs[0], dict(txs[0])]                         # each valid alone, but the second ignores the first's update
    elif what == "huge_amount":
        txs, _, _ = scenario_with_amount(None, amount=1 << 250)
        c = n.RollupCircuit(1, 3)
    elif what == "big_s":
        t["txData"][7] += o.SUBORDER
    elif what == "index":
        t["txData"][0] += 8
    with pytest.raises(zkr_hip.ZkrError) as e:
        c.calculate_witness(as_inputs(txs))
    assert e.value.code == -7 and needle in str(e.value), str(e.value)


def scenario_with_amount(balance_total, amount=None, depth=3):
    """One transaction whose amount + fee equals the sender's whole balance (or with the given amount)."""
    import rollup as o
    rnd = random.Random(33)
    privs = [rnd.randrange(o.R) for _ in range(2)]
    accounts = {i: [*o.gen_public_key(privs[i]), 10 ** 18, 0] for i in range(2)}
    tree = o.Tree(depth)
    for i in range(2):
        tree.update(i, o.leaf_hash(accounts[i][:2], accounts[i][2], accounts[i][3]))
    fee = 10 ** 17
    amt = amount if amount is not None else accounts[0][2] - fee
    return [o.process_tx_inputs(tree, accounts, 0, 1, amt, fee, privs[0])], tree, privs


def test_balance_edge_is_exact():
    """balance == amount + fee + 1 passes, balance == amount + fee does not (processtx.circom:98-101, strict >)."""
    import rollup as o
    from zkr_hip import rollup as n
    c = n.RollupCircuit(1, 3)
    rnd = random.Random(34)
    privs = [rnd.randrange(o.R) for _ in range(2)]
    accounts = {i: [*o.gen_public_key(privs[i]), 10 ** 18, 0] for i in range(2)}
    tree = o.Tree(3)
    for i in range(2):
        tree.update(i, o.leaf_hash(accounts[i][:2], accounts[i][2], accounts[i][3]))
    tx = o.process_tx_inputs(tree, accounts, 0, 1, 10 ** 18 - 10 ** 17 - 1, 10 ** 17, privs[0])
    w = ints(c.calculate_witness(as_inputs([tx])))
    assert w[1] == tree.root


def test_product_side_operator_state_matches_the_oracle():
    """zkr_hip.rollup.RollupState (native hash / signatures) builds the same circuit inputs and roots as the oracle's
    state machine, including a transfer to oneself."""
    import rollup as o
    from zkr_hip import rollup as n
    rnd = random.Random(71)
    privs = [rnd.randrange(o.R) for _ in range(3)]
    st, tree, accounts = n.RollupState(4), o.Tree(4), {}
    for i, pv in enumerate(privs):
        pub = o.gen_public_key(pv)
        st.deposit(i, pub, 10 ** 19 + i, i)
        accounts[i] = [*pub, 10 ** 19 + i, i]
        tree.update(i, o.leaf_hash(pub, 10 ** 19 + i, i))
    assert st.tree.root == tree.root and n.BalanceTree(4).root == o.Tree(4).root
    txs_n, txs_o = [], []
    for frm, to in ((0, 1), (1, 1), (2, 0), (0, 2)):
        txs_n.append(st.transfer(frm, to, 10 ** 17 * (frm + 1), 10 ** 15, privs[frm]))
        txs_o.append(o.process_tx_inputs(tree, accounts, frm, to, 10 ** 17 * (frm + 1), 10 ** 15, privs[frm]))
        want = dict(txs_o[-1])
        assert want.pop("newBalanceTreeRoot") == st.tree.root == tree.root
        assert txs_n[-1] == want
    c = n.RollupCircuit(4, 4)
    wb = c.calculate_witness(st.batch_inputs(txs_n))
    assert c.public_signals(wb) == o.batch_public_signals(txs_o)


def test_withdraw_circuit_derives_the_public_key():
    """prover/__tests__/withdraw.test.ts:27-36 / publickeyderivation.test.ts:26-34: main.publicKey == genPublicKey(priv)
    for privateKey = formatPrivKeyForBabyJub(priv); here also on the reference's fixed key pairs."""
    import rollup as o
    import zkr_hip
    from zkr_hip import rollup as n
    c = n.WithdrawCircuit()
    nv, npub, cons = parse_r1cs(c.r1cs())
    assert npub == 3
    rnd = random.Random(81)
    cases = [(int(kp["priv"]), tuple(int(v) for v in kp["pub"])) for kp in _kat()["keypairs"]] + [(pv, o.gen_public_key(pv)) for pv in (rnd.randrange(o.R), 0)]
    for priv, pub in cases:
        fk = n.format_priv_key(priv)
        assert fk == o.format_priv_key(priv) and fk < 1 << 253
        nul = rnd.randrange(o.R)
        wb = c.calculate_witness({"privateKey": fk, "nullifier": nul})
        w = ints(wb)
        assert len(w) == nv and first_violated(cons, w, o.R) == -1
        assert c.public_signals(wb) == [pub[0], pub[1], nul]
    used = {s for row in cons for lc in row for s, _c in lc}
    assert set(range(1, nv)) <= used
    with pytest.raises(zkr_hip.ZkrError) as e:
        c.calculate_witness({"privateKey": 1 << 253, "nullifier": 1})     # Num2Bits(253), publickeyderivation.circom:12-13
    assert e.value.code == -7


def test_ragged_or_out_of_field_circuit_inputs_are_refused():
    """A path one element short, an extra transaction, or a value >= r passed as a flat list never reaches the builder."""
    import zkr_hip
    from zkr_hip import rollup as n
    c = n.RollupCircuit(2, 3)
    txs, _, _ = scenario(2, 3, 23)
    inp = as_inputs(txs)
    short = dict(inp, txSenderPathElements=[inp["txSenderPathElements"][0], inp["txSenderPathElements"][1][:-1]])
    with pytest.raises(zkr_hip.ZkrError) as e:
        c.calculate_witness(short)
    assert e.value.code == -5
    with pytest.raises(zkr_hip.ZkrError):
        c.calculate_witness(dict(inp, balanceTreeRoot=inp["balanceTreeRoot"] + [1]))
    flat = c.flatten_inputs(inp)
    with pytest.raises(zkr_hip.ZkrError) as e:
        c.calculate_witness(flat[:5] + [n.SNARK_FIELD_SIZE] + flat[6:])       # the C ABI takes canonical field elements only
    assert e.value.code == -5 and ">= r" in str(e.value)
    assert c.calculate_witness(dict(inp, txData=[[v + n.SNARK_FIELD_SIZE for v in t] for t in inp["txData"]])) == c.calculate_witness(inp)  # dict inputs are reduced like circom's


def test_product_alone_reproduces_the_committed_rollup_fixture():
    """tests/golden/rollup_tx.json (written by the pinned oracle, make_rollup_golden.py): the native library alone --
    no oracle import on this path -- derives the same keys, signs the same transactions, builds the same tree and the
    same public signals."""
    from zkr_hip import rollup as n
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "rollup_tx.json")))
    privs = [int(v) for v in fx["privs"]]
    assert [[str(c) for c in n.gen_public_key(p)] for p in privs] == fx["pubs"]
    assert [str(n.format_priv_key(p)) for p in privs] == fx["formatted"]
    assert str(n.multi_hash([32767])) == fx["hash_1"] and str(n.hash_left_right(12345, 45678)) == fx["hash_lr"]
    st = n.RollupState(fx["depth"])
    for i, p in enumerate(privs):
        st.deposit(i, n.gen_public_key(p), 7 * 10 ** 18 + i, i)
    assert str(st.tree.root) == fx["root_before"]
    txs = [st.transfer(0, 1, 3 * 10 ** 17, 10 ** 15, privs[0]), st.transfer(2, 2, 5 * 10 ** 17, 2 * 10 ** 15, privs[2])]
    strs = lambda x: [strs(v) for v in x] if isinstance(x, (list, tuple)) else str(x)
    assert {f: strs(v) for f, v in st.batch_inputs(txs).items()} == fx["inputs"]
    assert str(st.tree.root) == fx["root_after"]
    c = n.RollupCircuit(fx["batch"], fx["depth"])
    wb = c.calculate_witness(fx["inputs"])                      # decimal strings, as the reference passes them
    assert [str(v) for v in c.public_signals(wb)] == fx["public_signals"]


def test_gpu_witness_program_equals_the_host_builder_signal_for_signal():
    """csrc/rollup_witness.hpp -- the program ONE GPU THREAD runs per transaction in zkr_rollup_witness_batch_device --
    compiled for the host and compared with the gadget program of rollup.cpp (which the tests above hold against the emitted
    constraint system and the pinned oracle): every signal of the witness, byte for byte, for two geometries, a self-send and
    the committed golden batch; and where the host builder refuses inputs, the program names the same statement."""
    import rollup as o
    from zkr_hip import rollup as n

    def gadget_witness(c, inputs):  # zkr_rollup_witness with its fast path (this very program on host threads) switched off
        os.environ["ZKR_WITNESS_GADGETS"] = "1"
        try:
            return c.calculate_witness(inputs)
        finally:
            del os.environ["ZKR_WITNESS_GADGETS"]

    for (batch, depth, seed, self_send) in ((2, 6, 41, False), (2, 6, 43, True), (4, 3, 7, False), (1, 3, 9, False)):
        c = n.RollupCircuit(batch, depth)
        txs, tree, _ = scenario(batch, depth, seed, self_send, n_accounts=5 if depth > 2 else 4)
        want = gadget_witness(c, as_inputs(txs))
        got, stmt, _tx = c.witness_program_host(as_inputs(txs))
        assert stmt is None and got == want, (batch, depth, seed)
        assert c.calculate_witness(as_inputs(txs)) == want                      # the fast path: (transaction, part) tasks on threads
        assert ints(got)[1] == tree.root
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "rollup_tx.json")))
    c = n.RollupCircuit(fx["batch"], fx["depth"])
    got, stmt, _tx = c.witness_program_host(fx["inputs"])
    assert stmt is None and got == gadget_witness(c, fx["inputs"]) == c.calculate_witness(fx["inputs"]) and [str(v) for v in c.public_signals(got)] == fx["public_signals"]
    c = n.RollupCircuit(2, 3)
    for what, needle in (("signature", "signature"), ("nonce", "nonce"), ("sender_path", "sender leaf"), ("recipient_leaf", "recipient leaf"),
                         ("intermediate_root", "intermediate root"), ("chain", "previous one"), ("big_s", "subgroup order"), ("index", "fits the tree")):
        txs, _, _ = scenario(2, 3, 21)
        t = txs[1]
        if what == "signature":
            t["txData"][7] = (t["txData"][7] + 1) % o.SUBORDER
        elif what == "nonce":
            t["txSenderNonce"] += 1
        elif what == "sender_path":
            t["txSenderPathElements"][1] += 1
        elif what == "recipient_leaf":
            t["txRecipientBalance"] += 1
        elif what == "intermediate_root":
            t["intermediateBalanceTreeRoot"] += 1
        elif what == "chain":
            txs = [txs[0], dict(txs[0])]
        elif what == "big_s":
            t["txData"][7] += o.SUBORDER
        elif what == "index":
            t["txData"][0] += 8
        _w, stmt, tx = c.witness_program_host(as_inputs(txs))
        assert stmt is not None and needle in stmt and tx == 1, (what, stmt, tx)
        with pytest.raises(Exception) as e:
            c.calculate_witness(as_inputs(txs))
        assert stmt in str(e.value), (what, stmt, str(e.value))


def test_randomly_damaged_inputs_are_refused_or_give_a_satisfying_witness(monkeypatch):
    """ADVICE r3 (rollup.cpp fast path): zkr_rollup_witness returns the value program's witness whenever that program's own
    statement checks pass -- so the list of checks has to be complete.  Random single-signal damage to valid circuit inputs:
    every call either refuses (ZKR_ERR_UNSATISFIED, or ZKR_ERR_ARG for a value >= r) or returns a witness that satisfies the
    EMITTED constraint system -- checked twice: inside the library (ZKR_WITNESS_VERIFY=1 evaluates the system on the
    fast-path result) and here with Python integers."""
    import rollup as o
    import zkr_hip
    from zkr_hip import rollup as n
    monkeypatch.setenv("ZKR_WITNESS_VERIFY", "1")
    c = n.RollupCircuit(2, 3)
    nv, npub, cons = parse_r1cs(c.r1cs())
    rnd = random.Random(20261003)
    refused = accepted = 0
    for trial in range(48):
        txs, _, _ = scenario(2, 3, 300 + trial % 6)
        flat = list(c.flatten_inputs(as_inputs(txs)))
        k = rnd.randrange(len(flat))
        how = rnd.randrange(4)
        if how == 0:
            flat[k] = (flat[k] + 1) % o.R
        elif how == 1:
            flat[k] = rnd.randrange(o.R)
        elif how == 2:
            flat[k] = flat[k] ^ (1 << rnd.randrange(250))
        else:
            flat[k] = flat[rnd.randrange(len(flat))]
        try:
            wb = c.calculate_witness(flat)
        except zkr_hip.ZkrError as e:
            assert e.code in (-7, -5), str(e)
            assert "internal" not in str(e), str(e)          # the fast path never hands out what the system rejects
            refused += 1
            continue
        w = ints(wb)
        assert first_violated(cons, w, o.R) == -1, "trial %d: accepted inputs whose witness violates the system" % trial
        accepted += 1
    assert refused >= 30 and refused + accepted == 48
