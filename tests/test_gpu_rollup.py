"""GPU: SURVEY 8(f-3) end to end -- the reference's tx circuit (tx.circom:3 = BatchProcessTx(2, 6)) set up, proved and
verified without circom / snarkjs: native constraint system -> zkr_setup_r1cs -> native witness -> HIP prover ->
native verifier, with the proof bit-identical to the oracle's closed form for fixed (toxic waste, r, s)."""
import pytest

import groth16 as g
import rollup as o
from test_rollup import as_inputs, first_violated, ints, parse_r1cs, scenario

pytestmark = pytest.mark.gpu


def _circ_dict(c, cons):
    return dict(nVars=c.n_vars, nPublic=c.n_public, nConstraints=c.n_constraints, domainSize=g.domain_size(c.n_constraints, c.n_public), rows=cons)


def test_tx_circuit_proof_matches_closed_form_and_verifies():
    import zkr_hip
    from zkr_hip import rollup as n
    c = n.RollupCircuit()                                   # tx.circom:3
    r1cs = c.r1cs()
    tox = g.toxic_from_seed(0x5A4B00F3)
    key, vk_bin = zkr_hip.ProvingKey.setup_r1cs(r1cs, toxic=[tox[k] for k in ("t", "alfa", "beta", "gamma", "delta")])
    info = key.info()
    assert info["domainSize"] == 1 << 17 and info["nVars"] == c.n_vars and info["nPublic"] == 73
    txs, tree, _ = scenario(2, 6, 41, n_accounts=5)
    wb = c.calculate_witness(as_inputs(txs))
    pub = c.public_signals(wb)
    assert pub == o.batch_public_signals(txs) and pub[0] == tree.root
    rng = g.SplitMix64(99)
    r, s = rng.fr(), rng.fr()
    proof = key.prove(wb, r, s)
    # the operator's checks (operator/src/snarks/common.ts:30-38, withdrawverifier.test.ts:27-37,56-65)
    assert zkr_hip.verify(vk_bin, proof, pub) is True
    bad = list(pub)
    bad[0] = (bad[0] + 1) % o.R                             # another new root
    assert zkr_hip.verify(vk_bin, proof, bad) is False
    sp = zkr_hip.solidity_proof(zkr_hip.proof_json_from_bytes(proof), pub)
    assert sp["inputs"] == [str(v) for v in pub] and len(sp["inputs"]) == 73      # TxVerifier.sol:281 uint[73]
    # bit-exact: the closed form from the toxic waste over the same constraint system and witness
    _, _, cons = parse_r1cs(r1cs)
    w = ints(wb)
    assert proof == g.proof_bytes(g.proof_from_toxic(_circ_dict(c, cons), tox, w, r, s))
    # a second batch on the same key (the operator proves batch after batch against one key)
    txs2, tree2, _ = scenario(2, 6, 42, self_send=True, n_accounts=3)
    wb2 = c.calculate_witness(as_inputs(txs2))
    p2 = key.prove(wb2)
    assert zkr_hip.verify(vk_bin, p2, c.public_signals(wb2)) and not zkr_hip.verify(vk_bin, p2, pub)


def test_tx_circuit_through_the_websnark_buffer_like_the_unchanged_reference_caller():
    """The reference's own data flow for its own circuit: setup (prover/package.json:34) -> provingKeyBin in the
    binarify.ts:143-206 layout (zkr_setup_r1cs_websnark) -> createProofGenerator (common.ts:10-53) with a new Bn128
    per proof -> proof == the device-key proof == the closed form, accepted by the verifier, Solidity-shaped output with
    73 inputs (TxVerifier.sol:281)."""
    import zkr_hip
    from zkr_hip import rollup as n
    c = n.RollupCircuit()
    r1cs = c.r1cs()
    tox = g.toxic_from_seed(0x5A4B00F3)
    tl = [tox[k] for k in ("t", "alfa", "beta", "gamma", "delta")]
    pkb, vk_bin = zkr_hip.setup_r1cs_websnark(r1cs, toxic=tl)
    hdr = [int.from_bytes(pkb[4 * i:4 * i + 4], "little") for i in range(10)]
    assert hdr[0] == c.n_vars and hdr[1] == 73 and hdr[2] == 1 << 17 and hdr[3] == 488
    assert len(pkb) == hdr[9] + 64 * hdr[2]                                  # binarify.ts:115-141 size formula, last section
    key, vk_bin2 = zkr_hip.ProvingKey.setup_r1cs(r1cs, toxic=tl)
    assert vk_bin2 == vk_bin
    txs, tree, _ = scenario(2, 6, 41, n_accounts=5)
    wb = c.calculate_witness(as_inputs(txs))
    w = ints(wb)
    rng = g.SplitMix64(99)
    r, s = rng.fr(), rng.fr()
    zkr_hip.clear_key_cache()
    loads0 = zkr_hip.key_cache_stats["loads"]
    gen = zkr_hip.create_proof_generator(pkb, zkr_hip.verifying_key_from_bytes(vk_bin), n_public=73)
    out = gen(w, r, s)
    out2 = gen(w, r, s)                                                        # the operator proves batch after batch
    assert zkr_hip.key_cache_stats["loads"] - loads0 == 1
    pb = zkr_hip.proof_bytes_from_json(out["proof"])
    assert pb == key.prove(wb, r, s) and out2 == out
    _, _, cons = parse_r1cs(r1cs)
    assert pb == g.proof_bytes(g.proof_from_toxic(_circ_dict(c, cons), tox, w, r, s))
    assert out["solidityProof"]["inputs"] == [str(v) for v in c.public_signals(wb)] and len(out["solidityProof"]["inputs"]) == 73
    assert out["solidityProof"]["b"][0] == [out["proof"]["pi_b"][0][1], out["proof"]["pi_b"][0][0]]     # (im, re), TxVerifier.sol:18-22
    zkr_hip.clear_key_cache()


def test_batch_of_four_depth_five_like_the_reference_test():
    """prover/__tests__/batchprocesstx.test.ts:247-253 runs BatchProcessTx(4, 5); here with a proof on top
    (fresh toxic waste from the OS CSPRNG)."""
    import zkr_hip
    from zkr_hip import rollup as n
    c = n.RollupCircuit(4, 5)
    key, vk_bin = zkr_hip.ProvingKey.setup_r1cs(c.r1cs())
    txs, tree, _ = scenario(4, 5, 43, n_accounts=6)
    wb = c.calculate_witness(as_inputs(txs))
    pub = c.public_signals(wb)
    assert pub[0] == tree.root
    proof = key.prove(wb)
    assert zkr_hip.verify(vk_bin, proof, pub)
    assert not zkr_hip.verify(vk_bin, proof, [pub[0]] + [pub[2], pub[1]] + pub[3:])


def test_withdraw_circuit_proof_matches_closed_form_and_verifies():
    """genWithdrawVerifierProof (operator/src/snarks/withdraw.ts:6-10; contracts/__tests__/withdrawverifier.test.ts:24-65)
    with every step native: the proof verifies, its public signals are (publicKey, nullifier), a changed one fails."""
    import zkr_hip
    from zkr_hip import rollup as n
    c = n.WithdrawCircuit()
    r1cs = c.r1cs()
    tox = g.toxic_from_seed(0x5A4B00F4)
    key, vk_bin = zkr_hip.ProvingKey.setup_r1cs(r1cs, toxic=[tox[k] for k in ("t", "alfa", "beta", "gamma", "delta")])
    priv = 4884893312420666846728788329068334659645814268254376926932808574485660215402      # scripts/index.js:108
    wb = c.calculate_witness({"privateKey": n.format_priv_key(priv), "nullifier": 123456789})
    pub = c.public_signals(wb)
    assert pub == [1503249839729450699258568253188655641897688629727284476088280559686550009373,
                   5002035132060692260001110027450125171390191669020671174899856008595079807949, 123456789]   # index.js:109-112
    rng = g.SplitMix64(5)
    r, s = rng.fr(), rng.fr()
    proof = key.prove(wb, r, s)
    assert zkr_hip.verify(vk_bin, proof, pub)
    assert not zkr_hip.verify(vk_bin, proof, pub[:2] + [pub[2] + 1])            # withdrawverifier.test.ts:56-65
    nv, npub, cons = parse_r1cs(r1cs)
    circ = dict(nVars=nv, nPublic=npub, nConstraints=len(cons), domainSize=g.domain_size(len(cons), npub), rows=cons)
    assert proof == g.proof_bytes(g.proof_from_toxic(circ, tox, ints(wb), r, s))


def test_gpu_batch_hash_and_tree_match_the_oracle():
    """zkr_mimcsponge_multihash_batch / zkr_balance_tree_build (csrc/rollup_gpu.hip, one thread per hash) == the pinned
    oracle's multiHash and balance tree, leaf for leaf and level for level; operands >= r are taken mod r."""
    import random
    import time
    from zkr_hip import rollup as n
    rnd = random.Random(91)
    for arity in (1, 2, 4, 5):
        rows = [[rnd.randrange(o.R) for _ in range(arity)] for _ in range(37)]
        rows[3][0] = (1 << 256) - 1
        rows[5][arity - 1] = o.R + 11
        assert n.multi_hash_batch(rows) == [o.multi_hash(r) for r in rows]
    assert n.multi_hash_batch([[32767]]) == [o.multi_hash([32767])]                      # hasher.test.ts:15-26
    depth = 5
    leaves = [rnd.randrange(o.R) for _ in range(19)]
    t = n.BalanceTree.from_leaves(depth, leaves)
    ref = o.Tree(depth)
    for i, v in enumerate(leaves):
        ref.update(i, v)
    assert t.levels == ref.levels and t.root == ref.root
    assert t.path(7) == ref.path(7)
    t.update(3, 12345)
    ref.update(3, 12345)
    assert t.root == ref.root                                                           # native host hash continues the GPU-built tree
    # a 2^16-account tree: idempotence-style property at size -- the root equals the fold of the GPU-hashed level below
    big = n.BalanceTree.from_leaves(16, [rnd.randrange(o.R) for _ in range(1 << 16)])
    assert big.root == o.hash_left_right(big.levels[15][0], big.levels[15][1])
    assert big.levels[1][:4] == n.multi_hash_batch([big.levels[0][2 * i:2 * i + 2] for i in range(4)])


def test_real_circuit_at_the_headline_size():
    """BatchProcessTx(18, 6): the reference's template filled to 1 008 108 constraints (2^20 domain, 649 public signals) --
    set up on the GPU, witness built natively for 18 chained transactions, proved, and accepted by the native verifier
    (single and batch form); the new root is the operator's."""
    import zkr_hip
    from zkr_hip import rollup as n
    c = n.RollupCircuit(18, 6)
    assert c.n_constraints + c.n_public + 1 <= 1 << 20 < 2 * (c.n_constraints + c.n_public + 1)
    key, vk_bin = zkr_hip.ProvingKey.setup_r1cs(c.r1cs())
    assert key.info()["domainSize"] == 1 << 20
    privs = [0x5A4B2000 + 31 * i for i in range(6)]
    st = n.RollupState(6)
    for i, pv in enumerate(privs):
        st.deposit(i, n.gen_public_key(pv), 10 ** 20, 0)
    txs = [st.transfer(j % 6, (j + 1 + j // 6) % 6, 10 ** 17 + j, 10 ** 15, privs[j % 6]) for j in range(18)]
    wb = c.calculate_witness(st.batch_inputs(txs))
    pub = c.public_signals(wb)
    assert pub[0] == st.tree.root and len(pub) == 649
    proofs = key.prove_batch([wb, wb], rs=[3, 4], ss=[5, 6])
    assert proofs[0] != proofs[1]
    assert zkr_hip.verify(vk_bin, proofs[0], pub) and zkr_hip.verify_batch(vk_bin, proofs, [pub, pub])
    bad = list(pub)
    bad[640] = (bad[640] + 1) % o.R
    assert not zkr_hip.verify(vk_bin, proofs[0], bad) and not zkr_hip.verify_batch(vk_bin, proofs, [pub, bad])


def test_gpu_witness_builder_satisfies_the_system_under_the_oracle_and_equals_the_host_builder():
    """zkr_rollup_witness_batch_device -- the witnesses of MANY rollup batches built on the GPU, one thread
    per transaction, left in HBM -- against zkr_rollup_witness (the host gadget program, itself checked against the emitted
    constraint system and the pinned oracle in tests/test_rollup.py): every batch byte for byte, for tx.circom's geometry
    (2, 6), a self-send, another geometry in the same process (4 transactions, depth 3) and the committed golden batch
    (tests/golden/rollup_tx.json); a witness taken from HBM proves to the same bytes as the host-built one."""
    import json
    import os
    import zkr_hip
    from zkr_hip import rollup as n
    c = n.RollupCircuit(2, 6)
    flats = []
    for seed, self_send in ((41, False), (42, False), (43, True), (44, False), (45, False)):
        txs, _, _ = scenario(2, 6, seed, self_send, n_accounts=5)
        flats.append(c.flatten_inputs(as_inputs(txs)))
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rollup_tx.json")))
    cf = n.RollupCircuit(fx["batch"], fx["depth"])             # the committed batch (decimal strings, as the reference passes them)
    df = cf.calculate_witness_batch_device([fx["inputs"]])
    wf = bytes(df[0].cpu().numpy().tobytes())
    assert wf == cf.calculate_witness(fx["inputs"]) and [str(v) for v in cf.public_signals(wf)] == fx["public_signals"]
    dev = c.calculate_witness_batch_device(flats)
    assert tuple(dev.shape) == (len(flats), c.n_vars * 32)
    # FIRST the checker that is not the product: every constraint of the emitted system evaluated on
    # the DEVICE-built witness with Python integers (the oracle's field arithmetic), and its public part against the oracle's
    # own statement of the circuit's signals -- (2, 6), the self-send (index 2) and the (4, 3) geometry below
    nv, npub, cons = parse_r1cs(c.r1cs())
    assert (nv, npub) == (c.n_vars, c.n_public)
    scen = [(41, False), (42, False), (43, True), (44, False), (45, False)]
    for i, (seed, self_send) in enumerate(scen):
        w = ints(bytes(dev[i].cpu().numpy().tobytes()))
        assert w[0] == 1 and all(v < o.R for v in w)
        assert first_violated(cons, w, o.R) == -1, "batch %d: the GPU-built witness violates the constraint system" % i
        txs_i, tree_i, _ = scenario(2, 6, seed, self_send, n_accounts=5)
        assert w[1:npub + 1] == o.batch_public_signals(txs_i) and w[1] == tree_i.root
    wgold = ints(wf)
    nvf, npf, consf = parse_r1cs(cf.r1cs())
    assert first_violated(consf, wgold, o.R) == -1 and [str(v) for v in wgold[1:npf + 1]] == fx["public_signals"]
    # second: byte equality with the host builder
    host = [c.calculate_witness(f) for f in flats]
    for i, wb in enumerate(host):
        assert bytes(dev[i].cpu().numpy().tobytes()) == wb, "batch %d differs from the host builder" % i
    c43 = n.RollupCircuit(4, 3)
    txs, tree, _ = scenario(4, 3, 7)
    d43 = c43.calculate_witness_batch_device([as_inputs(txs)] * 3)
    _, np43, cons43 = parse_r1cs(c43.r1cs())
    for i in range(3):
        w43 = ints(bytes(d43[i].cpu().numpy().tobytes()))
        assert first_violated(cons43, w43, o.R) == -1 and w43[1:np43 + 1] == o.batch_public_signals(txs)
    want = c43.calculate_witness(as_inputs(txs))
    assert all(bytes(d43[i].cpu().numpy().tobytes()) == want for i in range(3)) and ints(want)[1] == tree.root
    key, vk_bin = zkr_hip.ProvingKey.setup_r1cs(c.r1cs())
    got = key.prove_batch_device([dev[i].data_ptr() for i in range(3)], [5, 6, 7], [8, 9, 10])
    assert got == [key.prove(host[i], 5 + i, 8 + i) for i in range(3)]
    assert zkr_hip.verify_batch(vk_bin, got, [c.public_signals(host[i]) for i in range(3)])


@pytest.mark.parametrize("what,needle", [
    ("signature", "signature"), ("nonce", "nonce"), ("sender_path", "sender leaf"), ("recipient_leaf", "recipient leaf"),
    ("intermediate_root", "intermediate root"), ("chain", "previous one"), ("big_s", "subgroup order"), ("index", "fits the tree")])
def test_gpu_witness_builder_refuses_what_the_host_builder_refuses(what, needle):
    """Where Circuit.calculateWitness throws (operator/src/snarks/common.ts:15-17): the GPU builder names the same statement as
    the host builder, and the batch it sits in -- here the second of three."""
    import rollup as o
    import zkr_hip
    from zkr_hip import rollup as n
    c = n.RollupCircuit(2, 3)
    good, _, _ = scenario(2, 3, 20)
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
    with pytest.raises(zkr_hip.ZkrError) as eh:
        c.calculate_witness(as_inputs(txs))
    with pytest.raises(zkr_hip.ZkrError) as eg:
        c.calculate_witness_batch_device([as_inputs(good), as_inputs(txs), as_inputs(good)])
    assert eg.value.code == eh.value.code == -7 and needle in str(eg.value) and needle in str(eh.value), (str(eg.value), str(eh.value))
    assert "batch 1: transaction 1" in str(eg.value), str(eg.value)
