"""The two BASELINE.json configurations no other test reaches at their real shape ("What's missing" 1):

configs[4]  2^24-constraint synthetic R1CS (random A/B/C), single proof -- the L = 24 NTT plan (two strided passes),
            16.7 M-point G2 MSM, a 78 GB key arena;
configs[3]  a batch of 64 independent 2^20 rollup proofs sharded 8 per GPU -- here the eight shards run one after the
            other on the one GPU of the test box, through the same sharding / batch code the eight ranks run.

Both are checked through the size-independent property the domain offers: every proof must equal the toxic-waste closed
form (no MSM, no NTT: field dot products + three fixed-base multiplications, oracle/groth16.proof_from_aux) bit for bit
and satisfy the pairing equation of TxVerifier.sol:258-276.  Call site of the path: operator/src/snarks/common.ts:29.
"""
import pytest

import coracle
import groth16 as g

pytestmark = pytest.mark.gpu


def _proof_points(pb):
    v = [int.from_bytes(pb[32 * i:32 * i + 32], "little") for i in range(8)]
    return dict(pi_a=(v[0], v[1]), pi_b=((v[2], v[3]), (v[4], v[5])), pi_c=(v[6], v[7]))


def test_config4_dense_2_24_single_proof_equals_closed_form():
    import zkr_hip
    log_m, p = 24, 73
    zkr_hip.synth_set_shape(1)
    try:
        key, wb, aux = zkr_hip.ProvingKey.synth(log_m, p, 0x5A4B0001, 0x5A4B00FF)
    finally:
        zkr_hip.synth_set_shape(0)
    info = key.info()
    assert info["domainSize"] == 1 << log_m and info["nVars"] == 1 << log_m and info["nPublic"] == p
    assert info["ptsB2"] >= 0.75 * info["nVars"]          # dense shape: (almost) no infinity points, the G2 stress case
    assert key.arena()[1] > 40 << 30                       # far beyond the websnark format's 4 GiB u32 offsets
    rng = g.SplitMix64(2424)
    r, s = rng.fr(), rng.fr()
    proof = key.prove(wb, r, s)
    expect, vk, pub = g.proof_from_aux(aux, wb, p, r, s, dot=coracle.fr_dot)
    assert proof == g.proof_bytes(expect)
    assert g.is_valid(vk, _proof_points(proof), pub)                     # the oracle's independent pairing check
    vk_bin = key.synth_vk(aux)
    assert vk_bin == zkr_hip.binarify_verifying_key(vk)
    bad = list(pub)
    bad[7] = (bad[7] + 1) % g.R
    assert zkr_hip.verify(vk_bin, proof, pub) is True and zkr_hip.verify(vk_bin, proof, bad) is False
    # h alone at this size: linearity of the quotient's top coefficient is not available, but the NTT plan is also the
    # one zkr_ntt takes -- a forward / inverse round trip over 2^24 elements returns the input
    block = b"".join(rng.fr().to_bytes(32, "little") for _ in range(1 << 12))
    data = block * (1 << (log_m - 12))
    assert zkr_hip.ntt(zkr_hip.ntt(data), inverse=True) == data


def test_config3_batch_of_64_proofs_in_eight_shards_of_eight():
    import zkr_hip
    from zkr_hip import batch
    log_m, p, count, world = 20, 73, 64, 8
    key, w0, aux = zkr_hip.ProvingKey.synth(log_m, p, 0x5A4B0001, 0x5A4B00FF)
    vk_bin = key.synth_vk(aux)
    witnesses = [w0] + [zkr_hip.synth_witness(log_m, p, 0x5A4B0001, 0x5A4B0001 + 1000 + i) for i in range(1, count)]
    assert len(set(w[32 * 80:32 * 84] for w in witnesses)) == count      # 64 different statements
    rng = g.SplitMix64(6464)
    blinding = [(rng.fr(), rng.fr()) for _ in range(count)]
    merged = {}
    for rank in range(world):                               # what rank `rank` of the 8-GPU job runs (bench.py --gpus 8)
        mine = batch.shard_indices(count, rank, world)
        assert len(mine) == count // world
        local = batch.prove_batch(key, [w if i in mine else None for i, w in enumerate(witnesses)], blinding, rank, world)
        assert sorted(local) == mine
        merged.update(local)
    assert sorted(merged) == list(range(count))
    pubs = []
    for i in range(count):
        expect, _, pub = g.proof_from_aux(aux, witnesses[i], p, blinding[i][0], blinding[i][1], dot=coracle.fr_dot)
        assert merged[i] == g.proof_bytes(expect), "proof %d differs from the closed form" % i
        pubs.append(pub)
    assert zkr_hip.verify_batch(vk_bin, [merged[i] for i in range(count)], pubs) is True
    assert zkr_hip.verify_batch(vk_bin, [merged[(i + 1) % count] for i in range(count)], pubs) is False   # proofs shifted against their statements
