"""Generate tests/golden/synth_*.json with the Python oracle (oracle/groth16.py).

Run:  python tests/golden/make_golden.py
No reference implementation of the proving path can run here (websnark/snarkjs are un-vendored npm
dependencies), so these are oracle-generated known answers: seeded circuit, seeded
toxic waste, fixed blinding -> sha256 of the binarify.ts byte layouts, first h coefficients and the
proof, plus the toxic-waste closed form (no MSM/NTT) and the pairing verdict.  They pin the oracle
against regressions and give the GPU tests fixed targets.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import groth16 as g  # noqa: E402

CASES = [dict(log_m=5, p=3), dict(log_m=7, p=7), dict(log_m=10, p=73)]

for case in CASES:
    m, p = 1 << case["log_m"], case["p"]
    circ = g.synth_circuit(m, p, 0x5A4B0001)
    tox = g.toxic_from_seed(0x5A4B00FF)
    pk, vk = g.setup(circ, tox)
    pkb = g.binarify_proving_key(g.to_json_key(pk))
    w = circ["witness"]
    wb = g.binarify_witness(w)
    rng = g.SplitMix64(0xB11D + case["log_m"])
    r, s = rng.fr(), rng.fr()
    h = g.calc_h_websnark(pk, w)
    proof = g.prove(pk, w, r, s, "websnark") if case["log_m"] <= 7 else g.proof_from_toxic(circ, tox, w, r, s)
    closed = g.proof_from_toxic(circ, tox, w, r, s)
    assert proof == closed
    out = dict(log_m=case["log_m"], n_public=p, circuit_seed=0x5A4B0001, toxic_seed=0x5A4B00FF, nVars=circ["nVars"],
               r=str(r), s=str(s), pk_bin_sha256=g.sha256(pkb), pk_bin_len=len(pkb), witness_bin_sha256=g.sha256(wb),
               h_first4=[str(x) for x in h[:4]], h_sha256=g.sha256(b"".join(x.to_bytes(32, "little") for x in h)),
               proof=g.proof_to_json(proof), proof_bytes_hex=g.proof_bytes(proof).hex(),
               solidity_proof=g.solidity_proof(g.proof_to_json(proof), w[1:p + 1]),
               is_valid=g.is_valid(vk, proof, w[1:p + 1]))
    assert out["is_valid"]
    path = os.path.join(HERE, "synth_m%d.json" % case["log_m"])
    json.dump(out, open(path, "w"), indent=1)
    print(path, out["pk_bin_sha256"][:16], out["proof_bytes_hex"][:16])
