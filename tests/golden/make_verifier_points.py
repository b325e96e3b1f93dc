"""Extract the verifying-key constants baked into the reference's generated verifier contracts
into tests/golden/verifier_points.json (data only: decimal coordinates).

Run in the authoring container (needs /root/reference):  python tests/golden/make_verifier_points.py
Sources: /root/reference/contracts/contracts/TxVerifier.sol:24-35,176-257 and
         /root/reference/contracts/contracts/WithdrawVerifier.sol (same positions).
Solidity G2 limb order is [im, re] (TxVerifier.sol:18-22); kept as written, flagged in the JSON.
"""
import json
import os
import re

REF = "/root/reference/contracts/contracts"
out = {"g2_limb_order": "solidity [im, re]", "contracts": {}}
for name in ("TxVerifier", "WithdrawVerifier"):
    src = open(os.path.join(REF, name + ".sol")).read()
    body = src[src.index("function verifyingKey()"):src.index("function verify(")]
    g1 = re.findall(r"vk\.(alfa1|IC\[\d+\]) = Pairing\.G1Point\((\d+),\s*(\d+)\)", body)
    g2 = re.findall(r"vk\.(beta2|gamma2|delta2) = Pairing\.G2Point\(\[(\d+),\s*(\d+)\], \[(\d+),\s*(\d+)\]\)", body)
    gen = re.search(r"return G2Point\(\s*\[(\d+),\s*(\d+)\],\s*\[(\d+),\s*(\d+)\]", src)
    out["contracts"][name] = {
        "g1": {k: [x, y] for k, x, y in g1},
        "g2": {k: [[a, b], [c, d]] for k, a, b, c, d in g2},
        "g2_generator": [[gen.group(1), gen.group(2)], [gen.group(3), gen.group(4)]],
        "n_inputs": int(re.findall(r"uint\[(\d+)\] memory input\s*\)", src)[-1]),
    }
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "verifier_points.json")
json.dump(out, open(path, "w"), indent=1)
print({k: (len(v["g1"]), len(v["g2"])) for k, v in out["contracts"].items()})
