"""Extracts the reference's own known-answer data for the rollup circuit's witness-side crypto (SURVEY 8(c) iii, iv)
into tests/golden/rollup_kat.json.  Run in the authoring container only (reads /root/reference):

  * `mimcsponge_push32`: every PUSH32 operand of the generated MiMCSponge-220 contract
    contracts/build/contracts/CircomLib.json (the field modulus, then one word per round 1..218: the round constants
    before reduction mod r), in code order;
  * `keypairs`: the fixed BabyJub (private, public) pairs of scripts/index.js:108-118.
"""
import json
import os
import re

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

bc = bytes.fromhex(json.load(open(os.path.join(REF, "contracts/build/contracts/CircomLib.json")))["bytecode"][2:])
push, i = [], 0
while i < len(bc):
    op = bc[i]
    if 0x60 <= op <= 0x7F:
        n = op - 0x5F
        if n == 32:
            push.append("0x" + bc[i + 1:i + 33].hex())
        i += 1 + n
    else:
        i += 1

js = open(os.path.join(REF, "scripts/index.js")).read()
pairs = []
for name in ("A", "B"):
    priv = re.search(r"const priv%s = (\d+)n;" % name, js).group(1)
    pub = re.search(r"const pub%s = \[\s*(\d+)n,\s*(\d+)n\s*\];" % name, js).groups()
    pairs.append({"priv": priv, "pub": list(pub)})

json.dump({"source": "contracts/build/contracts/CircomLib.json (bytecode PUSH32 operands); scripts/index.js:108-118",
           "mimcsponge_push32": push, "keypairs": pairs}, open(os.path.join(HERE, "rollup_kat.json"), "w"), indent=1)
print(len(push), "PUSH32 words,", len(pairs), "key pairs")
