"""CPU oracle for SURVEY 8(f-3): the witness-side crypto and state transition of the reference's rollup circuit.

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg): the product never imports it.

Restates, in plain Python integers:
  * MiMCSponge-220 (`multiHash`, operator/src/utils/crypto.ts:28-38) over circomlib 0.0.20's permutation
    (prover/circuits/hasher.circom:3-16 instantiates `MiMCSponge(length, 220, 1)`, key 0);
  * BabyJub key derivation and EdDSA-MiMCSponge (crypto.ts:58-84 `formatPrivKeyForBabyJub` / `genPublicKey`,
    :143-177 `sign` / `verify`; in-circuit form prover/circuits/eddsa.circom:12-110);
  * the balance tree (operator/src/utils/merkletree.ts:44-83, full-tree equivalent) and the per-transaction state
    transition the circuit enforces (prover/circuits/processtx.circom:10-193, batchprocesstx.circom:3-75).

circomlib / snarkjs are un-vendored dependencies (prover/yarn.lock: circomlib 0.0.20, snarkjs 0.1.20): their
published algorithms are restated here [DEP-KNOWLEDGE] and PINNED against the reference's own vectors:
  (i)  the 220 MiMCSponge round constants embedded as PUSH32 words in the reference's generated contract
       contracts/build/contracts/CircomLib.json (tests/golden/rollup_kat.json, field "mimcsponge_push32"), and
  (ii) the two fixed (private key -> public key) pairs of scripts/index.js:108-118, which exercise the hash, the
       key formatting quirk (hex text as bytes, crypto.ts:20-22) and BabyJub scalar multiplication end to end.
"""

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617   # crypto.ts:16-18
NROUNDS = 220                                                                          # hasher.circom:8

# ---------------------------------------------------------------- keccak-256 (for the round constants)
_RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B, 0x0000000080000001,
       0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
       0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080,
       0x000000000000800A, 0x800000008000000A, 0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
_ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]
_M64 = (1 << 64) - 1


def _rol(v, n):
    n %= 64
    return ((v << n) | (v >> (64 - n))) & _M64 if n else v


def _keccak_f(a):
    for rc in _RC:
        c = [a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4] for x in range(5)]
        d = [c[(x - 1) % 5] ^ _rol(c[(x + 1) % 5], 1) for x in range(5)]
        a = [[a[x][y] ^ d[x] for y in range(5)] for x in range(5)]
        b = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                b[y][(2 * x + 3 * y) % 5] = _rol(a[x][y], _ROT[x][y])
        a = [[b[x][y] ^ ((~b[(x + 1) % 5][y]) & b[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
        a[0][0] ^= rc
    return a


def keccak256(data: bytes) -> bytes:
    rate = 136
    q = rate - len(data) % rate
    p = bytes(data) + (b"\x81" if q == 1 else b"\x01" + bytes(q - 2) + b"\x80")
    a = [[0] * 5 for _ in range(5)]
    for off in range(0, len(p), rate):
        for i in range(rate // 8):
            a[i % 5][i // 5] ^= int.from_bytes(p[off + 8 * i:off + 8 * i + 8], "little")
        a = _keccak_f(a)
    return b"".join(a[i % 5][i // 5].to_bytes(8, "little") for i in range(4))


_cts = None


def mimc_constants():
    """circomlib mimcsponge getConstants("mimcsponge", 220): c = keccak256(seed text), then c = keccak256(c) per round,
    constant = c mod r; first and last constants are zero (contracts/migrations/2_deploy_mimcsponge.js:9-10 builds the
    on-chain hasher from the same seed and round count)."""
    global _cts
    if _cts is None:
        c = keccak256(b"mimcsponge")
        out = [0] * NROUNDS
        for i in range(1, NROUNDS):
            c = keccak256(c)
            out[i] = int.from_bytes(c, "big") % R
        out[NROUNDS - 1] = 0
        _cts = out
    return _cts


def mimc_feistel(xl, xr, k=0):
    cts = mimc_constants()
    for i in range(NROUNDS):
        t = (xl + k + cts[i]) % R
        t5 = pow(t, 5, R)
        if i < NROUNDS - 1:
            xl, xr = (xr + t5) % R, xl
        else:
            xr = (xr + t5) % R
    return xl, xr


def multi_hash(arr, key=0):
    """mimcsponge.multiHash(arr) with one output (crypto.ts:28-30); inputs are reduced mod r as the field ops do."""
    r_, c_ = 0, 0
    for v in arr:
        r_ = (r_ + int(v)) % R
        r_, c_ = mimc_feistel(r_, c_, key)
    return r_


def hash_left_right(l, r):  # crypto.ts:36-38
    return multi_hash([l, r])


# ---------------------------------------------------------------- BabyJub (twisted Edwards over Fr)
BJ_A = 168700
BJ_D = 168696
BASE8 = (5299619240641551281634865583518297030282874472190772894086521144482721001553,
         16950150798460657717958625567821834550301663161624707787222815936182638968203)   # eddsa.circom:87-90
SUBORDER = 2736030358979909402780800718157159386076813972158567259200215660948447373041   # eddsa.circom:32 (+1)


def bj_add(p, q):
    x1, y1 = p
    x2, y2 = q
    t = BJ_D * x1 * x2 % R * y1 % R * y2 % R
    x3 = (x1 * y2 + y1 * x2) * pow(1 + t, R - 2, R) % R
    y3 = (y1 * y2 - BJ_A * x1 * x2) * pow(1 - t, R - 2, R) % R
    return (x3, y3)


def bj_mul(p, e):
    acc, q = (0, 1), p
    while e:
        if e & 1:
            acc = bj_add(acc, q)
        q = bj_add(q, q)
        e >>= 1
    return acc


def bj_on_curve(p):
    x, y = p
    return (BJ_A * x * x + y * y - 1 - BJ_D * x * x % R * y * y) % R == 0


def _hex_text(i):          # bigInt2Buffer, crypto.ts:20-22: the hex digits as TEXT bytes
    return format(int(i), "x").encode()


def _prune(b):             # circomlib eddsa.pruneBuffer
    b = bytearray(b)
    b[0] &= 0xF8
    b[31] &= 0x7F
    b[31] |= 0x40
    return bytes(b)


def format_priv_key(priv):
    """formatPrivKeyForBabyJub (crypto.ts:58-76)."""
    return int.from_bytes(_prune(_hex_text(multi_hash([priv]))[:32]), "little") >> 3


def gen_public_key(priv):
    """genPublicKey (crypto.ts:78-84)."""
    assert 0 <= priv < R
    return bj_mul(BASE8, format_priv_key(priv))


def sign(priv, msg):
    """sign (crypto.ts:143-168): returns (R8x, R8y, S)."""
    m = multi_hash(msg)
    h1 = _hex_text(multi_hash([priv]))
    s = int.from_bytes(_prune(h1[:32]), "little")
    a = bj_mul(BASE8, s >> 3)
    rb = _hex_text(multi_hash([int.from_bytes(h1[32:64] + m.to_bytes(32, "little"), "big")]))
    r = int.from_bytes(rb, "little") % SUBORDER
    r8 = bj_mul(BASE8, r)
    hm = multi_hash([r8[0], r8[1], a[0], a[1], m])
    return (r8[0], r8[1], (r + hm * s) % SUBORDER)


def verify_msg_hash(m, sig, pub):
    """circomlib eddsa.verifyMiMCSponge(msgHash, sig, pubKey) == what eddsa.circom:12-110 enforces."""
    r8 = (sig[0], sig[1])
    if not (bj_on_curve(r8) and bj_on_curve(pub)) or sig[2] >= SUBORDER:
        return False
    hm = multi_hash([r8[0], r8[1], pub[0], pub[1], m])
    return bj_mul(BASE8, sig[2]) == bj_add(r8, bj_mul(bj_mul(pub, 8), hm))


def verify(msg, sig, pub):  # crypto.ts:170-177
    return verify_msg_hash(multi_hash(msg), sig, pub)


# ---------------------------------------------------------------- balance tree + state transition
class Tree:
    """merkletree.ts:44-83 as a full tree: `depth` hash levels, 2^depth leaves, empty leaf = zero value."""

    def __init__(self, depth, zero=0):
        self.depth = depth
        self.levels = [[zero] * (1 << depth)]
        for _ in range(depth):
            p = self.levels[-1]
            self.levels.append([hash_left_right(p[2 * i], p[2 * i + 1]) for i in range(len(p) // 2)])

    @property
    def root(self):
        return self.levels[-1][0]

    def update(self, idx, leaf):
        self.levels[0][idx] = leaf
        for l in range(self.depth):
            idx >>= 1
            self.levels[l + 1][idx] = hash_left_right(self.levels[l][2 * idx], self.levels[l][2 * idx + 1])

    def path(self, idx):
        """getUpdatePath(idx).pathElements (sibling per level, leaf level first)."""
        out = []
        for l in range(self.depth):
            out.append(self.levels[l][idx ^ 1])
            idx >>= 1
        return out


def leaf_hash(pub, balance, nonce):  # helpers.ts:80-82
    return multi_hash([pub[0], pub[1], balance, nonce])


def process_tx_inputs(tree, accounts, frm, to, amount, fee, priv):
    """One transaction against `tree` (updated in place) and `accounts` = {idx: [pubx, puby, balance, nonce]}:
    returns the circuit inputs of ProcessTx (processtx.circom:13-67, the flow of prover/__tests__/processtx.test.ts:24-130)
    as a dict of ints / lists."""
    sa, ra = accounts[frm], accounts[to]
    nonce = sa[3] + 1
    sig = sign(priv, [frm, to, amount, fee, nonce])
    inp = dict(balanceTreeRoot=tree.root, txData=[frm, to, amount, fee, nonce, sig[0], sig[1], sig[2]],
               txSenderPublicKey=sa[:2], txSenderBalance=sa[2], txSenderNonce=sa[3], txSenderPathElements=tree.path(frm),
               txRecipientPublicKey=ra[:2], txRecipientBalance=ra[2], txRecipientNonce=ra[3], txRecipientPathElements=tree.path(to))
    sa[2] -= amount + fee
    sa[3] = nonce
    tree.update(frm, leaf_hash(sa[:2], sa[2], sa[3]))
    inp["intermediateBalanceTreeRoot"] = tree.root
    inp["intermediateBalanceTreePathElements"] = tree.path(to)
    ra = accounts[to]                         # same list when frm == to (processtx.circom:141-159)
    ra[2] += amount
    tree.update(to, leaf_hash(ra[:2], ra[2], ra[3]))
    inp["newBalanceTreeRoot"] = tree.root
    return inp


TX_FIELDS = ("balanceTreeRoot", "txData", "txSenderPublicKey", "txSenderBalance", "txSenderNonce", "txSenderPathElements",
             "txRecipientPublicKey", "txRecipientBalance", "txRecipientNonce", "txRecipientPathElements",
             "intermediateBalanceTreeRoot", "intermediateBalanceTreePathElements")   # batchprocesstx.circom:13-36, declaration order


def batch_public_signals(txs):
    """Public signals of BatchProcessTx(batch, depth) in circom's order: the output, then every input array in
    declaration order, each flattened over the batch (batchprocesstx.circom:10-36; 73 values for (2, 6))."""
    out = [txs[-1]["newBalanceTreeRoot"]]
    for f in TX_FIELDS:
        for t in txs:
            v = t[f]
            out.extend(v if isinstance(v, list) else [v])
    return [int(x) % R for x in out]
