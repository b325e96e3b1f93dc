"""Host side of the reference's rollup circuit without circom / snarkjs, over the C ABI.

Mirrors /root/reference/operator/src/utils/crypto.ts (multiHash :28-30, genPublicKey :78-84, sign :143-168,
verify :170-177) and the `compiler(tx.circom)` + `Circuit.calculateWitness` steps of
operator/src/snarks/common.ts:12-21 for `BatchProcessTx(batch, depth)`
(prover/circuits/batchprocesstx.circom:3-75).  All arithmetic is native (csrc/rollup.cpp).
"""
import ctypes

from .binding import ZkrError, _check, _take, lib

SNARK_FIELD_SIZE = 21888242871839275222246405745257275088548364400416034343698204186575808495617  # crypto.ts:16-18
TX_INPUT_FIELDS = ("balanceTreeRoot", "txData", "txSenderPublicKey", "txSenderBalance", "txSenderNonce", "txSenderPathElements",
                   "txRecipientPublicKey", "txRecipientBalance", "txRecipientNonce", "txRecipientPathElements",
                   "intermediateBalanceTreeRoot", "intermediateBalanceTreePathElements")   # batchprocesstx.circom:13-36


def _le(v):
    return int(v).to_bytes(32, "little")


def _ints(b):
    return [int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)]


def multi_hash(values) -> int:
    """multiHash(d) (crypto.ts:28-30); operands of any size are taken mod r as the reference's field operations do."""
    buf = b"".join(_le(int(v) % (1 << 256)) for v in values)
    assert all(0 <= int(v) < (1 << 256) for v in values)
    out = ctypes.create_string_buffer(32)
    _check(lib().zkr_mimcsponge_multihash(buf, len(values), out))
    return int.from_bytes(out.raw, "little")


def multi_hash_batch(rows, device=0):
    """multiHash of every row (equal lengths) on the GPU, one thread per hash (zkr_mimcsponge_multihash_batch)."""
    rows = [list(r) for r in rows]
    if not rows:
        return []
    arity = len(rows[0])
    assert all(len(r) == arity for r in rows)
    buf = b"".join(_le(int(v) % (1 << 256)) for r in rows for v in r)
    out = ctypes.create_string_buffer(32 * len(rows))
    _check(lib().zkr_mimcsponge_multihash_batch(buf, len(rows), arity, out, device))
    return _ints(out.raw)


def hash_left_right(left, right) -> int:   # crypto.ts:36-38
    return multi_hash([left, right])


def gen_public_key(priv):
    """genPublicKey(privKey) (crypto.ts:78-84) -> (x, y)."""
    out = ctypes.create_string_buffer(64)
    _check(lib().zkr_babyjub_pubkey(_le(priv), out))
    return tuple(_ints(out.raw))


def format_priv_key(priv) -> int:
    """formatPrivKeyForBabyJub(privKey) (crypto.ts:58-76): the scalar the circuits take as `privateKey`."""
    out = ctypes.create_string_buffer(32)
    _check(lib().zkr_babyjub_format_privkey(_le(priv), out))
    return int.from_bytes(out.raw, "little")


def sign(priv, msg):
    """sign(prv, msg) (crypto.ts:143-168) -> {"R8": (x, y), "S": s}."""
    buf = b"".join(_le(v) for v in msg)
    out = ctypes.create_string_buffer(96)
    _check(lib().zkr_eddsa_sign(_le(priv), buf, len(msg), out))
    v = _ints(out.raw)
    return {"R8": (v[0], v[1]), "S": v[2]}


def verify(msg, sig, pub) -> bool:
    """verify(msg, sig, pubKey) (crypto.ts:170-177)."""
    buf = b"".join(_le(v) for v in msg)
    sb = _le(sig["R8"][0]) + _le(sig["R8"][1]) + _le(sig["S"])
    pb = _le(pub[0]) + _le(pub[1])
    ok = ctypes.c_int()
    _check(lib().zkr_eddsa_verify(buf, len(msg), sb,
                                  pb, ctypes.byref(ok)))
    return bool(ok.value)


class RollupCircuit:
    """BatchProcessTx(batch, depth): what `new Circuit(await compiler("tx.circom"))` is to the reference
    (common.ts:12-15).  `r1cs()` feeds ProvingKey.setup_r1cs; `calculate_witness(inputs)` returns the witness as
    binarifyWitness would lay it out (nVars x 32 B) and raises ZkrError(-7) where calculateWitness would throw."""

    def __init__(self, batch=2, depth=6):   # tx.circom:3
        self.batch, self.depth = batch, depth
        nv, npub, nc = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
        _check(lib().zkr_rollup_info(batch, depth, ctypes.byref(nv), ctypes.byref(npub), ctypes.byref(nc)))
        self.n_vars, self.n_public, self.n_constraints = nv.value, npub.value, nc.value

    def r1cs(self) -> bytes:
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        _check(lib().zkr_rollup_r1cs(self.batch, self.depth, ctypes.byref(p), ctypes.byref(n)))
        return _take(p, n.value)

    def flatten_inputs(self, inputs):
        """The circuitInputs object of the reference (one entry per input signal of BatchProcessTx, arrays over the
        batch; batchprocesstx.test.ts builds it) -> the flat list in signal order."""
        flat = []

        def walk(v):
            if isinstance(v, (list, tuple)):
                for x in v:
                    walk(x)
            else:
                flat.append(int(v) % SNARK_FIELD_SIZE)
        for f in TX_INPUT_FIELDS:
            walk(inputs[f])
        if len(flat) != self.n_public - 1:
            raise ZkrError(-5, "circuit inputs have %d values, BatchProcessTx(%d, %d) takes %d" % (len(flat), self.batch, self.depth, self.n_public - 1))
        return flat

    def calculate_witness(self, inputs) -> bytes:
        flat = self.flatten_inputs(inputs) if isinstance(inputs, dict) else [int(v) for v in inputs]
        buf = b"".join(_le(v) for v in flat)
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        _check(lib().zkr_rollup_witness(self.batch, self.depth, buf, len(flat), ctypes.byref(p), ctypes.byref(n)))
        return _take(p, n.value)

    def public_signals(self, witness_bin: bytes):
        """witness.slice(1, nPubInputs + nOutputs + 1) (common.ts:18-21)."""
        return _ints(witness_bin[32:32 * (self.n_public + 1)])

    def witness_program_host(self, inputs):
        """The GPU builder's per-transaction program (csrc/rollup_witness.hpp) run on the host for one batch (test hook):
        (witness bytes, statement text of the first violation or None, its transaction)."""
        flat = self.flatten_inputs(inputs) if isinstance(inputs, dict) else [int(v) for v in inputs]
        buf = b"".join(_le(v) for v in flat)
        out = ctypes.create_string_buffer(self.n_vars * 32)
        stmt, tx = ctypes.c_uint32(), ctypes.c_uint32()
        _check(lib().zkr_rollup_witness_program_host(self.batch, self.depth, buf, len(flat), out, ctypes.byref(stmt), ctypes.byref(tx)))
        return out.raw, (lib().zkr_rollup_statement_text(stmt.value).decode() if stmt.value else None), tx.value

    def calculate_witness_batch_device(self, inputs_list, device=0):
        """calculateWitness for MANY rollup batches on the GPU (zkr_rollup_witness_batch_device: one thread per transaction):
        a uint8 torch tensor [len(inputs_list), nVars * 32] in HBM, row i = the bytes calculate_witness(inputs_list[i])
        returns, ready for ProvingKey.prove_batch_device([t[i].data_ptr() ...]).  Raises ZkrError(-7) naming the first batch,
        transaction and statement that fails."""
        import torch
        flats = [self.flatten_inputs(x) if isinstance(x, dict) else [int(v) for v in x] for x in inputs_list]
        for f in flats:
            if len(f) != self.n_public - 1:
                raise ZkrError(-5, "circuit inputs have %d values, BatchProcessTx(%d, %d) takes %d" % (len(f), self.batch, self.depth, self.n_public - 1))
        buf = b"".join(_le(v) for f in flats for v in f)
        out = torch.empty((len(flats), self.n_vars * 32), dtype=torch.uint8, device=torch.device("cuda", device))   # an allocation: nothing to wait for
        _check(lib().zkr_rollup_witness_batch_device(self.batch, self.depth, buf, self.n_public - 1, len(flats), out.data_ptr(), device))
        return out


class WithdrawCircuit:
    """Withdraw() (prover/circuits/withdraw.circom:4-25): public signals publicKey[0], publicKey[1], nullifier."""
    n_public = 3

    def r1cs(self) -> bytes:
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        _check(lib().zkr_withdraw_r1cs(ctypes.byref(p), ctypes.byref(n)))
        return _take(p, n.value)

    def calculate_witness(self, inputs) -> bytes:
        """inputs = {"privateKey": formatted key, "nullifier": ...} (withdraw.test.ts:22-25)."""
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        _check(lib().zkr_withdraw_witness(_le(int(inputs["privateKey"])), _le(int(inputs["nullifier"]) % SNARK_FIELD_SIZE), ctypes.byref(p), ctypes.byref(n)))
        return _take(p, n.value)

    def public_signals(self, witness_bin: bytes):
        return _ints(witness_bin[32:128])


class BalanceTree:
    """operator/src/utils/merkletree.ts:14-83 as a full binary tree over the native hash: `depth` levels,
    2^depth leaves, empty leaf = zero value; `path(i)` = getUpdatePath(i).pathElements."""

    def __init__(self, depth, zero=0):
        self.depth = depth
        self.levels = [[zero] * (1 << depth)]
        z = zero
        for _ in range(depth):                       # an empty tree needs one hash per level
            z = hash_left_right(z, z)
            self.levels.append([z] * (len(self.levels[-1]) // 2))

    @classmethod
    def from_leaves(cls, depth, leaves, zero=0, device=0):
        """The tree over `leaves` (padded with the zero value) built on the GPU (zkr_balance_tree_build)."""
        t = cls.__new__(cls)
        t.depth = depth
        n = 1 << depth
        full = [int(v) for v in leaves] + [zero] * (n - len(leaves))
        out = ctypes.create_string_buffer(32 * (2 * n - 1))
        _check(lib().zkr_balance_tree_build(b"".join(_le(v) for v in full), depth, out, device))
        vals, t.levels, off = _ints(out.raw), [], 0
        for l in range(depth + 1):
            t.levels.append(vals[off:off + (n >> l)])
            off += n >> l
        return t

    @property
    def root(self):
        return self.levels[-1][0]

    def update(self, idx, leaf):
        self.levels[0][idx] = leaf
        for l in range(self.depth):
            idx >>= 1
            self.levels[l + 1][idx] = hash_left_right(self.levels[l][2 * idx], self.levels[l][2 * idx + 1])

    def path(self, idx):
        out = []
        for l in range(self.depth):
            out.append(self.levels[l][idx ^ 1])
            idx >>= 1
        return out


class RollupState:
    """The operator's side of a batch (the flow of operator/__tests__/operatorLogic.test.ts:100-222 and
    prover/__tests__/processtx.test.ts:24-130): accounts in a balance tree; `transfer` signs one transaction, applies
    it and returns its ProcessTx inputs; `batch_inputs` stacks them into the circuitInputs of BatchProcessTx."""

    def __init__(self, depth=6):
        self.tree = BalanceTree(depth)
        self.accounts = {}

    def deposit(self, index, pub, balance, nonce=0):
        self.accounts[index] = [int(pub[0]), int(pub[1]), int(balance), int(nonce)]
        self.tree.update(index, multi_hash(self.accounts[index]))       # hashBalanceTreeLeaf, helpers.ts:80-82

    def transfer(self, frm, to, amount, fee, priv):
        sa, ra, tree = self.accounts[frm], self.accounts[to], self.tree
        nonce = sa[3] + 1
        sig = sign(priv, [frm, to, amount, fee, nonce])                 # formatTx, helpers.ts:59-73
        inp = dict(balanceTreeRoot=tree.root, txData=[frm, to, amount, fee, nonce, sig["R8"][0], sig["R8"][1], sig["S"]],
                   txSenderPublicKey=sa[:2], txSenderBalance=sa[2], txSenderNonce=sa[3], txSenderPathElements=tree.path(frm),
                   txRecipientPublicKey=ra[:2], txRecipientBalance=ra[2], txRecipientNonce=ra[3], txRecipientPathElements=tree.path(to))
        sa[2] -= amount + fee
        sa[3] = nonce
        tree.update(frm, multi_hash(sa))
        inp["intermediateBalanceTreeRoot"] = tree.root
        inp["intermediateBalanceTreePathElements"] = tree.path(to)
        ra[2] += amount                                                 # the same record when frm == to (processtx.circom:141-159)
        tree.update(to, multi_hash(ra))
        return inp

    @staticmethod
    def batch_inputs(txs):
        return {f: [t[f] for t in txs] for f in TX_INPUT_FIELDS}
