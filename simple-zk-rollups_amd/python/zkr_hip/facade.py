"""Python mirror of the reference's proof facade for the hot path.

Reference: /root/reference/operator/src/snarks/common.ts:10-53 (createProofGenerator) and the
websnark surface it uses (buildBn128().groth16GenProof, common.ts:23,29).  Same names, argument
meaning and error behaviour; circuit compilation and witness calculation (common.ts:12-21) stay
with the caller, as the north-star leaves them unchanged.
"""
import collections
import hashlib
import threading

from .binding import ProvingKey, ZkrError

SNARK_FIELD_SIZE = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def proof_json_from_bytes(pb: bytes):
    """256-byte C-ABI proof -> the object groth16GenProof resolves to."""
    v = [str(int.from_bytes(pb[32 * i:32 * i + 32], "little")) for i in range(8)]
    return {"pi_a": [v[0], v[1], "1"], "pi_b": [[v[2], v[3]], [v[4], v[5]], ["1", "0"]], "pi_c": [v[6], v[7], "1"]}


def solidity_proof(proof, public_signals):
    """common.ts:43-50: drop the projective coordinate, swap (re, im) -> (im, re) in pi_b, inputs mod r."""
    return {"a": list(proof["pi_a"][:2]), "b": [list(reversed(x)) for x in proof["pi_b"]][:2],
            "c": list(proof["pi_c"][:2]), "inputs": [str(int(x) % SNARK_FIELD_SIZE) for x in public_signals]}


def binarify_verifying_key(vk) -> bytes:
    """snarkjs verifying-key JSON (vk_alfa_1, vk_beta_2, vk_gamma_2, vk_delta_2, IC; decimal strings or ints,
    projective third coordinate ignored) -> the vk_bin layout of zkr_verify (include/zkr.h)."""
    le = lambda v: int(v).to_bytes(32, "little")
    g1 = lambda p: le(p[0]) + le(p[1])
    g2 = lambda p: le(p[0][0]) + le(p[0][1]) + le(p[1][0]) + le(p[1][1])
    ic = vk["IC"]
    return g1(vk["vk_alfa_1"]) + g2(vk["vk_beta_2"]) + g2(vk["vk_gamma_2"]) + g2(vk["vk_delta_2"]) + len(ic).to_bytes(4, "little") + b"".join(g1(p) for p in ic)


def binarify_r1cs(circuit_def) -> bytes:
    """circom 0.0.35 circuit JSON (what `compiler(...)` returns at common.ts:12-14 and `circom -o build/tx.json`
    writes: nVars, nPubInputs, nOutputs, constraints = [[A, B, C], ...] with {signal: coefficient} maps) -> the
    r1cs_bin layout of zkr_setup_r1cs (include/zkr.h).  Also accepts nPublic directly and (signal, coef) pair lists."""
    p = circuit_def["nPublic"] if "nPublic" in circuit_def else circuit_def["nPubInputs"] + circuit_def["nOutputs"]
    cons = circuit_def["constraints"]
    out = [int(circuit_def["nVars"]).to_bytes(4, "little"), int(p).to_bytes(4, "little"), len(cons).to_bytes(4, "little")]
    for row in cons:
        for lc in row:
            items = list(lc.items()) if isinstance(lc, dict) else list(lc)
            out.append(len(items).to_bytes(4, "little"))
            for sig, cf in items:
                out.append(int(sig).to_bytes(4, "little") + (int(cf) % SNARK_FIELD_SIZE).to_bytes(32, "little"))
    return b"".join(out)


def verifying_key_from_bytes(vk_bin: bytes):
    """vk_bin (zkr_verify layout) -> the snarkjs verifying-key JSON fields that layout carries (decimal strings,
    projective third coordinates added), e.g. to feed `snarkjs generateverifier` (prover/package.json:35,38)."""
    rd = lambda o: str(int.from_bytes(vk_bin[o:o + 32], "little"))
    g1 = lambda o: [rd(o), rd(o + 32), "1"]
    g2 = lambda o: [[rd(o), rd(o + 32)], [rd(o + 64), rd(o + 96)], ["1", "0"]]
    nic = int.from_bytes(vk_bin[448:452], "little")
    return {"protocol": "groth", "nPublic": nic - 1, "vk_alfa_1": g1(0), "vk_beta_2": g2(64), "vk_gamma_2": g2(192), "vk_delta_2": g2(320),
            "IC": [g1(452 + 64 * i) for i in range(nic)]}


def solidity_verifying_key(vk_bin: bytes):
    """vk_bin -> the constants of the generated verifier contract's `verifyingKey()` in the contract's own encoding
    (contracts/contracts/TxVerifier.sol:176-257): G1 as [x, y]; G2 as [[x.im, x.re], [y.im, y.re]] -- the EVM precompile
    order of TxVerifier.sol:18-22, the reverse of the snarkjs JSON -- decimal strings.  The tail of SURVEY 8(f-2): what
    `snarkjs generateverifier` (prover/package.json:35,38) pastes into TxVerifier.sol for a new setup."""
    rd = lambda o: str(int.from_bytes(vk_bin[o:o + 32], "little"))
    g1 = lambda o: [rd(o), rd(o + 32)]
    g2 = lambda o: [[rd(o + 32), rd(o)], [rd(o + 96), rd(o + 64)]]
    nic = int.from_bytes(vk_bin[448:452], "little")
    if len(vk_bin) != 452 + 64 * nic:
        raise ValueError("vk_bin length does not match its IC count")
    return {"alfa1": g1(0), "beta2": g2(64), "gamma2": g2(192), "delta2": g2(320), "IC": [g1(452 + 64 * i) for i in range(nic)]}


def solidity_verifying_key_source(vk_bin: bytes, indent="        ") -> str:
    """The statements of `function verifyingKey()` for these constants (assignment per constant, IC array sized
    nPublic + 1), ready to replace TxVerifier.sol:177-255 / the same lines of WithdrawVerifier.sol after a new setup."""
    k = solidity_verifying_key(vk_bin)
    g1 = lambda p: "Pairing.G1Point(%s,%s)" % (p[0], p[1])
    g2 = lambda p: "Pairing.G2Point([%s,%s], [%s,%s])" % (p[0][0], p[0][1], p[1][0], p[1][1])
    lines = ["vk.alfa1 = %s;" % g1(k["alfa1"]), "vk.beta2 = %s;" % g2(k["beta2"]), "vk.gamma2 = %s;" % g2(k["gamma2"]),
             "vk.delta2 = %s;" % g2(k["delta2"]), "vk.IC = new Pairing.G1Point[](%d);" % len(k["IC"])]
    lines += ["vk.IC[%d] = %s;" % (i, g1(p)) for i, p in enumerate(k["IC"])]
    return "".join(indent + ln + "\n" for ln in lines)


def proof_bytes_from_json(proof) -> bytes:
    """Inverse of proof_json_from_bytes."""
    le = lambda v: int(v).to_bytes(32, "little")
    a, b, c = proof["pi_a"], proof["pi_b"], proof["pi_c"]
    return le(a[0]) + le(a[1]) + le(b[0][0]) + le(b[0][1]) + le(b[1][0]) + le(b[1][1]) + le(c[0]) + le(c[1])


def is_valid(verifying_key, proof, public_signals) -> bool:
    """groth.isValid(vk, proof, publicSignals) (common.ts:30-34) on the native host verifier (zkr_verify)."""
    from .binding import verify
    signals = [int(x) for x in public_signals]
    if any(v < 0 or v >= SNARK_FIELD_SIZE for v in signals):
        return False  # TxVerifier.sol:265 `require(input[i] < r)`: the chain refuses what snarkjs would reduce mod r
    try:
        pb = proof_bytes_from_json(proof)
    except OverflowError:
        return False  # a coordinate that does not fit 256 bits is not a field element
    return verify(binarify_verifying_key(verifying_key), pb, signals)


# Process-level cache of device keys (SURVEY.md 8(b) "Ownership").  The reference builds a NEW object for every proof
# (`await buildBn128()` at common.ts:23, scripts/index.js:40) and hands it the same provingKeyBin again (common.ts:28),
# so a cache on the object would re-parse, re-upload and rebuild the window tables on every call.  An entry is one key
# CONTENT with its replicas, one per (device, ordinal) asked for; the least recently used of KEY_CACHE_SLOTS contents is dropped
# with its replicas (a ProvingKey frees its arena when the last reference goes).
#
# Identity of a content -- EXACT, as the reference is by re-parsing the key on every call: candidates are found by the
# cheap SAMPLED digest (length + first and last 4 KiB + 64 blocks of 4 KiB spread evenly: 0.3 MB whatever the size, < 0.2 ms) and
# confirmed by comparing the caller's buffer with the buffer the entry was built from, byte for byte (a memcmp: ~8 ms for the tx
# circuit's 57 MB, against 50-100 ms for a cryptographic digest of it); the SAME buffer object is recognised without the
# comparison.  The entry keeps that buffer alive (a reference to immutable `bytes`, a private copy of anything mutable).
# ZKR_KEY_FINGERPRINT=sampled: the sampled digest alone (keys of one setup that differ in a lone coefficient alias);
# =full: a digest of every byte instead of the comparison (nothing retained).
KEY_CACHE_SLOTS = 2
FP_BLOCK, FP_STRIDED = 4096, 64
MAX_SHARD_SETS = 2                      # device lists whose shards an entry keeps (least recently used dropped)
_key_cache = collections.OrderedDict()   # (fingerprint, variant) -> {"ref": buffer or None, (device, ordinal): ProvingKey, ("shards", ...): [...]}
_key_cache_lock = threading.Lock()
key_cache_stats = {"loads": 0, "hits": 0, "replications": 0, "compares": 0}


def key_fingerprint(buf, full=None):
    """(length, digest) of a provingKeyBin.  full=True (or env ZKR_KEY_FINGERPRINT=full): digest of every byte; full=False: first and
    last 4 KiB + 64 blocks of 4 KiB spread evenly in between (0.3 MB whatever the size: sees any difference in geometry, setup or
    broad content, not a lone 36-byte coefficient); default: the sampled one unless the environment asks for `full` -- the key cache
    makes the sampled digest exact by comparing buffers (see above)."""
    import os
    n = len(buf)
    if full is None:
        full = os.environ.get("ZKR_KEY_FINGERPRINT") == "full"
    h = hashlib.blake2b(digest_size=16)
    if full or n <= FP_BLOCK * (FP_STRIDED + 2):
        h.update(buf if isinstance(buf, (bytes, bytearray, memoryview)) else bytes(buf))
    else:
        h.update(bytes(buf[:FP_BLOCK]))
        span = n - 2 * FP_BLOCK
        for i in range(FP_STRIDED):
            o = FP_BLOCK + (span - FP_BLOCK) * i // (FP_STRIDED - 1)
            h.update(bytes(buf[o:o + FP_BLOCK]))
        h.update(bytes(buf[n - FP_BLOCK:]))
    return (n, h.digest())


_fp_of_object = []   # [(bytes object, fingerprint)]: immutable buffers seen last, so that the SAME object costs no digest at all (two slots)


def _fingerprint_memoised(buf):
    if type(buf) is not bytes:           # anything mutable is digested on every call
        return key_fingerprint(buf)
    for obj, fp in _fp_of_object:
        if obj is buf:
            return fp
    fp = key_fingerprint(buf)
    _fp_of_object.insert(0, (buf, fp))
    del _fp_of_object[2:]
    return fp


def _entry(proving_key_bin):   # caller holds the lock
    import os
    mode = os.environ.get("ZKR_KEY_FINGERPRINT", "")
    fp = _fingerprint_memoised(proving_key_bin)
    exact = mode not in ("sampled", "full") and len(proving_key_bin) > FP_BLOCK * (FP_STRIDED + 2)   # small buffers are hashed whole anyway
    found = None
    variants = [k for k in _key_cache if k[0] == fp]
    for k in variants:
        ref = _key_cache[k]["ref"]
        if not exact or ref is proving_key_bin:
            found = k
            break
        key_cache_stats["compares"] += 1
        if ref == proving_key_bin:                          # bytes / bytearray / memoryview compare by content (memcmp)
            found = k
            break
    if found is None:
        found = (fp, 1 + max([k[1] for k in variants], default=-1))
        ref = None
        if exact:
            ref = proving_key_bin if type(proving_key_bin) is bytes else bytes(proving_key_bin)
        _key_cache[found] = {"ref": ref}
    _key_cache.move_to_end(found)
    while len(_key_cache) > KEY_CACHE_SLOTS:
        _key_cache.popitem(last=False)
    return _key_cache[found]


def _replica(ent, proving_key_bin, device, ordinal):
    key = ent.get((device, ordinal))
    if key is not None:
        key_cache_stats["hits"] += 1
        return key
    src = next((k for s, k in ent.items() if s != "ref" and s[0] != "shards"), None)
    if src is not None:   # another replica holds this content: device-to-device copy instead of a second parse
        key = src.replicate(device)
        key_cache_stats["replications"] += 1
    else:
        key = ProvingKey.load_websnark(proving_key_bin, device)
        key_cache_stats["loads"] += 1
    ent[(device, ordinal)] = key
    return key


def cached_key(proving_key_bin, device=0):
    """The device key for this provingKeyBin: parsed and uploaded on first sight, reused afterwards by every Bn128."""
    with _key_cache_lock:
        return _replica(_entry(proving_key_bin), proving_key_bin, device, 0)


def cached_replicas(proving_key_bin, devices):
    """One replica per entry of `devices` (a device listed twice gets two): the buffer is parsed at most once, every further
    replica is a device-to-device copy (zkr_key_replicate)."""
    with _key_cache_lock:
        ent, seen, keys = _entry(proving_key_bin), {}, []
        for d in devices:
            ordinal = seen.get(d, 0)
            seen[d] = ordinal + 1
            keys.append(_replica(ent, proving_key_bin, d, ordinal))
        return keys


def cached_shards(proving_key_bin, devices):
    """The shards of this key over `devices` (shard i = part i of len(devices) on devices[i]; zkr_key_shard), built once from
    the cached whole key and kept with it."""
    with _key_cache_lock:
        ent = _entry(proving_key_bin)
        slot = ("shards",) + tuple(devices)
        shards = ent.get(slot)
        if shards is not None:
            key_cache_stats["hits"] += 1
            ent[slot] = ent.pop(slot)                      # most recently used last
            return shards
        whole = next((k for s, k in ent.items() if s != "ref" and s[0] != "shards"), None)
        if whole is None:
            whole = _replica(ent, proving_key_bin, devices[0], 0)
        shards = [whole.shard(i, len(devices), d) for i, d in enumerate(devices)]
        sets = [k for k in ent if k != "ref" and k[0] == "shards"]
        while len(sets) >= MAX_SHARD_SETS:                 # bounded: a caller cycling through device lists does not pile up shard sets
            del ent[sets.pop(0)]
        ent[slot] = shards
        return shards


def clear_key_cache():
    with _key_cache_lock:
        _key_cache.clear()
        del _fp_of_object[:]


class Bn128:
    """What `await buildBn128()` returns; only groth16GenProof is used by the reference."""

    def __init__(self, device=0):
        self.device = device

    def groth16GenProof(self, witness_bin: bytes, proving_key_bin: bytes, r=None, s=None, devices=None):
        # the reference re-encodes and re-parses the key on every call (common.ts:28-29) on a fresh object (:23)
        if devices is not None and len(devices) > 1:   # ONE proof over several GPUs: cached shards, zkr_prove_sharded
            from .binding import prove_sharded
            return proof_json_from_bytes(prove_sharded(cached_shards(proving_key_bin, list(devices)), witness_bin, r, s))
        return proof_json_from_bytes(cached_key(proving_key_bin, self.device).prove(witness_bin, r, s))

    def groth16GenProofBatch(self, witness_bins, proving_key_bin, rs=None, ss=None, devices=None):
        """Independent proofs of one rollup batch on one key in one native call (index.js groth16GenProofBatch).  devices:
        the GPUs of this node the batch is sharded over (proof i on devices[i mod len]; SURVEY 8(e), BASELINE config 4) --
        zkr_prove_batch_multi over cached replicas; None: this object's device."""
        from .binding import prove_batch_multi
        if not witness_bins:
            return []
        if devices is None:
            proofs = cached_key(proving_key_bin, self.device).prove_batch(witness_bins, rs, ss)
        else:
            proofs = prove_batch_multi(cached_replicas(proving_key_bin, list(devices)), witness_bins, rs, ss)
        return [proof_json_from_bytes(p) for p in proofs]


def build_bn128(device=0):
    return Bn128(device)


def groth16_gen_proof(witness_bin, proving_key_bin, device=0, r=None, s=None):
    return build_bn128(device).groth16GenProof(witness_bin, proving_key_bin, r, s)


def create_proof_generator(proving_key_bin, verifying_key, n_public, device=0, is_valid=is_valid):
    """common.ts:10-53 with the circuit/witness steps supplied by the caller: returns
    fn(witness: list[int]) -> {proof, solidityProof}; raises Error("Invalid proof generated")
    (common.ts:36-38) when `is_valid(vk, proof, publicSignals)` rejects the proof (default: the native
    verifier; any callable with snarkjs' groth.isValid signature can be passed)."""
    bn = build_bn128(device)

    def gen(witness, r=None, s=None):
        public_signals = witness[1:n_public + 1]  # common.ts:18-21
        witness_bin = b"".join(int(x).to_bytes(32, "little") for x in witness)  # binarifyWitness, binarify.ts:10-48
        proof = bn.groth16GenProof(witness_bin, proving_key_bin, r, s)
        if not is_valid(verifying_key, proof, public_signals):
            raise ZkrError(-100, "Invalid proof generated")
        return {"proof": proof, "solidityProof": solidity_proof(proof, public_signals)}

    return gen
