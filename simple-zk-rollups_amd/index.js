"use strict";
// Host side of the MI355X Groth16 prover for Node.js (CommonJS; tsc is not needed to consume it,
// index.d.ts carries the types).  Same surface as the two things the reference uses:
//   websnark  : buildBn128() -> { groth16GenProof(witnessBin, provingKeyBin) }      (operator/src/snarks/common.ts:5,23,29)
//   snarkjs   : groth.genProof(provingKey, witness) -> { proof, publicSignals }      (north-star "witness in, {proof, publicSignals} out")
// plus binarifyWitness / binarifyProvingKey, the native-BigInt equivalents of operator/src/utils/binarify.ts.
// All arithmetic happens in csrc/libzkr_hip.so on the GPU; there is no JS or CPU fallback prover.
const path = require("path");

const Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583n;
const R = 21888242871839275222246405745257275088548364400416034343698204186575808495617n;
const MONT = 1n << 256n;

let addon = null;
let deviceCount = -1;
function native() {
  if (addon) return addon;
  const a = require(path.join(__dirname, "napi", "zkr_napi.node"));
  deviceCount = a.load(process.env.ZKR_HIP_LIB || path.join(__dirname, "csrc", "libzkr_hip.so"));
  addon = a;
  return addon;
}

function leBytesToDecimal(buf, off) {
  let v = 0n;
  for (let i = 31; i >= 0; i--) v = (v << 8n) | BigInt(buf[off + i]);
  return v.toString();
}

function proofFromBytes(pb) {
  const s = [];
  for (let i = 0; i < 8; i++) s.push(leBytesToDecimal(pb, 32 * i));
  return { pi_a: [s[0], s[1], "1"], pi_b: [[s[2], s[3]], [s[4], s[5]], ["1", "0"]], pi_c: [s[6], s[7], "1"] };
}

// 32-byte little-endian encoding of 0 <= v < 2^256; anything else is a caller error (big-integer's shift loop in
// binarify.ts:18-26 would silently keep the low 256 bits: a value x + k 2^256 must not pass for x)
function bigintToLe32(v) {
  const out = Buffer.alloc(32);
  let x = BigInt(v);
  if (x < 0n || x >= MONT) throw new RangeError("value does not fit 256 bits (or is negative): " + (x < 0n ? "-" : "") + "0x" + (x < 0n ? -x : x).toString(16).slice(0, 24) + "...");
  for (let i = 0; i < 32; i++) { out[i] = Number(x & 0xffn); x >>= 8n; }
  return out;
}
// public signals as the verifier takes them: out of [0, r) never verifies (TxVerifier.sol:265 `require(input[i] < r)`;
// snarkjs would reduce mod r, the chain refuses: the facade sides with the chain)
const signalsInRange = (ps) => ps.every((x) => { const v = BigInt(x); return v >= 0n && v < R; });

// Identity of a provingKeyBin -- EXACT, as the reference is by re-parsing the key on every call (operator/src/snarks/common.ts:28;
// ADVICE r4): candidates are found by the cheap SAMPLED digest (length + SHA-256 of the first 4 KiB -- geometry, alfa / beta / delta
// of the setup --, the last 4 KiB and 64 blocks of 4 KiB spread evenly in between: 0.3 MB whatever the size, < 0.1 ms) and confirmed
// by comparing the caller's buffer with the buffer the entry was built from, byte for byte (Buffer.compare = memcmp: ~8 ms for the tx
// circuit's 57 MB, against ~50 ms for a SHA-256 of it); the SAME buffer object is recognised without the comparison.  The entry
// keeps a private copy of the bytes it was built from (what other buffers are compared with) and the caller's object (identity
// only).  Contract of the identity shortcut: a buffer handed to the library is not rewritten in place while its key is cached --
// the reference's callers build a new buffer per call (common.ts:28) and always take the comparison.
// ZKR_KEY_FINGERPRINT=sampled: the sampled digest alone (two keys of one setup that differ in a lone coefficient alias);
// =full: a SHA-256 of every byte instead of the comparison (nothing retained).
const FP_BLOCK = 4096, FP_STRIDED = 64;
function bytesOf(buf) { return buf instanceof ArrayBuffer ? new Uint8Array(buf) : new Uint8Array(buf.buffer, buf.byteOffset, buf.byteLength); }
function keyFingerprint(buf, full) {
  const u8 = bytesOf(buf), n = u8.length, h = require("crypto").createHash("sha256");
  if (full === undefined) full = process.env.ZKR_KEY_FINGERPRINT === "full";
  if (full || n <= FP_BLOCK * (FP_STRIDED + 2)) h.update(u8);
  else {
    h.update(u8.subarray(0, FP_BLOCK));
    const span = n - 2 * FP_BLOCK;
    for (let i = 0; i < FP_STRIDED; i++) {
      const o = FP_BLOCK + Math.floor((span - FP_BLOCK) * i / (FP_STRIDED - 1));
      h.update(u8.subarray(o, o + FP_BLOCK));
    }
    h.update(u8.subarray(n - FP_BLOCK, n));
  }
  return n + ":" + h.digest("hex");
}

// Process-level cache of device keys (SURVEY.md 8(b) "Ownership").  The reference builds a NEW object for every proof
// (`await buildBn128()` at operator/src/snarks/common.ts:23 and scripts/index.js:40) and passes it the same
// provingKeyBin again (common.ts:28): a cache on the object would re-parse, re-upload and rebuild the window tables on
// every call.  An entry is one key CONTENT with its replicas, one per (device, ordinal) it has been asked for
// (groth16GenProofBatch with opts.devices); the least recently used of KEY_CACHE_SLOTS contents is dropped with all its
// replicas (their device memory goes with the handles).
const KEY_CACHE_SLOTS = 2;
const MAX_SHARD_SETS = 2;    // device lists whose shards an entry keeps (least recently used dropped)
const keyCache = new Map();  // "fingerprint#variant" -> Map("device#ordinal" -> native key handle; "ref" -> a private copy of the buffer the entry was built from, "src" -> that buffer object), in recency order
const keyCacheStats = { loads: 0, hits: 0, replications: 0, compares: 0 };
// a whole-key replica of the entry (any device), or undefined: the source of device-to-device copies and of shards
function anyReplica(ent) {
  for (const [slot, key] of ent) if (slot !== "ref" && slot !== "src" && !slot.startsWith("shards:")) return key;
  return undefined;
}
function cacheEntry(provingKeyBin) {
  const mode = process.env.ZKR_KEY_FINGERPRINT || "", fp = keyFingerprint(provingKeyBin);
  const u8 = bytesOf(provingKeyBin);
  const exact = mode !== "sampled" && mode !== "full" && u8.length > FP_BLOCK * (FP_STRIDED + 2);  // small buffers are hashed whole anyway
  let found, top = -1;
  for (const [k, ent] of keyCache) {
    if (!k.startsWith(fp + "#")) continue;
    top = Math.max(top, Number(k.slice(fp.length + 1)));
    if (!exact) { found = k; break; }
    const ref = ent.get("ref");
    if (ref === undefined) continue;  // an entry made under ZKR_KEY_FINGERPRINT=sampled / full keeps no bytes: another variant to an exact lookup
    if (ent.get("src") === provingKeyBin) { found = k; break; }  // the very object the entry was built from (see the contract above)
    keyCacheStats.compares++;
    if (Buffer.compare(ref, u8) === 0) { found = k; break; }
  }
  let ent;
  if (found !== undefined) { ent = keyCache.get(found); keyCache.delete(found); }  // re-insert: most recently used last
  else {
    found = fp + "#" + (top + 1);
    ent = new Map();
    // ADVICE r5: the bytes an entry is compared against are a PRIVATE copy -- the caller's buffer is mutable, and a buffer
    // rewritten in place after the load would make a later, different key compare equal to it; the caller's object itself is kept
    // for the identity shortcut only
    if (exact) { ent.set("ref", Buffer.from(u8)); ent.set("src", provingKeyBin); }
  }
  keyCache.set(found, ent);
  while (keyCache.size > KEY_CACHE_SLOTS) keyCache.delete(keyCache.keys().next().value);
  return ent;
}
function cachedKey(provingKeyBin, device) {
  const ent = cacheEntry(provingKeyBin), slot = device + "#0";
  let key = ent.get(slot);
  if (key !== undefined) keyCacheStats.hits++;
  else {
    // another device already holds this key: copy it device to device instead of parsing the buffer again
    const src = anyReplica(ent);
    if (src !== undefined) { key = native().keyReplicate(src, device, 0); keyCacheStats.replications++; }
    else { key = native().keyLoad(provingKeyBin, device); keyCacheStats.loads++; }
    ent.set(slot, key);
  }
  return key;
}
// one replica per entry of `devices` (a device listed twice gets two replicas: two proof pipelines on that GPU); the buffer is
// parsed at most once, every further replica is a device-to-device copy of the first (zkr_key_replicate)
function cachedReplicas(provingKeyBin, devices) {
  const ent = cacheEntry(provingKeyBin), seen = new Map(), keys = [];
  for (const d of devices) {
    const ord = seen.get(d) || 0;
    seen.set(d, ord + 1);
    const slot = d + "#" + ord;
    let key = ent.get(slot);
    if (key !== undefined) keyCacheStats.hits++;
    else {
      const src = anyReplica(ent);
      if (src !== undefined) { key = native().keyReplicate(src, d, 0); keyCacheStats.replications++; }
      else { key = native().keyLoad(provingKeyBin, d); keyCacheStats.loads++; }
      ent.set(slot, key);
    }
    keys.push(key);
  }
  return keys;
}
// the shards of a key over `devices` (one shard per entry, shard i = part i of devices.length on devices[i]): built once from the
// cached whole key (zkr_key_shard: window levels copied device to device) and kept with it in the cache entry
function cachedShards(provingKeyBin, devices) {
  const ent = cacheEntry(provingKeyBin), slot = "shards:" + devices.join(",");
  let shards = ent.get(slot);
  if (shards !== undefined) { keyCacheStats.hits++; ent.delete(slot); ent.set(slot, shards); return shards; }
  let whole = anyReplica(ent);
  if (whole === undefined) { whole = native().keyLoad(provingKeyBin, devices[0]); keyCacheStats.loads++; ent.set(devices[0] + "#0", whole); }
  shards = devices.map((d, i) => native().keyShard(whole, i, devices.length, d));
  keyCacheStats.shardings = (keyCacheStats.shardings || 0) + 1;
  const sets = Array.from(ent.keys()).filter((k) => k.startsWith("shards:"));
  while (sets.length >= MAX_SHARD_SETS) ent.delete(sets.shift());  // bounded: a caller cycling through device lists does not pile up shard sets
  ent.set(slot, shards);
  return shards;
}
function clearKeyCache() { keyCache.clear(); }

// blinding: null / undefined, or one {r, s} per proof (all or none)
async function proveBatchOn(keys, witnessBins, blinding) {
  const fixed = blinding && blinding.length === witnessBins.length && blinding.every((o) => o && o.r !== undefined && o.s !== undefined);
  if (blinding && blinding.some((o) => o && (o.r !== undefined || o.s !== undefined)) && !fixed) throw new Error("proveBatch: fix the blinding of every proof or of none");
  const rs = fixed ? Buffer.concat(blinding.map((o) => bigintToLe32(o.r))) : null;
  const ss = fixed ? Buffer.concat(blinding.map((o) => bigintToLe32(o.s))) : null;
  // keys.slice(): the addon references THIS array until the promise settles, so no replica is finalized under the job
  const pbs = keys.length === 1 ? await native().proveBatch(keys[0], witnessBins, rs, ss) : await native().proveBatchMulti(keys.slice(), witnessBins, rs, ss);
  const out = [];
  for (let i = 0; i < witnessBins.length; i++) out.push(proofFromBytes(pbs.subarray(256 * i, 256 * i + 256)));
  return out;
}

class Bn128 {
  constructor(device) { this.device = device || 0; this._fp = null; this._key = null; }

  // websnark signature.  The reference re-encodes and re-parses the key on every call (common.ts:28-29), on a fresh
  // object (common.ts:23); here the parsed, uploaded key is looked up in the process-level cache above, so only the
  // first proof of a key in a process pays for it.  opts.r / opts.s (BigInt|string) fix the blinding.
  // opts.devices: ONE proof over several GPUs (BASELINE configs[2] / [4] are single proofs; SURVEY 8(e) row 2) -- every MSM of the
  // proof is cut into devices.length contiguous ranges, shard i lives on devices[i] with 1/length of the key's tables, each shard
  // computes its partial sums (one host thread per shard inside zkr_prove_sharded), the host adds 640 bytes per shard and
  // assembles: the same proof as on one device.  The shards are built once per (key, devices) and cached with the key.
  async groth16GenProof(witnessBin, provingKeyBin, opts) {
    const a = native();
    const r = opts && opts.r !== undefined ? bigintToLe32(opts.r) : null;
    const s = opts && opts.s !== undefined ? bigintToLe32(opts.s) : null;
    if (opts && opts.devices && opts.devices.length > 1) {
      const devices = opts.devices.map(Number);
      if (devices.some((d) => !Number.isInteger(d) || d < 0 || d >= deviceCount)) throw new Error("groth16GenProof: opts.devices names a GPU this node does not have (" + deviceCount + " visible)");
      const shards = cachedShards(provingKeyBin, devices);
      return proofFromBytes(await a.proveSharded(shards.slice(), witnessBin, r, s));
    }
    this._key = cachedKey(provingKeyBin, opts && opts.devices && opts.devices.length === 1 ? Number(opts.devices[0]) : this.device);
    this._fp = "websnark";
    const pb = await a.prove(this._key, witnessBin, r, s);
    return proofFromBytes(pb);
  }

  // Independent proofs of one rollup batch on the same key: ONE native call (zkr_prove_batch on a libuv worker) that
  // pipelines uploads and proofs over the key's two workspaces and, for circuits far below the chip's size (the
  // reference's tx circuit, 2^17), runs several proofs in shared launches.
  // opts: an array -- opts[i] fixes the blinding {r, s} of proof i (all or none) -- or an object
  //   { devices: [0, 1, ...], blinding: [{r, s}, ...] }
  // devices: the GPUs of this node the batch is sharded over (BASELINE config 4: 64 proofs, 8 per GPU; SURVEY 8(e)): the key
  // is parsed once, copied device to device to the others (zkr_key_replicate, cached like the first) and proof i runs on
  // devices[i mod devices.length], one host thread per device inside zkr_prove_batch_multi; the result keeps the caller's order.
  async groth16GenProofBatch(witnessBins, provingKeyBin, opts) {
    if (witnessBins.length === 0) return [];
    native();
    const devices = opts && !Array.isArray(opts) && opts.devices ? opts.devices.map(Number) : null;
    const blinding = Array.isArray(opts) ? opts : opts && opts.blinding;
    if (devices && devices.length === 0) throw new Error("groth16GenProofBatch: opts.devices is empty");
    if (devices && devices.some((d) => !Number.isInteger(d) || d < 0 || d >= deviceCount)) throw new Error("groth16GenProofBatch: opts.devices names a GPU this node does not have (" + deviceCount + " visible)");
    if (!devices) {
      this._key = cachedKey(provingKeyBin, this.device);
      this._fp = "websnark";
      return this.proveBatch(witnessBins, blinding);
    }
    const keys = cachedReplicas(provingKeyBin, devices);
    this._key = keys[0];
    this._fp = "websnark";
    return proveBatchOn(keys, witnessBins, blinding);
  }
  // the same with the key currently held (after setup / loadKeyFile / a groth16GenProof call); opts.devices as above: the held
  // key is replicated to the devices it is not on yet (replicas are kept on this object)
  async proveBatch(witnessBins, opts) {
    if (!this._key) throw new Error("no key loaded");
    if (witnessBins.length === 0) return [];
    const devices = opts && !Array.isArray(opts) && opts.devices ? opts.devices.map(Number) : null;
    const blinding = Array.isArray(opts) ? opts : opts && opts.blinding;
    if (!devices) return proveBatchOn([this._key], witnessBins, blinding);
    if (!this._replicas || this._replicas.src !== this._key) this._replicas = { src: this._key, map: new Map([[native().keyDevice(this._key) + "#0", this._key]]) };
    const seen = new Map(), keys = [];
    for (const d of devices) {
      const ord = seen.get(d) || 0;
      seen.set(d, ord + 1);
      const slot = d + "#" + ord;
      if (!this._replicas.map.has(slot)) this._replicas.map.set(slot, native().keyReplicate(this._key, d, 0));
      keys.push(this._replicas.map.get(slot));
    }
    return proveBatchOn(keys, witnessBins, blinding);
  }

  // ---- a key held on the device without the websnark buffer in between (INTEGRATION.md section 5)
  // Groth16 setup of a compiled circuit on the GPU: what `snarkjs setup --protocol groth` does (prover/package.json:34,37).
  // circuitDef: circom's circuit JSON (common.ts:12-14).  opts.toxic: [t, alfa, beta, gamma, delta] for reproducible
  // tests; by default fresh toxic waste is drawn from the OS CSPRNG inside the library.  Returns the verifying key JSON.
  setup(circuitDef, opts) {
    const tox = opts && opts.toxic ? Buffer.concat(opts.toxic.map(bigintToLe32)) : null;
    const [key, vkBin] = native().setupR1cs(binarifyR1cs(circuitDef), tox, this.device);
    this._key = key; this._fp = "setup";
    return verifyingKeyFromBytes(vkBin);
  }
  // the same for a constraint system already in the r1cs_bin layout (RollupCircuit.r1cs())
  setupR1cs(r1csBin, opts) {
    const tox = opts && opts.toxic ? Buffer.concat(opts.toxic.map(bigintToLe32)) : null;
    const [key, vkBin] = native().setupR1cs(r1csBin, tox, this.device);
    this._key = key; this._fp = "setup";
    return verifyingKeyFromBytes(vkBin);
  }
  saveKey(path) { if (!this._key) throw new Error("no key loaded"); native().keySave(this._key, path); }
  loadKeyFile(path) { this._key = native().keyLoadFile(path, this.device); this._fp = "file:" + path; }
  // proof with the key currently held (after setup / loadKeyFile / a groth16GenProof call)
  async prove(witnessBin, opts) {
    if (!this._key) throw new Error("no key loaded");
    const r = opts && opts.r !== undefined ? bigintToLe32(opts.r) : null;
    const s = opts && opts.s !== undefined ? bigintToLe32(opts.s) : null;
    return proofFromBytes(await native().prove(this._key, witnessBin, r, s));
  }

  keyInfo() {
    if (!this._key) return null;
    const v = native().keyInfo(this._key);
    return { nVars: v[0], nPublic: v[1], domainSize: v[2], nnzA: v[3], nnzB: v[4] };
  }

  terminate() { this._key = null; this._fp = null; this._replicas = null; }
}

async function buildBn128(device) {
  native();
  return new Bn128(device);
}

// ---- binarify.ts equivalents on native BigInt (operator/src/utils/binarify.ts:10-48, 50-207)
function binarifyWitness(witness) {
  const out = Buffer.alloc(witness.length * 32);
  witness.forEach((w, i) => bigintToLe32(BigInt(w)).copy(out, 32 * i));
  return out.buffer.slice(out.byteOffset, out.byteOffset + out.byteLength);
}

function binarifyProvingKey(pk) {
  const n = pk.nVars, p = pk.nPublic, m = pk.domainSize;
  const polSize = (pol) => 36 * Object.keys(pol).length + 4;
  let size = 40 + 3 * 64 + 2 * 128;
  for (let i = 0; i < n; i++) size += polSize(pk.polsA[i]) + polSize(pk.polsB[i]);
  size += n * 64 * 2 + n * 128 + (n - p - 1) * 64 + m * 64;
  const buf = Buffer.alloc(size);
  let off = 0;
  const u32 = (v) => { buf.writeUInt32LE(Number(v), off); off += 4; };
  const fq = (v) => { bigintToLe32((BigInt(v) * MONT) % Q).copy(buf, off); off += 32; };
  const fr = (v) => { bigintToLe32((BigInt(v) * MONT) % R).copy(buf, off); off += 32; };
  const pt = (P) => { fq(P[0]); fq(P[1]); };
  const pt2 = (P) => { fq(P[0][0]); fq(P[0][1]); fq(P[1][0]); fq(P[1][1]); };
  const pol = (d) => { const keys = Object.keys(d); u32(keys.length); for (const k of keys) { u32(k); fr(d[k]); } };
  u32(n); u32(p); u32(m);
  const ptrAt = off; off += 28;
  pt(pk.vk_alfa_1); pt(pk.vk_beta_1); pt(pk.vk_delta_1); pt2(pk.vk_beta_2); pt2(pk.vk_delta_2);
  const ptrs = [];
  ptrs.push(off); for (let i = 0; i < n; i++) pol(pk.polsA[i]);
  ptrs.push(off); for (let i = 0; i < n; i++) pol(pk.polsB[i]);
  ptrs.push(off); for (let i = 0; i < n; i++) pt(pk.A[i]);
  ptrs.push(off); for (let i = 0; i < n; i++) pt(pk.B1[i]);
  ptrs.push(off); for (let i = 0; i < n; i++) pt2(pk.B2[i]);
  ptrs.push(off); for (let i = p + 1; i < n; i++) pt(pk.C[i]);
  ptrs.push(off); for (let i = 0; i < m; i++) pt(pk.hExps[i]);
  if (off !== size) throw new Error("binarifyProvingKey: size mismatch");
  ptrs.forEach((v, i) => buf.writeUInt32LE(v, ptrAt + 4 * i));
  return buf.buffer.slice(buf.byteOffset, buf.byteOffset + buf.byteLength);
}

// snarkjs-0.1.20 groth.genProof shape on top of the same core
async function genProof(provingKey, witness, opts) {
  const bn = await buildBn128(opts && opts.device);
  const proof = await bn.groth16GenProof(binarifyWitness(witness), binarifyProvingKey(provingKey), opts);
  proof.protocol = "groth";
  const publicSignals = witness.slice(1, provingKey.nPublic + 1).map((x) => BigInt(x).toString());
  return { proof, publicSignals };
}

// circom 0.0.35 circuit JSON (nVars, nPubInputs, nOutputs, constraints = [[A, B, C], ...] of {signal: coefficient})
// -> the r1cs_bin layout of zkr_setup_r1cs (include/zkr.h)
function binarifyR1cs(cd) {
  const p = cd.nPublic !== undefined ? cd.nPublic : cd.nPubInputs + cd.nOutputs;
  const parts = [];
  const u32 = (v) => { const b = Buffer.alloc(4); b.writeUInt32LE(Number(v), 0); parts.push(b); };
  u32(cd.nVars); u32(p); u32(cd.constraints.length);
  for (const row of cd.constraints)
    for (const lc of row) {
      const keys = Object.keys(lc);
      u32(keys.length);
      for (const k of keys) { u32(k); parts.push(bigintToLe32(((BigInt(lc[k]) % R) + R) % R)); }
    }
  return Buffer.concat(parts);
}

// vk_bin -> the constants of the generated verifier's `verifyingKey()` in the contract's own encoding
// (contracts/contracts/TxVerifier.sol:176-257): G1 as [x, y], G2 as [[x.im, x.re], [y.im, y.re]] (the EVM precompile order of
// TxVerifier.sol:18-22, the reverse of the snarkjs JSON) -- what `snarkjs generateverifier` (prover/package.json:35,38)
// pastes into TxVerifier.sol after a new setup (SURVEY 8(f-2)).
function solidityVerifyingKey(vk) {
  const rd = (o) => leBytesToDecimal(vk, o);
  const g1 = (o) => [rd(o), rd(o + 32)];
  const g2 = (o) => [[rd(o + 32), rd(o)], [rd(o + 96), rd(o + 64)]];
  const nic = vk.readUInt32LE(448);
  if (vk.length !== 452 + 64 * nic) throw new Error("vk_bin length does not match its IC count");
  const IC = [];
  for (let i = 0; i < nic; i++) IC.push(g1(452 + 64 * i));
  return { alfa1: g1(0), beta2: g2(64), gamma2: g2(192), delta2: g2(320), IC };
}
// the statements of `function verifyingKey()` for these constants
function solidityVerifyingKeySource(vk, indent) {
  const k = solidityVerifyingKey(vk), ind = indent === undefined ? "        " : indent;
  const g1 = (p) => `Pairing.G1Point(${p[0]},${p[1]})`;
  const g2 = (p) => `Pairing.G2Point([${p[0][0]},${p[0][1]}], [${p[1][0]},${p[1][1]}])`;
  const lines = [`vk.alfa1 = ${g1(k.alfa1)};`, `vk.beta2 = ${g2(k.beta2)};`, `vk.gamma2 = ${g2(k.gamma2)};`, `vk.delta2 = ${g2(k.delta2)};`,
    `vk.IC = new Pairing.G1Point[](${k.IC.length});`].concat(k.IC.map((p, i) => `vk.IC[${i}] = ${g1(p)};`));
  return lines.map((l) => ind + l + "\n").join("");
}

function verifyingKeyFromBytes(vk) {
  const rd = (o) => leBytesToDecimal(vk, o);
  const g1 = (o) => [rd(o), rd(o + 32), "1"];
  const g2 = (o) => [[rd(o), rd(o + 32)], [rd(o + 64), rd(o + 96)], ["1", "0"]];
  const nic = vk.readUInt32LE(448);
  const IC = [];
  for (let i = 0; i < nic; i++) IC.push(g1(452 + 64 * i));
  return { protocol: "groth", nPublic: nic - 1, vk_alfa_1: g1(0), vk_beta_2: g2(64), vk_gamma_2: g2(192), vk_delta_2: g2(320), IC };
}

// ---- groth.isValid(vk, proof, publicSignals) (operator/src/snarks/common.ts:30-34) on the native host verifier
function binarifyVerifyingKey(vk) {
  const g1 = (P) => Buffer.concat([bigintToLe32(P[0]), bigintToLe32(P[1])]);
  const g2 = (P) => Buffer.concat([bigintToLe32(P[0][0]), bigintToLe32(P[0][1]), bigintToLe32(P[1][0]), bigintToLe32(P[1][1])]);
  const n = Buffer.alloc(4);
  n.writeUInt32LE(vk.IC.length, 0);
  return Buffer.concat([g1(vk.vk_alfa_1), g2(vk.vk_beta_2), g2(vk.vk_gamma_2), g2(vk.vk_delta_2), n].concat(vk.IC.map(g1)));
}

function proofToBytes(proof) {
  const v = [proof.pi_a[0], proof.pi_a[1], proof.pi_b[0][0], proof.pi_b[0][1], proof.pi_b[1][0], proof.pi_b[1][1], proof.pi_c[0], proof.pi_c[1]];
  return Buffer.concat(v.map(bigintToLe32));
}

function isValid(vk, proof, publicSignals) {
  if (!signalsInRange(publicSignals)) return false;
  const pub = Buffer.concat(publicSignals.map((x) => bigintToLe32(BigInt(x))).concat([Buffer.alloc(0)]));
  let pb;
  try { pb = proofToBytes(proof); } catch (e) { if (e instanceof RangeError) return false; throw e; }  // a coordinate >= 2^256 is not a field element
  return native().verify(binarifyVerifyingKey(vk), pb, pub);
}

// every proof of a batch under one key, merged into one pairing product (zkr_verify_batch): true iff all verify
function isValidBatch(vk, proofs, publicSignalsList) {
  if (proofs.length === 0) return true;
  if (!publicSignalsList.every(signalsInRange)) return false;
  const pub = Buffer.concat(publicSignalsList.map((ps) => Buffer.concat(ps.map((x) => bigintToLe32(BigInt(x))).concat([Buffer.alloc(0)]))));
  let pbs;
  try { pbs = Buffer.concat(proofs.map(proofToBytes)); } catch (e) { if (e instanceof RangeError) return false; throw e; }
  return native().verifyBatch(binarifyVerifyingKey(vk), pbs, pub, proofs.length);
}

// operator/src/snarks/common.ts:43-50
function solidityProof(proof, publicSignals) {
  return {
    a: proof.pi_a.slice(0, 2),
    b: proof.pi_b.map((x) => x.slice().reverse()).slice(0, 2),
    c: proof.pi_c.slice(0, 2),
    inputs: publicSignals.map((x) => (BigInt(x) % R).toString()),
  };
}

// ---- the rollup circuit without circom / snarkjs (SURVEY 8(f-3)): operator/src/utils/crypto.ts natively, and
// BatchProcessTx(batch, depth) (prover/circuits/batchprocesstx.circom:3-75) as constraint system + witness builder
const concatLe = (values) => Buffer.concat(values.map((v) => bigintToLe32(BigInt(v))).concat([Buffer.alloc(0)]));
const leToBigInts = (buf) => { const out = []; for (let o = 0; o < buf.length; o += 32) out.push(BigInt(leBytesToDecimal(buf, o))); return out; };
// multiHash(d) (crypto.ts:28-30); hash / hashLeftRight (:32-38)
function multiHash(values) { return leToBigInts(native().rollupCrypto(0, concatLe(values.map((v) => BigInt(v) % R))))[0]; }
const hashLeftRight = (l, r) => multiHash([l, r]);
// the same on the GPU, one thread per hash: rows of equal length -> one hash per row
function multiHashBatch(rows, device) {
  if (rows.length === 0) return [];
  const flat = [];
  rows.forEach((r) => { if (r.length !== rows[0].length) throw new Error("multiHashBatch: rows of unequal length"); r.forEach((v) => flat.push(BigInt(v) % MONT)); });
  return leToBigInts(native().rollupGpuHash(concatLe(flat), rows[0].length, device || 0));
}
// the balance tree of operator/src/utils/merkletree.ts:44-83 over `leaves` (padded with zeroValue to 2^depth), hashed level
// by level on the GPU: { root, levels[l][i], path(i) = getUpdatePath(i).pathElements }
function buildBalanceTree(depth, leaves, zeroValue, device) {
  const n = 1 << depth;
  const full = leaves.map(BigInt).concat(new Array(n - leaves.length).fill(BigInt(zeroValue || 0)));
  const all = leToBigInts(native().rollupGpuHash(concatLe(full), 0, device || 0));
  const levels = [];
  for (let l = 0, off = 0; l <= depth; off += n >> l, l++) levels.push(all.slice(off, off + (n >> l)));
  return { depth, levels, root: levels[depth][0], path: (i) => levels.slice(0, depth).map((lv, l) => lv[(i >> l) ^ 1]) };
}
// genPublicKey(privKey) (crypto.ts:78-84)
function genPublicKey(priv) { return leToBigInts(native().rollupCrypto(1, bigintToLe32(priv))); }
// formatPrivKeyForBabyJub(privKey) (crypto.ts:58-76): the scalar the circuits take as `privateKey`
function formatPrivKeyForBabyJub(priv) { return leToBigInts(native().rollupCrypto(4, bigintToLe32(priv)))[0]; }
// sign(prv, msg) -> { R8: [x, y], S } (crypto.ts:143-168); verify(msg, sig, pubKey) (:170-177)
function sign(priv, msg) {
  const v = leToBigInts(native().rollupCrypto(2, bigintToLe32(priv), concatLe(msg)));
  return { R8: [v[0], v[1]], S: v[2] };
}
function verify(msg, sig, pub) {
  if ([sig.R8[0], sig.R8[1], sig.S, pub[0], pub[1]].some((v) => BigInt(v) < 0n || BigInt(v) >= MONT)) return false;
  return native().rollupCrypto(3, concatLe(msg), concatLe([sig.R8[0], sig.R8[1], sig.S]), concatLe(pub));
}

const TX_INPUT_FIELDS = ["balanceTreeRoot", "txData", "txSenderPublicKey", "txSenderBalance", "txSenderNonce", "txSenderPathElements",
  "txRecipientPublicKey", "txRecipientBalance", "txRecipientNonce", "txRecipientPathElements",
  "intermediateBalanceTreeRoot", "intermediateBalanceTreePathElements"];  // batchprocesstx.circom:13-36

// What `new Circuit(await compiler("tx.circom"))` is to the reference (common.ts:12-15), for BatchProcessTx(batch, depth):
// r1cs() feeds Bn128.setupCircuit; calculateWitness(circuitInputs) returns the witnessBin ArrayBuffer groth16GenProof /
// prove take (binarifyWitness layout) and throws where Circuit.calculateWitness would.
class RollupCircuit {
  constructor(batch, depth) {
    this.batch = batch === undefined ? 2 : Number(batch);   // tx.circom:3
    this.depth = depth === undefined ? 6 : Number(depth);
    const [nVars, nPublic, nConstraints, r1cs] = native().rollupCircuit(this.batch, this.depth, null);
    Object.assign(this, { nVars, nPublic, nConstraints });
    this._r1cs = r1cs;
  }
  r1cs() { return this._r1cs; }
  calculateWitness(circuitInputs) {
    const flat = [];
    const walk = (v) => { if (Array.isArray(v)) v.forEach(walk); else flat.push(((BigInt(v) % R) + R) % R); };
    if (Array.isArray(circuitInputs)) walk(circuitInputs); else TX_INPUT_FIELDS.forEach((f) => walk(circuitInputs[f]));
    if (flat.length !== this.nPublic - 1) throw new Error(`BatchProcessTx(${this.batch}, ${this.depth}) takes ${this.nPublic - 1} input values, got ${flat.length}`);
    const w = native().rollupCircuit(this.batch, this.depth, concatLe(flat));
    return w.buffer.slice(w.byteOffset, w.byteOffset + w.byteLength);
  }
  // witness.slice(1, nPubInputs + nOutputs + 1) (common.ts:18-21)
  publicSignals(witnessBin) { return leToBigInts(Buffer.from(witnessBin, 32, 32 * this.nPublic)); }
}

// Withdraw() (prover/circuits/withdraw.circom:4-25): public signals publicKey[0], publicKey[1], nullifier
class WithdrawCircuit {
  constructor() { this.nPublic = 3; }
  r1cs() { return native().withdrawCircuit(null); }
  // circuitInputs = { privateKey: formatPrivKeyForBabyJub(priv), nullifier }  (withdraw.test.ts:22-25)
  calculateWitness(circuitInputs) {
    const w = native().withdrawCircuit(bigintToLe32(BigInt(circuitInputs.privateKey)), bigintToLe32(BigInt(circuitInputs.nullifier) % R));
    return w.buffer.slice(w.byteOffset, w.byteOffset + w.byteLength);
  }
  publicSignals(witnessBin) { return leToBigInts(Buffer.from(witnessBin, 32, 96)); }
}

module.exports = {
  buildBn128, genProof, binarifyWitness, binarifyProvingKey, solidityProof, proofFromBytes, isValid, isValidBatch, binarifyVerifyingKey,
  binarifyR1cs, verifyingKeyFromBytes, solidityVerifyingKey, solidityVerifyingKeySource,
  // which form the last sharded proof took and why ({form: "split" | "replicated" | "none", reason}); how a key handle came to its device
  shardedLastForm: () => native().shardedLastForm(), keyReplication: (key) => native().keyReplication(key),
  keyCacheStats: () => Object.assign({ shardedLastForm: addon ? native().shardedLastForm() : { form: "none", reason: "" } }, { entries: keyCache.size, handles: Array.from(keyCache.values()).reduce((a, e) => a + Array.from(e.entries()).reduce((b, [k, v]) => b + (k === "ref" || k === "src" ? 0 : Array.isArray(v) ? v.length : 1), 0), 0) }, keyCacheStats), clearKeyCache, keyFingerprint,
  _cacheEntry: cacheEntry,  // the cache's lookup alone (no device involved): tests/test_node_host.py
  multiHash, multiHashBatch, buildBalanceTree, hashLeftRight, genPublicKey, formatPrivKeyForBabyJub, sign, verify, RollupCircuit, WithdrawCircuit,
  deviceCount: () => { native(); return deviceCount; },
  version: () => native().version(),
};
