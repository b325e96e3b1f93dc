/* zkr.h -- C ABI of libzkr_hip.so, the MI355X (gfx950) Groth16 prover behind
 * `wasmBn128.groth16GenProof(witnessBin, provingKeyBin)`.
 *
 * Drop-in boundary (reference call sites, /root/reference = kendricktan/simple-zk-rollups):
 *   operator/src/snarks/common.ts:23   const wasmBn128 = await buildBn128();
 *   operator/src/snarks/common.ts:27   witnessBin   = binarifyWitness(witness)      (binarify.ts:10-48)
 *   operator/src/snarks/common.ts:28   provingKeyBin= binarifyProvingKey(provingKey) (binarify.ts:50-207)
 *   operator/src/snarks/common.ts:29   proof = await wasmBn128.groth16GenProof(witnessBin, provingKeyBin)
 *   scripts/index.js:40,46             second copy of the same call
 * The N-API shim (simple-zk-rollups_amd/napi/zkr_napi.c) binds exactly these entry points; see
 * INTEGRATION.md for the one-line change in common.ts.
 *
 * Conventions: plain pointers and sizes, no C++/torch types.  Every function returns 0 on success
 * and a negative zkr_status on failure; zkr_last_error() gives a thread-local message.  All field
 * elements cross the boundary as 32-byte little-endian integers.  "std" = standard form,
 * "mont" = Montgomery form (x * 2^256 mod p), exactly as binarify.ts:78-90 writes key material.
 * There is NO CPU fallback: without a usable HIP device every compute entry point fails with
 * ZKR_ERR_NO_DEVICE.
 */
#ifndef ZKR_H
#define ZKR_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  ZKR_OK = 0,
  ZKR_ERR_NO_DEVICE = -1,   /* no HIP device / HIP runtime error at init */
  ZKR_ERR_BAD_KEY = -2,     /* provingKeyBin header/size/section mismatch (binarify.ts:115-141) */
  ZKR_ERR_BAD_WITNESS = -3, /* witness length != nVars * 32 */
  ZKR_ERR_HIP = -4,         /* HIP runtime failure during a call */
  ZKR_ERR_ARG = -5,         /* null pointer / out-of-range argument */
  ZKR_ERR_DEGENERATE = -6,  /* a proof element is the point at infinity (cannot be serialised) */
  ZKR_ERR_UNSATISFIED = -7  /* circuit inputs violate a constraint (where Circuit.calculateWitness throws) */
} zkr_status;

typedef struct zkr_key zkr_key; /* device-resident proving key (one contiguous HBM arena + workspace) */

/* ---- lifecycle ------------------------------------------------------------------------------ */
const char *zkr_last_error(void);
const char *zkr_version(void);
/* Number of HIP devices visible (0 when none; never fails). */
int zkr_device_count(void);
/* PCI address of HIP device `device` ("0000:c1:00.0"; out_len >= 16): lets a monitoring harness find the device's sysfs
 * node (/sys/bus/pci/devices/<id>: hwmon clock and power) among the GPUs of the node. */
int zkr_device_pci_bus_id(int device, char *out, size_t out_len);

/* Parse a websnark-format proving key (the ArrayBuffer of binarifyProvingKey, binarify.ts:143-206),
 * build the device layout (CSR QAP rows, compacted Montgomery point tables, twiddles) and upload it
 * to `device`.  Replaces the per-call parse inside groth16GenProof (common.ts:28-29). */
int zkr_key_load_websnark(const void *pk_bin, size_t pk_len, int device, zkr_key **out);
void zkr_key_free(zkr_key *key);

/* Key geometry: out[0]=nVars out[1]=nPublic out[2]=domainSize out[3]=nnzA out[4]=nnzB
 * out[5..9]=points kept (non-infinity) in the A,B1,B2,C,H tables. */
int zkr_key_info(const zkr_key *key, uint64_t out[10]);

/* Packed key on disk: the device arena -- CSR rows, window tables, twiddles -- written as is, so a
 * later process loads it with one read + one upload: no JSON, no binarifyProvingKey (binarify.ts:50-207, per call in
 * the reference), no re-parse and no window-table rebuild.  The file is position independent; its header carries
 * a magic and the total length. */
int zkr_key_save(const zkr_key *key, const char *path);
int zkr_key_load_file(const char *path, int device, zkr_key **out);

/* MSM geometry of the device key: window bits c and window count K = ceil(255/c) of the A,B1,B2,C,H tables.  Each
 * table holds K levels per base point (2^(ck) P), so an MSM costs K mixed additions per point. */
int zkr_key_windows(const zkr_key *key, uint32_t c_out[5], uint32_t k_out[5]);

/* Multi-GPU replication: the key is ONE position-independent arena in HBM.
 * rank 0:  zkr_key_arena(key,&ptr,&len)  -> broadcast `len`, then the bytes (RCCL over xGMI)
 * rank k:  zkr_key_adopt_arena(dev_ptr,len,device,&key)  (takes a device pointer holding the bytes;
 *          the caller keeps ownership of that memory and must keep it alive until zkr_key_free). */
int zkr_key_arena(const zkr_key *key, void **dev_ptr, size_t *len);
int zkr_key_adopt_arena(void *dev_ptr, size_t len, int device, zkr_key **out);
/* The same replication with 1/10 of the bytes on the wire: zkr_key_base_arena gives a compact device buffer (owned by
 * the key, valid until zkr_key_free) holding the QAP rows, the BASE points of every table and the rank maps, without
 * the precomputed window levels and twiddles; zkr_key_adopt_base_arena copies it into a full arena on `device` and
 * rebuilds those levels there (about 0.25 s at 2^20).  The caller's buffer may be released as soon as the call returns;
 * the rebuilt arena is byte-identical to the sender's. */
int zkr_key_base_arena(zkr_key *key, void **dev_ptr, size_t *len);
int zkr_key_adopt_base_arena(const void *dev_ptr, size_t len, int device, zkr_key **out);

/* The same replication inside ONE process, device to device (SURVEY.md 8(b) "Threading": one host thread per GPU;
 * 8(e)): the form a Node operator uses -- the reference's host awaits its proofs from one process
 * (operator/src/snarks/common.ts:23-29) -- where no launcher starts a rank per GPU.  Copies `key` (on its own device)
 * to `dst_device` and returns an independent key there (own workspaces; free both with zkr_key_free, in any order).
 * mode FULL: the whole arena in peer copies of at most 1 GiB, adopted as is (nothing recomputed; over xGMI about
 * 30 ms at 2^20); BASE: the compact arena (1/10 of the bytes) and a rebuild of the window levels on dst_device
 * (about 0.1 s at 2^20) -- for device pairs without direct peer access, where the runtime stages the copy through host
 * memory; AUTO: FULL when the devices can address each other (or are the same device), else BASE.  dst_device may be
 * key's own device: a second replica there (two host threads, two proof pipelines on one GPU).  The replica's arena is
 * byte-identical to the source's in every mode. */
enum { ZKR_REPLICATE_AUTO = 0, ZKR_REPLICATE_FULL = 1, ZKR_REPLICATE_BASE = 2 };
int zkr_key_replicate(const zkr_key *key, int dst_device, int mode, zkr_key **out);
/* HIP ordinal of the device the key lives on. */
int zkr_key_device(const zkr_key *key);
/* How the key came to its device: *mode_out = 0 (loaded, built or adopted there) or the form zkr_key_replicate chose
 * (ZKR_REPLICATE_FULL / ZKR_REPLICATE_BASE); *peer_direct_out (may be NULL) = whether source and destination could address each
 * other's memory (0 with FULL: the runtime staged the copy through the host). */
int zkr_key_replication(const zkr_key *key, int *mode_out, int *peer_direct_out);

/* ---- the hot path --------------------------------------------------------------------------- */
/* One Groth16 proof.  witness_std: nVars x 32 B standard form (binarifyWitness layout, host memory).
 * r32/s32: blinding scalars (32 B LE, < r); pass NULL for both to draw them from the OS CSPRNG as
 * the reference does.  proof_out: 256 B = pi_a (x,y) | pi_b (x.re,x.im,y.re,y.im) | pi_c (x,y),
 * affine, standard form -- the eight integers groth16GenProof returns as decimal strings
 * (consumed at common.ts:31-32,44-48).
 * stream: the hipStream_t that produced the witness (e.g. torch.cuda.current_stream().cuda_stream; NULL = the
 * default stream): the proof starts after the work already enqueued there and then runs on the key's own streams. */
int zkr_prove(zkr_key *key, const void *witness_std, size_t witness_len, const uint8_t *r32, const uint8_t *s32,
              uint8_t proof_out[256], void *stream);

/* Same, witness already resident in HBM on the key's device (nVars x 32 B, std form). */
int zkr_prove_device(zkr_key *key, const void *d_witness_std, const uint8_t *r32, const uint8_t *s32,
                     uint8_t proof_out[256], void *stream);

/* Pipelined form for batches of independent proofs (rollup batch, BASELINE config 4): zkr_prove_submit enqueues
 * the whole GPU side of one proof and returns at once with a ticket; zkr_prove_collect waits for that proof and
 * does the host assembly.  A key holds two proof workspaces (zkr_key_slots), so the GPU work of proof i+1 (submitted before
 * collecting proof i) covers proof i's reduction tail and host assembly.  The witness buffer must stay untouched
 * until the ticket is collected.  Submitting with both workspaces in flight fails with ZKR_ERR_ARG.
 * zkr_prove_device(...) == submit + collect. */
int zkr_prove_submit(zkr_key *key, const void *d_witness_std, const uint8_t *r32, const uint8_t *s32, void *stream, int *ticket);
int zkr_prove_collect(zkr_key *key, int ticket, uint8_t proof_out[256]);
/* A whole batch from host buffers (the rollup operator's case: `count` independent witnesses against one key, e.g. the
 * proofs of BASELINE config 4 assigned to this GPU): uploads and proofs pipelined over the key's workspaces.
 * witnesses_std: `count` host pointers to witness_len bytes each; r32s / s32s: count x 32 B blinding scalars or NULL
 * (drawn per proof); proofs_out: count x 256 B.  Stops at the first failing proof (its status is returned). */
int zkr_prove_batch(zkr_key *key, const void *const *witnesses_std, size_t witness_len, size_t count, const uint8_t *r32s, const uint8_t *s32s,
                    uint8_t *proofs_out);
/* The same for witnesses already resident in HBM (count device pointers, nVars x 32 B each; `stream` as for
 * zkr_prove_device).  Both batch calls FUSE proofs of small circuits: a key whose circuit is far below the size that fills
 * the chip (the reference's tx circuit, 2^17 constraints) runs every kernel over up to zkr_key_fuse(key) witnesses at once
 * (vectors end to end, one bucket set per proof and table), so a launch carries about the work of one 2^20 proof and the
 * fixed latency tail of the bucket reduction is paid once per group; two groups are in flight.  Proofs are the same bytes as
 * from zkr_prove_device. */
int zkr_prove_batch_device(zkr_key *key, const void *const *d_witnesses_std, size_t count, const uint8_t *r32s, const uint8_t *s32s, void *stream,
                           uint8_t *proofs_out);
/* A batch over SEVERAL GPUs of the node from one process (BASELINE config 4: 64 independent proofs, 8 per GPU): keys[j] are
 * replicas of one key on n_keys devices (zkr_key_replicate; the same device may appear through two replicas), proof i
 * goes to keys[i mod n_keys], one host thread per key runs zkr_prove_batch / zkr_prove_batch_device on its share (uploads
 * and proofs pipelined per device, small circuits fused), proofs_out keeps the caller's order.  No data moves between
 * the devices.  r32s / s32s as in zkr_prove_batch.  _device: d_witnesses_std[i] is resident on the device of
 * keys[i mod n_keys] and complete when the call is made.  The first failing share's status is returned with its
 * message (prefixed by the key's index and device); the other shares still run to their end. */
int zkr_prove_batch_multi(zkr_key *const *keys, size_t n_keys, const void *const *witnesses_std, size_t witness_len, size_t count, const uint8_t *r32s,
                          const uint8_t *s32s, uint8_t *proofs_out);
int zkr_prove_batch_multi_device(zkr_key *const *keys, size_t n_keys, const void *const *d_witnesses_std, size_t count, const uint8_t *r32s,
                                 const uint8_t *s32s, uint8_t *proofs_out);
/* ---- ONE proof over several GPUs (BASELINE configs[2] and [4] are single proofs) ----------------------
 * The MSMs shard by contiguous ranges: zkr_key_shard cuts every scalar vector of the proof -- the witness w (A, B1, B2, C
 * tables) and the quotient coefficients h (H table) -- into `parts` equal ranges and builds, on `device`, a key holding
 * range `part` of every point table (all window levels; the shard's window size is the one a key of ITS size would choose, so
 * the levels are rebuilt on `device` from the range's base points unless the window stays the same, in which case they are
 * copied device to device) plus the whole QAP and the twiddles: every shard computes h itself ("replicated compute"), so NOTHING is exchanged but the
 * partial sums at the end -- 640 bytes per shard.  The key is sharded, not replicated: a shard holds about 1/parts of the
 * table memory (78 GB at 2^24).
 * zkr_prove_partial(_device) runs a shard's whole share of one proof (host / HBM-resident witness, the FULL witness either
 * way) and returns its partial sums of A, B1, B2 and C + H (ZKR_PARTIAL_BYTES, this library's XYZZ points in Montgomery
 * form: opaque to the caller); zkr_prove_combine adds the `parts` records and does the usual assembly (r32 / s32 as in
 * zkr_prove; `key` = any of the shards, or the whole key: they carry the same alfa, beta, delta).  The proof is the same
 * bytes as zkr_prove's on the whole key.  zkr_prove_sharded(_device): the three steps in one call, one host thread per
 * shard (shards[i] = part i of `parts`; _device: d_witnesses_std[i] = the full witness resident on shard i's device). */
#define ZKR_PARTIAL_BYTES 640
int zkr_key_shard(const zkr_key *key, unsigned part, unsigned parts, int device, zkr_key **out);
int zkr_prove_partial(zkr_key *shard, const void *witness_std, size_t witness_len, uint8_t partial_out[ZKR_PARTIAL_BYTES]);
int zkr_prove_partial_device(zkr_key *shard, const void *d_witness_std, void *stream, uint8_t partial_out[ZKR_PARTIAL_BYTES]);
int zkr_prove_combine(zkr_key *key, const uint8_t *partials, size_t parts, const uint8_t *r32, const uint8_t *s32, uint8_t proof_out[256]);
int zkr_prove_sharded(zkr_key *const *shards, size_t parts, const void *witness_std, size_t witness_len, const uint8_t *r32, const uint8_t *s32,
                      uint8_t proof_out[256]);
int zkr_prove_sharded_device(zkr_key *const *shards, size_t parts, const void *const *d_witnesses_std, const uint8_t *r32, const uint8_t *s32,
                             uint8_t proof_out[256]);
/* zkr_prove_sharded(_device) run the shards concurrently, so calcH need not be repeated by every shard: with 2, 4 or 8 shards
 * (each owning one aligned block of h) on devices that can access one another's memory, shard j evaluates 1/parts of the QAP rows
 * and 1/parts of every transform's butterflies; the stages that pair elements of different blocks read and write the other
 * shards' buffers directly (xGMI peer access; shards on one device: plain loads), with a host barrier of the shards' threads
 * between the five phases (four barriers).  The proof is the same bytes.  ZKR_SHARD_SPLIT_H=0: every shard computes h for itself (what
 * zkr_prove_partial on its own always does).  zkr_prove_sharded_split_stats: of the calling thread's last sharded proof --
 * *parts_out = its shards if calcH was split (else 0), phase_ms_out[8 * part + phase] = host time of that shard's phase
 * (enqueue until its stream was idle; phases 0..4 = 1..5 of csrc/zkr_prove.hip calc_h_split, the last one enqueue time only). */
int zkr_prove_sharded_split_stats(unsigned *parts_out, double phase_ms_out[64]);
/* WHICH form the calling thread's last zkr_prove_sharded(_device) took, and why -- a sharded proof that fell back to replicated
 * calcH is otherwise only a slower number: *form_out = ZKR_SHARDED_SPLIT_H / ZKR_SHARDED_REPLICATED_H (NONE: no sharded proof on
 * this thread yet, or it failed before running), reason_out = one line ("no peer access from device 2 to device 5", "3 shards:
 * the split needs 2, 4 or 8", "ZKR_SHARD_SPLIT_H=0", "all shards on one device", ...).
 * Shards on DIFFERENT devices: their first sharded proof runs BOTH forms and compares the sums; the split is kept only if they
 * agree (a warning on stderr and replicated calcH from then on otherwise).  ZKR_SHARD_SPLIT_H=1 skips that check. */
enum { ZKR_SHARDED_NONE = 0, ZKR_SHARDED_SPLIT_H = 1, ZKR_SHARDED_REPLICATED_H = 2 };
int zkr_prove_sharded_last_form(int *form_out, char *reason_out, size_t reason_len);
/* Measurement only (bench.py's shard leg on a one-GPU box): ONE shard runs its share of a proof with a split calcH ALONE, its own
 * buffers standing in for the other shards' -- the time a shard takes with a GPU to itself and no exchange (*ms_out); what it
 * computes is meaningless and is discarded. */
int zkr_bench_shard_split_solo(zkr_key *shard, const void *d_witness_std, double *ms_out);
/* out[0] = part, out[1] = parts (0, 1 for a whole key), out[2..3] = first scalar and count of the witness range,
 * out[4..5] = the same for h. */
int zkr_key_shard_info(const zkr_key *key, uint32_t out[6]);
/* Number of proof workspaces of the key = submits that can be in flight; proofs one batch submit fuses (1 at 2^20 and above). */
int zkr_key_slots(const zkr_key *key);
int zkr_key_fuse(const zkr_key *key);

/* ---- acceptance check (host only, no GPU) ---------------------------------------------------------
 * The pairing equation `groth.isValid(vk, proof, publicSignals)` evaluates at common.ts:30-38 and
 * TxVerifier.verify evaluates on chain (TxVerifier.sol:258-276):
 *   vk_x = IC_0 + sum_i input_i IC_{i+1};  e(-A, B) e(alfa, beta) e(vk_x, gamma) e(C, delta) == 1.
 * vk_bin: 64 B vk_alfa_1 | 128 B vk_beta_2 | 128 B vk_gamma_2 | 128 B vk_delta_2 | u32 nIC | nIC x 64 B IC, every
 * coordinate a 32-byte LE standard-form integer, G2 as (x.re, x.im, y.re, y.im) (the order of the snarkjs JSON
 * key; index.js / facade.py convert the JSON).  proof: the 256 bytes zkr_prove returns.  public_std: nIC - 1
 * inputs, 32 B each.  *valid = 1 / 0; inputs >= r (TxVerifier.sol:265) and off-curve proof points give 0.
 * Returns an error only for a malformed key or a wrong input count. */
int zkr_verify(const void *vk_bin, size_t vk_len, const uint8_t proof[256], const void *public_std, size_t n_public, int *valid);
/* n_proofs proofs (n_proofs x 256 B) with their public signals (n_proofs x n_public x 32 B) under one key, merged by a
 * random linear combination (128-bit coefficients from the OS CSPRNG) into ONE product of n_proofs + 3 pairings with one
 * final exponentiation: *all_valid = 1 iff every proof verifies (an invalid one slips through with
 * probability 2^-128); about 0.5 ms per proof instead of 3.7.  Use zkr_verify to locate a failing proof. */
int zkr_verify_batch(const void *vk_bin, size_t vk_len, const uint8_t *proofs, const void *publics_std, size_t n_proofs, size_t n_public, int *all_valid);

/* ---- stage hooks (tests, profiling) ----------------------------------------------------------- */
/* In-place NTT of n = 2^logn standard-form elements in host memory; natural order in and out (words >= r are reduced first). */
int zkr_ntt(void *data_std, unsigned logn, int inverse, int device);
/* sum_i scalars[i] * points[i].  points: Montgomery affine as in the key sections (64 B G1 / 128 B G2;
 * x == 0 encodes infinity, binarify.ts:92-102); scalars: std 32 B.  out: std affine; *is_inf set when
 * the sum is the point at infinity. */
int zkr_msm_g1(const void *points_mont, const void *scalars_std, size_t n, uint8_t out[64], int *is_inf, int device);
int zkr_msm_g2(const void *points_mont, const void *scalars_std, size_t n, uint8_t out[128], int *is_inf, int device);
/* h = upper-half coefficients of A(x)B(x), natural order, std form, m x 32 B. */
int zkr_calc_h(zkr_key *key, const void *witness_std, size_t witness_len, void *h_out);

/* Per-stage device timing, measured with hipEvents on the launch stream.  Stage names:
 * "ingest","spmv","ntt","msm_sort","msm_accum_g1","msm_accum_g2","msm_big","msm_reduce","total".
 * "msm_accum_g1"/"msm_accum_g2" bracket exactly one msm_accum_kernel launch each time. */
int zkr_prof_enable(zkr_key *key, int on);
int zkr_prof_reset(zkr_key *key);
int zkr_prof_get(zkr_key *key, const char *stage, double *ms_total, uint64_t *launches);

/* ---- synthetic workload + device-side setup (benchmarks) -------------- */
/* Rollup-shaped seeded R1CS + witness + Groth16 key generated from seeded toxic waste, key points
 * computed ON DEVICE by fixed-base multiplication (the websnark binary is never materialised, so
 * this works past its 4 GiB u32-offset limit).  witness_out: malloc'ed nVars x 32 B std (free with
 * zkr_free).  aux_out (optional, may be NULL): malloc'ed blob for the checker --
 *   u64 nVars | 5 x 32 B toxic (t,alfa,beta,gamma,delta) | nVars x 32 B a_s | b_s | c_s  (std form)
 *   | (nPublic+1) x 64 B IC points (std affine) | 128 B vk_gamma_2 (std) */
int zkr_synth_key(unsigned log_m, unsigned n_public, uint64_t circuit_seed, uint64_t toxic_seed, int device,
                  zkr_key **key_out, void **witness_out, size_t *witness_len, void **aux_out, size_t *aux_len);
/* Verifying key of a zkr_synth_key key in the vk_bin layout of zkr_verify (malloc'ed, free with zkr_free);
 * aux = the checker blob zkr_synth_key returned for this key. */
int zkr_synth_vk(const zkr_key *key, const void *aux, size_t aux_len, void **vk_out, size_t *vk_len);
/* Same circuit + setup rendered as a websnark-format key on the host (small sizes; tests). */
int zkr_synth_websnark(unsigned log_m, unsigned n_public, uint64_t circuit_seed, uint64_t toxic_seed, int device,
                       void **pk_out, size_t *pk_len, void **witness_out, size_t *witness_len);
/* Another satisfying witness of the same synthetic circuit (host only, no GPU): structure comes from
 * circuit_seed, free values from witness_seed (zkr_synth_key uses witness_seed = circuit_seed). */
int zkr_synth_witness(unsigned log_m, unsigned n_public, uint64_t circuit_seed, uint64_t witness_seed, void **witness_out,
                      size_t *witness_len);
/* Groth16 trusted setup of an arbitrary R1CS, key elements computed on the GPU: what
 * `snarkjs setup --protocol groth -c build/tx.json --pk ... --vk ...` does in the reference's workflow
 * (prover/package.json:34,37), producing the device key directly (no JSON, no binarifyProvingKey) and the verifying key
 * in the vk_bin layout of zkr_verify.  r1cs_bin: u32 nVars | u32 nPublic (outputs + public inputs) | u32 nConstraints |
 * per constraint, for each of A, B, C: u32 k, then k x (u32 signal, 32 B coefficient, standard form LE) -- the
 * `constraints` array of circom's circuit JSON (index.js / facade.py convert it).  toxic160: t, alfa, beta, gamma,
 * delta (5 x 32 B, non-zero, < r) for reproducible test setups, or NULL to draw them from the OS CSPRNG inside the call
 * (they are wiped before it returns).  domainSize = smallest power of two >= nConstraints + nPublic + 1. */
int zkr_setup_r1cs(const void *r1cs_bin, size_t r1cs_len, const uint8_t *toxic160, int device, zkr_key **key_out, void **vk_out, size_t *vk_len);
/* The same setup, delivering the proving key as the bytes `binarifyProvingKey(provingKey)` produces from snarkjs' JSON key
 * (binarify.ts:143-206; malloc'ed, free with zkr_free; at most 4 GiB, the format's u32 offsets) instead of a device key:
 * the provingKeyBin an UNCHANGED reference caller passes to groth16GenProof on every call (common.ts:28-29). */
int zkr_setup_r1cs_websnark(const void *r1cs_bin, size_t r1cs_len, const uint8_t *toxic160, int device, void **pk_out, size_t *pk_len, void **vk_out,
                            size_t *vk_len);

/* Circuit shape drawn by the zkr_synth_* calls of the CALLING THREAD (thread-local, default 0): 0 = rollup-shaped (default; 1-3 terms
 * per row, 3 % boolean and 2 % small signals, a third of the signals absent from B), 1 = dense random (BASELINE.json
 * configs[4]: every row is (4 random signals) x (4 random signals) = new signal; no infinity points in any query). */
int zkr_synth_set_shape(unsigned shape);
void zkr_free(void *p);

/* ---- the reference's rollup circuit without circom / snarkjs (host only) ---- */
/* Witness-side crypto of operator/src/utils/crypto.ts.  Field elements are 32 B little-endian, standard form.
 * zkr_mimcsponge_multihash = multiHash (crypto.ts:28-30; MiMCSponge-220, key 0, one output; operands taken mod r);
 * zkr_babyjub_pubkey = genPublicKey (crypto.ts:78-84, including formatPrivKeyForBabyJub :58-76), pub = x | y;
 * zkr_eddsa_sign = sign (crypto.ts:143-168) over n message elements, sig = R8x | R8y | S;
 * zkr_eddsa_verify = verify (crypto.ts:170-177), *valid = 0 / 1. */
int zkr_mimcsponge_multihash(const uint8_t *in, size_t n, uint8_t out[32]);
int zkr_babyjub_pubkey(const uint8_t priv[32], uint8_t pub[64]);
int zkr_eddsa_sign(const uint8_t priv[32], const uint8_t *msg, size_t n, uint8_t sig[96]);
int zkr_eddsa_verify(const uint8_t *msg, size_t n, const uint8_t sig[96], const uint8_t pub[64], int *valid);
/* The same hash batch-parallel on the GPU (one thread per hash): out[t] = multiHash(inputs[t * arity .. + arity)), host
 * buffers, standard form; and the reference's balance tree (operator/src/utils/merkletree.ts:44-83, hashLeftRight of
 * adjacent pairs) built level by level on the device: leaves = 2^depth x 32 B, levels_out = all levels concatenated,
 * leaves first, root last ((2^(depth+1) - 1) x 32 B).  No CPU fallback (ZKR_ERR_NO_DEVICE). */
int zkr_mimcsponge_multihash_batch(const void *inputs_std, size_t count, unsigned arity, void *out_std, int device);
int zkr_balance_tree_build(const void *leaves_std, unsigned depth, void *levels_out, int device);
/* BatchProcessTx(batch, depth) (prover/circuits/batchprocesstx.circom:3-75; `tx.circom` = (2, 6)) as a rank-1
 * constraint system in the r1cs_bin layout of zkr_setup_r1cs, and its witness builder -- the counterpart of
 * `compiler(tx.circom)` + `Circuit.calculateWitness` (operator/src/snarks/common.ts:12-17).  Public signals keep
 * circom's order: newBalanceTreeRoot, then balanceTreeRoot[batch], txData[batch][8], txSenderPublicKey[batch][2],
 * txSenderBalance[batch], txSenderNonce[batch], txSenderPathElements[batch][depth], txRecipientPublicKey[batch][2],
 * txRecipientBalance[batch], txRecipientNonce[batch], txRecipientPathElements[batch][depth],
 * intermediateBalanceTreeRoot[batch], intermediateBalanceTreePathElements[batch][depth]  (73 for (2, 6)).
 * zkr_rollup_witness takes the n_public - 1 inputs in that order (32 B each) and returns the full witness
 * (nVars x 32 B, the buffer binarifyWitness would produce; free with zkr_free); inputs that violate the circuit fail
 * with ZKR_ERR_UNSATISFIED and a message naming the first violated statement.  The constraint system is this build's
 * own formulation of the circuit (see csrc/rollup.cpp): its keys come from zkr_setup_r1cs, not from a circom build. */
/* Withdraw() (prover/circuits/withdraw.circom:4-25): public signals publicKey[0], publicKey[1], nullifier; the private
 * input is the FORMATTED key, zkr_babyjub_format_privkey = formatPrivKeyForBabyJub (crypto.ts:58-76), as in
 * prover/__tests__/withdraw.test.ts:22-25.  Same conventions as the two calls below. */
int zkr_babyjub_format_privkey(const uint8_t priv[32], uint8_t out[32]);
int zkr_withdraw_r1cs(void **r1cs_bin, size_t *r1cs_len);
int zkr_withdraw_witness(const uint8_t private_key[32], const uint8_t nullifier[32], void **witness_bin, size_t *witness_len);
int zkr_rollup_info(uint32_t batch, uint32_t depth, uint32_t *n_vars, uint32_t *n_public, uint32_t *n_constraints);
int zkr_rollup_r1cs(uint32_t batch, uint32_t depth, void **r1cs_bin, size_t *r1cs_len);
int zkr_rollup_witness(uint32_t batch, uint32_t depth, const uint8_t *inputs, size_t n_inputs, void **witness_bin, size_t *witness_len);
/* The same for MANY rollup batches at once, ON THE GPU (one thread per transaction): inputs = n_batches x n_inputs x 32 B
 * (host memory), d_witnesses = device memory for n_batches x nVars x 32 B, each witness laid out as zkr_rollup_witness returns
 * it (byte for byte) and ready for zkr_prove_batch_device -- `Circuit.calculateWitness` (operator/src/snarks/common.ts:15-17)
 * for a queue of batches without touching the host cores.  ZKR_ERR_UNSATISFIED names the first batch / transaction /
 * statement that fails; ZKR_ERR_ARG an input that is not below r.  No CPU fallback (ZKR_ERR_NO_DEVICE). */
int zkr_rollup_witness_batch_device(uint32_t batch, uint32_t depth, const uint8_t *inputs, size_t n_inputs, size_t n_batches, void *d_witnesses, int device);
/* Test hook: the program one GPU thread runs per transaction (csrc/rollup_witness.hpp), run on the HOST for one batch --
 * witness_out = nVars x 32 B as zkr_rollup_witness lays it out, *stmt = code of the first violated statement (0: none;
 * zkr_rollup_statement_text gives its wording), *tx = its transaction.  Lets the CPU suite compare the GPU builder's program
 * with the host builder signal for signal. */
int zkr_rollup_witness_program_host(uint32_t batch, uint32_t depth, const uint8_t *inputs, size_t n_inputs, uint8_t *witness_out, uint32_t *stmt, uint32_t *tx);
const char *zkr_rollup_statement_text(uint32_t stmt);

/* Integer-ALU microbenchmark: sustained Fq Montgomery multiplications per second on `device` with the multiplier of the
 * hot path (9 x 29-bit limbs, 162 multiply-adds without carry words, csrc/field29.hpp); used for the secondary (VALU)
 * roofline.  _legacy: the 8 x 32-bit multiplier of csrc/field.hpp (136 multiply-adds + 136 carry additions), kept for
 * the boundary formats and the cold kernels. */
/* Limb-level self test of the hot path's Montgomery product forms (csrc/field29.hpp: 9 x 29-bit limbs, radix 2^261) as the
 * device runs them.  records: n x 8 operands x 9 raw limbs (a b c d e f g h; limbs below 2^29, the top limb free);
 * out: n x 9 raw limbs of form 0: a b, 1: a^2, 2: a b + c d, 3: a b + c d + e f + g h (x 2^-261 mod the field's modulus,
 * lazily reduced).  field 0 = Fq, 1 = Fr.  The host build of the same header (libzkr_hostarith.so zkt29_raw_forms) must
 * give the same limbs bit for bit: tests/test_gpu_stages.py. */
int zkr_selftest_f29_forms(int device, int field, int form, const uint32_t *records, size_t n, uint32_t *out);
int zkr_bench_fq_mul(int device, double *gmuls_per_s);
int zkr_bench_fq_mul_legacy(int device, double *gmuls_per_s);

#ifdef __cplusplus
}
#endif
#endif /* ZKR_H */
