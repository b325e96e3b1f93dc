// Types for the Node.js host of the MI355X Groth16 prover (mirrors the websnark / snarkjs surfaces the
// reference uses: operator/src/snarks/common.ts:4-5,23,29).
export interface Groth16Proof {
  pi_a: [string, string, string];
  pi_b: [[string, string], [string, string], [string, string]];
  pi_c: [string, string, string];
  protocol?: "groth";
}
/** devices (groth16GenProof): ONE proof over these GPUs -- the key is cut into devices.length shards (cached), the proof is the same bytes. */
export interface ProveOptions { r?: bigint | string; s?: bigint | string; device?: number; devices?: number[]; }
/** devices: HIP ordinals of the GPUs a batch is sharded over (a device may be listed twice: two proof pipelines on it). */
export interface BatchOptions { devices?: number[]; blinding?: ProveOptions[]; }
export interface Bn128 {
  /** websnark-compatible: ArrayBuffers produced by binarifyWitness / binarifyProvingKey. */
  groth16GenProof(witnessBin: ArrayBuffer | Uint8Array, provingKeyBin: ArrayBuffer | Uint8Array, opts?: ProveOptions): Promise<Groth16Proof>;
  /** Independent proofs on one key in one native call: pipelined two deep, proofs of small circuits fused into shared launches. */
  /** opts: per-proof blinding (array), or { devices, blinding }: the batch is sharded over the listed GPUs of this node
   *  (proof i on devices[i % devices.length]; the key is parsed once and copied device to device to the others). */
  groth16GenProofBatch(witnessBins: Array<ArrayBuffer | Uint8Array>, provingKeyBin: ArrayBuffer | Uint8Array, opts?: ProveOptions[] | BatchOptions): Promise<Groth16Proof[]>;
  /** The same with the key currently held on the device. */
  proveBatch(witnessBins: Array<ArrayBuffer | Uint8Array>, opts?: ProveOptions[] | BatchOptions): Promise<Groth16Proof[]>;
  /** Groth16 setup of circom's circuit JSON on the GPU (snarkjs setup --protocol groth); returns the verifying key JSON. */
  setup(circuitDef: any, opts?: { toxic?: Array<bigint | string> }): any;
  /** The same for a constraint system in the r1cs_bin layout (RollupCircuit.r1cs()). */
  setupR1cs(r1csBin: Uint8Array, opts?: { toxic?: Array<bigint | string> }): any;
  saveKey(path: string): void;
  loadKeyFile(path: string): void;
  /** Proof with the key currently held on the device. */
  prove(witnessBin: ArrayBuffer | Uint8Array, opts?: ProveOptions): Promise<Groth16Proof>;
  keyInfo(): { nVars: number; nPublic: number; domainSize: number; nnzA: number; nnzB: number } | null;
  terminate(): void;
}
export function buildBn128(device?: number): Promise<Bn128>;
export function genProof(provingKey: any, witness: Array<bigint | string>, opts?: ProveOptions): Promise<{ proof: Groth16Proof; publicSignals: string[] }>;
export function binarifyWitness(witness: Array<bigint | string>): ArrayBuffer;
export function binarifyProvingKey(provingKey: any): ArrayBuffer;
export function solidityProof(proof: Groth16Proof, publicSignals: Array<bigint | string>): { a: string[]; b: string[][]; c: string[]; inputs: string[] };
/** snarkjs groth.isValid(vk, proof, publicSignals) on the native host verifier (no GPU needed). */
export function isValid(verifyingKey: any, proof: Groth16Proof, publicSignals: Array<bigint | string>): boolean;
/** All proofs of a batch under one key with one merged pairing product (random linear combination). */
export function isValidBatch(verifyingKey: any, proofs: Groth16Proof[], publicSignalsList: Array<Array<bigint | string>>): boolean;
export function binarifyVerifyingKey(verifyingKey: any): Uint8Array;
export function binarifyR1cs(circuitDef: any): Uint8Array;
export function verifyingKeyFromBytes(vkBin: Uint8Array): any;
export function proofFromBytes(proofBytes: Uint8Array): Groth16Proof;
export function deviceCount(): number;
export function version(): string;

// ---- the rollup circuit without circom / snarkjs (operator/src/utils/crypto.ts, prover/circuits/batchprocesstx.circom)
export interface Signature { R8: [bigint, bigint]; S: bigint; }
export function multiHash(values: Array<bigint | string | number>): bigint;
export function hashLeftRight(left: bigint, right: bigint): bigint;
/** multiHash of every row on the GPU (one thread per hash). */
export function multiHashBatch(rows: Array<Array<bigint | string | number>>, device?: number): bigint[];
/** The reference's balance tree over the leaves (padded to 2^depth), hashed on the GPU. */
export function buildBalanceTree(depth: number, leaves: Array<bigint | string>, zeroValue?: bigint, device?: number): { depth: number; levels: bigint[][]; root: bigint; path(i: number): bigint[] };
export function genPublicKey(privKey: bigint): [bigint, bigint];
export function formatPrivKeyForBabyJub(privKey: bigint): bigint;
export function sign(privKey: bigint, msg: Array<bigint | string | number>): Signature;
export function verify(msg: Array<bigint | string | number>, sig: Signature, pubKey: [bigint, bigint]): boolean;
/** BatchProcessTx(batch, depth) (tx.circom = (2, 6)) as constraint system + witness builder. */
export class RollupCircuit {
  constructor(batch?: number, depth?: number);
  readonly batch: number; readonly depth: number; readonly nVars: number; readonly nPublic: number; readonly nConstraints: number;
  r1cs(): Uint8Array;
  /** circuitInputs as the reference builds them (one array per input signal over the batch); throws where Circuit.calculateWitness would. */
  calculateWitness(circuitInputs: { [signal: string]: any } | Array<bigint | string>): ArrayBuffer;
  publicSignals(witnessBin: ArrayBuffer): bigint[];
}
/** Withdraw() (withdraw.circom): public signals publicKey[0], publicKey[1], nullifier. */
export class WithdrawCircuit {
  readonly nPublic: number;
  r1cs(): Uint8Array;
  calculateWitness(circuitInputs: { privateKey: bigint | string; nullifier: bigint | string }): ArrayBuffer;
  publicSignals(witnessBin: ArrayBuffer): bigint[];
}

// ---- process-level key cache (the reference builds a new Bn128 per proof: common.ts:23) and verifier constants
/** Device keys loaded / found in the cache by groth16GenProof so far in this process. */
export function keyCacheStats(): { loads: number; hits: number; replications: number; shardings?: number; entries: number; handles: number; shardedLastForm: ShardedForm };
/** Which form the last sharded proof (groth16GenProof with opts.devices) took and why: calcH split over the shards, or computed by every shard for itself. */
export interface ShardedForm { form: "none" | "split" | "replicated"; reason: string }
export function shardedLastForm(): ShardedForm;
/** How a native key handle came to its device: loaded there ("none") or copied device to device (zkr_key_replicate) in the full or the compact form. */
export function keyReplication(key: unknown): { mode: "none" | "full" | "base"; peerDirect: boolean };
export function keyFingerprint(provingKeyBin: ArrayBuffer | Uint8Array, full?: boolean): string;
export function clearKeyCache(): void;
/** vk_bin -> the constants of the generated verifier's verifyingKey() in the contract's encoding (G2 as [im, re]). */
export function solidityVerifyingKey(vkBin: Uint8Array): { alfa1: string[]; beta2: string[][]; gamma2: string[][]; delta2: string[][]; IC: string[][] };
/** The Solidity statements of verifyingKey() for those constants (what `snarkjs generateverifier` emits). */
export function solidityVerifyingKeySource(vkBin: Uint8Array, indent?: string): string;
