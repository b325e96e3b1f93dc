"""zkr_hip -- Python host side of the MI355X Groth16 prover (ctypes over the C ABI in include/zkr.h).

This is plumbing for tests, bench.py and the multi-GPU batch driver; the product is
csrc/libzkr_hip.so.  There is no CPU fallback: if the library or a HIP device is missing every
compute entry point raises ZkrError.

Mirrors the reference's proof facade for the hot path
(/root/reference/operator/src/snarks/common.ts:10-53): see facade.py.
"""
from .binding import (ZkrError, ProvingKey, lib, device_count, device_pci_bus_id, version, ntt, msm_g1, msm_g2, bench_fq_mul, selftest_f29_forms,
                      synth_websnark, synth_witness, synth_set_shape, verify, verify_batch, setup_r1cs_websnark, prove_batch_multi, prove_batch_multi_device, prove_sharded, prove_sharded_device, sharded_split_stats, sharded_last_form, PROOF_BYTES, PARTIAL_BYTES, STAGES)
from .facade import (build_bn128, groth16_gen_proof, proof_json_from_bytes, proof_bytes_from_json, solidity_proof, create_proof_generator,
                     binarify_verifying_key, binarify_r1cs, verifying_key_from_bytes, is_valid, cached_key, cached_replicas, cached_shards, key_fingerprint, clear_key_cache, key_cache_stats,
                     solidity_verifying_key, solidity_verifying_key_source)
from .batch import shard_indices, broadcast_key, replicate_key, broadcast_arena, prove_batch, gather_proofs

__all__ = ["ZkrError", "ProvingKey", "lib", "device_count", "device_pci_bus_id", "version", "ntt", "msm_g1", "msm_g2", "bench_fq_mul", "selftest_f29_forms",
           "synth_websnark", "synth_witness", "setup_r1cs_websnark", "cached_key", "cached_replicas", "cached_shards", "key_fingerprint", "clear_key_cache", "key_cache_stats", "solidity_verifying_key", "solidity_verifying_key_source", "synth_set_shape", "verify", "verify_batch", "binarify_verifying_key", "binarify_r1cs", "verifying_key_from_bytes", "is_valid", "proof_bytes_from_json", "PROOF_BYTES", "STAGES", "build_bn128", "groth16_gen_proof", "proof_json_from_bytes",
           "solidity_proof", "create_proof_generator", "prove_batch_multi", "prove_batch_multi_device", "prove_sharded", "prove_sharded_device", "sharded_split_stats", "sharded_last_form", "PARTIAL_BYTES", "shard_indices", "replicate_key", "broadcast_key", "broadcast_arena", "prove_batch", "gather_proofs"]
