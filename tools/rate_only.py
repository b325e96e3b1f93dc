"""Pipelined proofs/s of the synthetic 2^log_m rollup circuit and the average launch of the G1 accumulation, WITHOUT verifying
anything: for A/B runs of measurement-only builds whose sums are garbage (ZKR_HIP_LIB=tools/bin/libzkr_hip_<variant>.so).
python tools/rate_only.py [log_m=20] [steps=30] [tag]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "simple-zk-rollups_amd", "python"))
import torch
import zkr_hip

log_m = int(sys.argv[1]) if len(sys.argv) > 1 else 20
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
tag = sys.argv[3] if len(sys.argv) > 3 else os.path.basename(os.environ.get("ZKR_HIP_LIB", "default"))
key, w0, _ = zkr_hip.ProvingKey.synth(log_m, device=0, want_aux=False)
wits = [torch.frombuffer(bytearray(w0), dtype=torch.uint8).cuda(0)] + [
    torch.frombuffer(bytearray(zkr_hip.synth_witness(log_m, 73, 0x5A4B0001, 0x5A4B0001 + i)), dtype=torch.uint8).cuda(0) for i in range(1, 4)]
stream = torch.cuda.current_stream().cuda_stream
run = lambda n: key.prove_batch_device([wits[i % 4].data_ptr() for i in range(n)], [1000003 + i for i in range(n)], [2000003 + i for i in range(n)], stream)
run(6)
key.prof_enable(True)
key.prof_reset()
torch.cuda.synchronize()
t = time.perf_counter()
run(steps)
torch.cuda.synchronize()
el = time.perf_counter() - t
pr = key.prof()
g1, g2 = pr["msm_accum_g1"], pr["msm_accum_g2"]
print("[%s] 2^%d: %.2f proofs/s (%.3f ms per proof); msm_accum_kernel<Fq> %.4f ms per launch (%d), <Fq2> %.4f ms" % (
    tag, log_m, steps / el, 1e3 * el / steps, g1[0] / max(g1[1], 1), g1[1], g2[0] / max(g2[1], 1)))
