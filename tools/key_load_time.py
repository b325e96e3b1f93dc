"""Time of zkr_key_load_websnark (what the first groth16GenProof of a process pays) for the tx circuit's provingKeyBin and a
2^20 key: python tools/key_load_time.py   (under rocprofv3 --kernel-trace --stats for the kernel share)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "simple-zk-rollups_amd", "python"))
import zkr_hip
from zkr_hip import rollup

circ = rollup.RollupCircuit(2, 6)
pk, vk = zkr_hip.setup_r1cs_websnark(circ.r1cs(), device=0)
for rep in range(3):
    t = time.perf_counter()
    k = zkr_hip.ProvingKey.load_websnark(pk, device=0)
    dt = time.perf_counter() - t
    k.close()
    print("tx circuit provingKeyBin (%.0f MB): zkr_key_load_websnark %.1f ms" % (len(pk) / 1e6, 1e3 * dt))
if len(sys.argv) > 1:
    pkb = zkr_hip.binding.synth_websnark(int(sys.argv[1]), 73, 0x5A4B0001, 0x5A4B00FF, 0)[0]
    for rep in range(2):
        t = time.perf_counter()
        k = zkr_hip.ProvingKey.load_websnark(pkb, device=0)
        dt = time.perf_counter() - t
        k.close()
        print("2^%s provingKeyBin (%.0f MB): zkr_key_load_websnark %.1f ms" % (sys.argv[1], len(pkb) / 1e6, 1e3 * dt))
