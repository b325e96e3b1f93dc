O=gpurun_out/r2l; mkdir -p $O
run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-tx-circuit --no-bcast-modes --steps 40 > $O/bench_$tag.json 2>/dev/null; python3 -c "
import json; d=json.load(open('$O/bench_$tag.json')); s=d['stage_ms_per_proof']; print('$tag', round(d['value'],1), {k: round(v,2) for k,v in s.items()})"; }
L=simple-zk-rollups_amd/csrc
run redw3 ZKR_HIP_LIB=$L/libzkr_hip.so
run redw2 ZKR_HIP_LIB=$L/libzkr_hip_redw2.so
run redw1 ZKR_HIP_LIB=$L/libzkr_hip_redw1.so
run redw3_accg2w1 ZKR_HIP_LIB=$L/libzkr_hip.so ZKR_ACC_W_G2=1
run redw3_again ZKR_HIP_LIB=$L/libzkr_hip.so
