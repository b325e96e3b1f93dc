O=gpurun_out/r2h; mkdir -p $O
run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-tx-circuit --no-bcast-modes --steps 40 > $O/bench_$tag.json 2>/dev/null; python3 -c "
import json; d=json.load(open('$O/bench_$tag.json')); s=d['stage_ms_per_proof']; print('$tag', round(d['value'],1), {k: round(v,2) for k,v in s.items()})"; }
run base ZKR_NTT_PRIO=1
run accw3 ZKR_NTT_PRIO=1 ZKR_ACC_W_G1=3
run accw1 ZKR_NTT_PRIO=1 ZKR_ACC_W_G1=1
run red3 ZKR_NTT_PRIO=1 ZKR_RED_STREAMS=3
run noprio ZKR_NTT_PRIO=1 ZKR_NO_PRIO=1
run glog4 ZKR_NTT_PRIO=1 ZKR_MSM_GLOG=4
run glog3 ZKR_NTT_PRIO=1 ZKR_MSM_GLOG=3
run depth3 ZKR_NTT_PRIO=3
