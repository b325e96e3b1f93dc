O=gpurun_out/r2u; mkdir -p $O
run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-tx-circuit --no-bcast-modes --steps 40 > $O/bench_$tag.json 2>/dev/null; python3 -c "
import json; d=json.load(open('$O/bench_$tag.json')); s=d['stage_ms_per_proof']; print('$tag', round(d['value'],1), {k: round(v,2) for k,v in s.items()})"; }
run base ZKR_ACC_PRIO=0
run acc1 ZKR_ACC_PRIO=1
run acc2 ZKR_ACC_PRIO=2
run acc3 ZKR_ACC_PRIO=3
run acc2ntt0 ZKR_ACC_PRIO=2 ZKR_NTT_PRIO=0
run acc1ntt2 ZKR_ACC_PRIO=1 ZKR_NTT_PRIO=2
