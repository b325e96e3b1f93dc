O=gpurun_out/r2g; mkdir -p $O
for P in 0 1 2; do ZKR_NTT_PRIO=$P python bench.py --no-cpu-baseline --no-tx-circuit --no-bcast-modes --steps 40 > $O/bench_prio$P.json 2>/dev/null; python3 -c "
import json; d=json.load(open('$O/bench_prio$P.json')); print('prio $P', d['value'], d['stage_ms_per_proof'])"; done
