O=gpurun_out/r2j; mkdir -p $O
for C in 20 22 21; do ZKR_MSM_C=$C python bench.py --log-m 24 --steps 4 --warmup 1 --no-cpu-baseline --no-tx-circuit --no-bcast-modes > $O/bench_2_24_c$C.json 2>/dev/null; python3 -c "
import json; d=json.load(open('$O/bench_2_24_c$C.json')); print('c=$C', round(d['value'],2), {k: round(v,1) for k,v in d['stage_ms_per_proof'].items()}, d['key']['setup_s'])"; done
ZKR_MSM_C=22 ZKR_SERIAL=1 python bench.py --log-m 24 --steps 2 --warmup 1 --no-pipeline --no-cpu-baseline --no-tx-circuit --no-bcast-modes > $O/bench_2_24_c22_serial.json 2>/dev/null; python3 -c "
import json; d=json.load(open('$O/bench_2_24_c22_serial.json')); print('c=22 serial', round(d['value'],2), {k: round(v,1) for k,v in d['stage_ms_per_proof'].items()})"
