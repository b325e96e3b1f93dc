#!/bin/bash
# round 5: the N = 8 code paths rehearsed on the one GPU (every rank / replica / shard on device 0; never a scaling number), then the soak
O=gpurun_out/r5_08; mkdir -p $O
S=$(date +%s); ZKR_BENCH_BACKEND=gloo ZKR_BENCH_ONE_GPU=1 python bench.py --gpus 8 --steps 4 --warmup 1 --log-m 18 > $O/bench_ranks8_one_gpu.json 2> $O/bench_ranks8.err; echo "ranks8 rc=$? $(( $(date +%s) - S )) s" > $O/summary.txt
S=$(date +%s); python bench.py --gpus 8 --inproc --devices 0,0,0,0,0,0,0,0 --steps 4 --warmup 1 --log-m 20 > $O/bench_inproc8_one_gpu.json 2> $O/bench_inproc8.err; echo "inproc8 rc=$? $(( $(date +%s) - S )) s" >> $O/summary.txt
python tools/multi_gpu_preflight.py --devices 0,0,0,0 --copy-mib 256 > $O/preflight4.jsonl 2> $O/preflight4.err; echo "preflight4 rc=$?" >> $O/summary.txt
python tests/soak.py 16 90 > $O/soak.txt 2>&1; echo "soak rc=$?" >> $O/summary.txt
cat $O/summary.txt; tail -3 $O/soak.txt; tail -1 $O/preflight4.jsonl
python - <<'P'
import json
for f in ("bench_ranks8_one_gpu", "bench_inproc8_one_gpu"):
    try:
        d = json.load(open("gpurun_out/r5_08/%s.json" % f))
        c = d["config"]
        print(f, d["n_gpus"], round(d["value"], 1), c.get("key_replication"), c.get("sharded_parts"), c.get("sharded_form"), c.get("sharded_ms"), c.get("sharded_reason"), (d.get("intra_proof_sharding") or {}).get("error"))
    except Exception as e:
        print(f, "no line:", e)
P
