O=gpurun_out/r2c; mkdir -p $O
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/valu_clock.hip -o /tmp/valu_clock 2>/dev/null && /tmp/valu_clock > $O/valu_clock.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q -k "not config4 and not config3" 2>&1 | tail -15 > $O/pytest.log; tail -3 $O/pytest.log
python - > $O/omp_scaling.txt 2>&1 <<'PY'
import sys, time, os
sys.path.insert(0, "oracle"); sys.path.insert(0, "simple-zk-rollups_amd/python")
import coracle, zkr_hip
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "omp max", coracle.max_threads())
try:
    print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("cpu.max n/a", e)
pkb, wb = zkr_hip.synth_websnark(17, 73, 0x5A4B0001, 0x5A4B00FF)
for th in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    t0 = time.time(); p, tm = coracle.prove_mt(pkb, wb, 5, 7, threads=th, want_timings=True); dt = time.time() - t0
    print("threads %3d: %.3f s (calcH %.3f, msm %.3f) used %d" % (th, dt, tm[0], tm[1], tm[3]))
PY
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err; head -c 300 $O/bench.json
