mkdir -p gpurun_out/r2v
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r2v/pytest.log; tail -3 gpurun_out/r2v/pytest.log
bash tools/profile_round2.sh r2v
bash tools/profile_counters_r2.sh r2v_pmc > /dev/null 2>&1
bash tools/profile_timeline.sh > /dev/null 2>&1
