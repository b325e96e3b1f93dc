mkdir -p gpurun_out/r2i
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r2i/pytest.log; tail -3 gpurun_out/r2i/pytest.log
bash tools/profile_round2.sh r2i
