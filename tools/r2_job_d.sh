O=gpurun_out/r2d; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q -k "not config4 and not config3" 2>&1 | tail -15 > $O/pytest.log; tail -3 $O/pytest.log
timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err; head -c 400 $O/bench.json
