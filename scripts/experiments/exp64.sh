# round 5: final refresh of the test log and the bench line (python-side additions since profiles/round5 was taken; same binary)
O=gpurun_out/exp64; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -rs -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
