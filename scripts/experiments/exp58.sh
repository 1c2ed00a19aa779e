# round 5: the fuzzers over the final binary (new back-substitution, packed H, two-team sweep): LM path, clips, sharded, windows, solves
O=gpurun_out/exp58; mkdir -p $O
timeout 1500 python tests/tools/fuzz_lm_path.py 0 300 > $O/fuzz_lm_path.log 2>&1; tail -2 $O/fuzz_lm_path.log
timeout 1500 python tests/tools/fuzz_clips_path.py 0 200 > $O/fuzz_clips_path.log 2>&1; tail -2 $O/fuzz_clips_path.log
timeout 1500 python tests/tools/fuzz_sharded_path.py 0 80 > $O/fuzz_sharded_path.log 2>&1; tail -2 $O/fuzz_sharded_path.log
timeout 1500 python tests/tools/fuzz_window_path.py 0 100 > $O/fuzz_window_path.log 2>&1; tail -2 $O/fuzz_window_path.log
timeout 1500 python tests/tools/fuzz_solves.py 300 40 > $O/fuzz_solves.log 2>&1; tail -2 $O/fuzz_solves.log
