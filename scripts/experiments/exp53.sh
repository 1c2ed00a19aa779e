# round 5: generic-skeleton FTE batched over clips with the controller on the device: tests, the whole shipped video
O=gpurun_out/exp53; mkdir -p $O
timeout 1500 python -m pytest tests/test_skel_fte.py -m gpu -x -q -s > $O/skel_tests.log 2>&1; echo "rc=$?" >> $O/skel_tests.log; grep -v amdgpu.ids $O/skel_tests.log | tail -25
