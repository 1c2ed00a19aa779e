# round 5: flake hunt on the final binary: the chunk tests (incl. the bit-reproducibility soak) five times, the whole GPU suite twice
O=gpurun_out/exp61; mkdir -p $O
for i in 1 2 3 4 5; do timeout 900 python -m pytest tests/test_gpu_chunk.py -m gpu -x -q 2>&1 | tail -1; done | tee $O/chunk_x5.log
for i in 1 2; do timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -1; done | tee $O/suite_x2.log
