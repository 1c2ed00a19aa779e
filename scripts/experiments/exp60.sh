# round 5: soak / chunk stress / tail stress of the final binary (-> profiles/round5/soak_and_stress_final_binary.txt)
O=gpurun_out/exp60; mkdir -p $O
(timeout 600 python scripts/soak.py; timeout 900 python scripts/chunk_stress.py 5 120; timeout 600 python scripts/tail_stress.py) 2>&1 | grep -v amdgpu.ids > $O/soak_and_stress.txt
tail -12 $O/soak_and_stress.txt
