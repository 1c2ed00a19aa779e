#!/bin/bash
# One GPU-box round: tests, smoke, bench, rocprof kernel stats, PMC counters.  Outputs under gpurun_out/round/;
# copy the summaries to profiles/<round>/ afterwards (scripts/collect_profiles.sh).
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/round
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 2400 python -m pytest tests -m gpu -q -rs -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log
export TMPDIR=/tmp
cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-secondary"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/rocprof_pmc_fetch -o fetch -- $B --steps 3 --warmup 1 > /dev/null 2> $OUT/rocprof_pmc_fetch.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/rocprof_pmc_write -o write -- $B --steps 3 --warmup 1 > /dev/null 2> $OUT/rocprof_pmc_write.err
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $OUT/rocprof_pmc_sq -o sq -- $B --steps 3 --warmup 1 > /dev/null 2> $OUT/rocprof_pmc_sq.err
cd $R
python scripts/summarize_pmc.py $OUT $OUT/summary > $OUT/summary.log 2>&1
# the bench line LAST: it quotes the PMC summary of this very binary (bench.py reads profiles/<PROFILE_DIR>, build-id checked)
mkdir -p $R/profiles/round6 && cp $OUT/summary/pmc_traffic.json $OUT/summary/pmc_mfma_lds.json $R/profiles/round6/ 2>/dev/null
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rocprof_stats -o stats -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/bench_under_rocprof.json 2> $OUT/rocprof_stats.err)
timeout 900 python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
find $OUT/rocprof_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/summary/rocprofv3_kernel_stats.csv
# the widened rows and the secondary configs: kernel stats of a bench run WITH the secondary metrics (config 2's fused kernel,
# the config-5 clip chain, k_sba_*, k_skel_*), the SBA / EKF reports, the wait-state counters of every FTE kernel
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rocprof_all -o all -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2> $OUT/rocprof_all.err
cd $R
find $OUT/rocprof_all -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/summary/rocprofv3_kernel_stats_with_secondary_configs.csv
rm -rf $OUT/rocprof_all
timeout 600 python scripts/extras_report.py $OUT/summary > $OUT/extras.log 2>&1
timeout 900 bash scripts/pmc_waits.sh > /dev/null 2>&1; cp $R/gpurun_out/waits/waits.txt $OUT/summary/pmc_wait_states.txt 2>/dev/null
timeout 900 python scripts/shard_model.py 2>&1 | grep -v amdgpu.ids > $OUT/summary/shard_model.txt
# config 5's SBA half at full size: kernel stats and wait states of the fused kernels
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rocprof_sba -o sba -- python $R/scripts/sba_config5.py f64 10 > /dev/null 2> $OUT/rocprof_sba.err)
find $OUT/rocprof_sba -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/summary/sba_config5_kernel_stats.csv; rm -rf $OUT/rocprof_sba
timeout 600 bash scripts/pmc_waits_sba.sh > /dev/null 2>&1; cp $R/gpurun_out/waits_sba/waits.txt $OUT/summary/pmc_wait_states_sba.txt 2>/dev/null; rm -rf $R/gpurun_out/waits_sba/p1 $R/gpurun_out/waits_sba/p2
timeout 300 python scripts/e2e_phases.py > $OUT/summary/e2e_phases.txt 2>&1
# the widened rows' kernels (k_ekf_*, k_skel_*, k_triangulate_*): kernel stats, wait states, HBM bytes per launch
timeout 1500 bash scripts/pmc_waits_any.sh extras python $R/scripts/extras_workload.py > /dev/null 2>&1
cp $R/gpurun_out/waits_extras/waits.txt $OUT/summary/pmc_wait_states_extras.txt 2>/dev/null; cp $R/gpurun_out/waits_extras/kernel_stats.csv $OUT/summary/extras_kernel_stats.csv 2>/dev/null
timeout 120 python scripts/backsub_stamps.py 100 2>&1 | grep -v amdgpu.ids > $OUT/summary/backsub_step_stamps.txt
timeout 120 python scripts/sweep_stamps.py 100 3 2>&1 | grep -v amdgpu.ids > $OUT/summary/sweep_node_stamps.txt
# round 6: the separator chain's kernels from the inside, short chains, the soak table, the skeleton probes
timeout 120 python scripts/sep_stamps.py 2>&1 | grep -v amdgpu.ids > $OUT/summary/separator_chain_stamps.txt
timeout 300 python scripts/short_chain_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/summary/short_chain_probe.txt
(timeout 600 python scripts/soak.py; timeout 600 python scripts/chunk_stress.py) 2>&1 | grep -v amdgpu.ids > $OUT/summary/soak_and_stress_final_binary.txt
timeout 300 python scripts/schwarz_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/summary/skeleton_schwarz_fixed_point.txt
timeout 300 python scripts/video_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/summary/skeleton_whole_video_windows.txt
(timeout 200 python scripts/skel_solve_probe.py; echo "--- the round-5 kernel (ACINO_SKEL_OLD_SOLVE=1):"; ACINO_SKEL_OLD_SOLVE=1 timeout 200 python scripts/skel_solve_probe.py) 2>&1 | grep -v amdgpu.ids > $OUT/summary/skeleton_solve_per_frame.txt
(/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I acinoset_amd/csrc -I include scripts/bench/strip_phase.hip -o /tmp/strip_phase 2>/dev/null && /tmp/strip_phase) > $OUT/summary/strip_phase_microbenchmark.txt 2>&1
(for a in "10000 loop 20210313 2:3,1:6,1:7,1:8,1:12" "3331 loop 5 4:3,3:7,2:10" "999 loop 4 4:3,3:7,2:12" "6000 walk 3 3:3,2:7,1:7" "4000 sprint 1 4:3,3:7,2:7" "400 sprint 4 4:3,3:7" "20000 trot 6 2:3,1:7"; do timeout 300 python scripts/levels_probe.py $a; done) 2>&1 | grep -v amdgpu.ids > $OUT/summary/levels_against_sweeps.txt
(timeout 900 python scripts/tail_race_probe.py 3331 2000 40; timeout 600 python scripts/tail_race_probe.py 10000 1000 20) 2>&1 | grep -v amdgpu.ids > $OUT/summary/tail_handoff_repeatability.txt
(/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I acinoset_amd/csrc -I include scripts/bench/chain16.hip -o /tmp/chain16 2>/dev/null && /tmp/chain16) > $OUT/summary/pivot_chain_microbenchmark.txt 2>&1
cp $OUT/bench.json $OUT/summary/bench_n1.json; cp $OUT/bench_under_rocprof.json $OUT/summary/bench_n1_under_rocprof.json
cp $OUT/pytest_gpu.log $OUT/smoke.log $OUT/summary/
# keep the merge-back small: the raw counter dumps are not needed once summarised
rm -rf $OUT/rocprof_pmc_fetch $OUT/rocprof_pmc_write $OUT/rocprof_pmc_sq
find $OUT/rocprof_stats -type f ! -name "*kernel_stats.csv" -delete
tail -4 $OUT/pytest_gpu.log; tail -2 $OUT/smoke.log; cut -c1-700 $OUT/bench.json; ls $OUT/summary
