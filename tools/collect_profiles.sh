#!/bin/bash
# Collects the round's measured artefacts on the GPU box into gpurun_out/$ROUND/ (copied into profiles/ afterwards by tools/assemble_profiles.py).
#   [ROUND=r05] bash tools/collect_profiles.sh [part]     part = bench | prof | mfma | tools | all
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
ROUND=${ROUND:-r05}
O=$R/gpurun_out/$ROUND
mkdir -p $O
part=${1:-all}
cd $R
if [ "$part" = bench ] || [ "$part" = all ]; then
  timeout -k 10 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err
  timeout -k 10 300 python bench.py --pattern-group 1 --no-cpu-baseline > $O/bench_cfg2_pattern_per_frame.json 2>> $O/bench.err
  timeout -k 10 300 python bench.py --pattern-group 1 --patterns-resident --no-cpu-baseline > $O/bench_cfg2_pattern_per_frame_resident.json 2>> $O/bench.err
  timeout -k 10 300 python bench.py --workload cfg1_siso_epa_ls --no-cpu-baseline > $O/bench_cfg1.json 2>> $O/bench.err
  timeout -k 10 300 python bench.py --workload cfg3_4x4_etu_mmse --no-cpu-baseline > $O/bench_cfg3.json 2>> $O/bench.err
  timeout -k 10 500 python bench.py --workload cfg4_dataset_mixed > $O/bench_cfg4.json 2>> $O/bench.err
  timeout -k 10 300 python bench.py --workload cfg4_dataset_mixed --pattern-group 1 --no-cpu-baseline > $O/bench_cfg4_pattern_per_frame.json 2>> $O/bench.err
  timeout -k 10 900 python bench.py --workload cfg5_4x4_2048_fullcov > $O/bench_cfg5.json 2>> $O/bench.err
  timeout -k 10 300 python bench.py --rccl-rehearsal --cpu-budget-s 5 --no-extra-legs > $O/bench_cfg2_rccl_one_rank.json 2>> $O/bench.err
  # the N = 2 line end to end, two gloo ranks sharing the one GPU: collective_ms, per_rank and cpu_baseline at N > 1
  timeout -k 10 300 python bench.py --gpus 2 --dist-backend gloo --cpu-budget-s 5 > $O/bench_cfg2_gloo2_one_gpu.json 2>> $O/bench.err
  timeout -k 10 300 python bench.py --gpus 2 --dist-backend gloo --workload cfg5_4x4_2048_fullcov --no-cpu-baseline > $O/bench_cfg5_gloo2_one_gpu.json 2>> $O/bench.err
  echo bench done
fi
if [ "$part" = prof ] || [ "$part" = all ]; then
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg2 -- python3 $R/bench.py --no-cpu-baseline --no-extra-legs > $O/prof_cfg2.json 2> $O/prof_cfg2.err
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg2pf -- python3 $R/bench.py --no-cpu-baseline --pattern-group 1 --steps 10 > $O/prof_cfg2pf.json 2> $O/prof_cfg2pf.err
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --no-cpu-baseline --no-extra-legs --steps 3 --warmup 1 > $O/pmc_fetch.json 2> $O/pmc_fetch.err
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --no-cpu-baseline --no-extra-legs --steps 3 --warmup 1 > $O/pmc_write.json 2> $O/pmc_write.err
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg5 -- python3 $R/bench.py --workload cfg5_4x4_2048_fullcov --no-cpu-baseline --steps 10 > $O/prof_cfg5.json 2> $O/prof_cfg5.err
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg4pf -- python3 $R/bench.py --workload cfg4_dataset_mixed --pattern-group 1 --no-cpu-baseline --steps 10 > $O/prof_cfg4pf.json 2> $O/prof_cfg4pf.err
  cd $R
  echo prof done
fi
if [ "$part" = mfma ] || [ "$part" = all ]; then
  # MFMA-pipe busy of the full-covariance filter GEMM at the bench's 2 304 frames (counters in a run of their own, kernel trace only)
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -- python3 $R/bench.py --workload cfg5_4x4_2048_fullcov --no-cpu-baseline --steps 5 --warmup 2 > $O/pmc_mfma.json 2> $O/pmc_mfma.err
  cd $R
  python3 tools/assemble_profiles.py --round $ROUND --only-mfma > $O/pmc_mfma_cgemm.txt 2>&1
  echo mfma done
fi
if [ "$part" = tools ] || [ "$part" = all ]; then
  timeout -k 10 200 python tools/stamps_patterns.py > $O/stamps_pattern_kernel.txt 2>&1
  timeout -k 10 200 python tools/bench_patterns.py > $O/bench_patterns.txt 2>&1
  bash tools/pmc_pattern_phases.sh 838 > $O/pmc_pattern_phases_P838.txt 2>&1
  bash tools/pmc_pattern_phases.sh 83 > $O/pmc_pattern_phases_P83.txt 2>&1
  bash tools/pmc_pattern_sq.sh > $O/pmc_sq_pattern_kernel.txt 2>&1
  (timeout -k 10 200 python tools/plan_parity.py --pilots 838 --patterns 64; timeout -k 10 200 python tools/plan_parity.py --pilots 419 --patterns 64;
   timeout -k 10 300 python tools/plan_parity.py --pilots 838 --method nearest; timeout -k 10 300 python tools/plan_parity.py --pilots 419 --method nearest) > $O/plan_parity.txt 2>&1
  timeout -k 10 200 python tools/bench_cubic.py > $O/bench_cubic.txt 2>&1
  timeout -k 10 200 python tools/dropin_latency.py --frames 100 > $O/dropin_latency.txt 2>&1
  echo tools done
fi
