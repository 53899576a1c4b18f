# How the profiles/r5_* set is produced on the GPU box (run through gpurun from the repo root; afterwards copy
# gpurun_out/r5final/* into profiles/).  Per workload: rocprofv3 --kernel-trace --stats (lanes on = the timed loop as it runs,
# and once more with --lanes 0: every kernel alone on the chip, in plan order, on fresh data -- neither warmed by back-to-back
# repeats nor stretched by a neighbour lane), two PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, no tracing), two SQ-counter
# passes for the workloads listed below, the summaries bench.py quotes (stamped with the kernel-source hash), and the bench line
# itself.  The program goes directly after `--`.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5final
if [ -z "$ONLY_BENCH" ] && [ -z "$ONLY_SET" ]; then rm -rf $O; fi      # ONLY_BENCH=1: just the bench lines again (bench.py changed, kernels did not)
mkdir -p $O
B="--no-cpu-baseline --no-extra --no-roofline"
run_set() {   # $1 = prec, $2 = clips, $3 = label, $4 = file tag (bench.py _wl_tag), $5 = 1: also the SQ-counter passes, $6... = extra bench arguments
  local P=$1 C=$2 L="$3" T=$4 SQ=$5; shift 5
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc FETCH_SIZE -d $O/f_$T -o f --output-format csv -- python3 $R/bench.py --prec $P --clips $C "$@" $B --steps 5 --windows 1 > /dev/null 2> $O/f_$T.err
  rocprofv3 --pmc WRITE_SIZE -d $O/w_$T -o w --output-format csv -- python3 $R/bench.py --prec $P --clips $C "$@" $B --steps 5 --windows 1 > /dev/null 2> $O/w_$T.err
  rocprofv3 --kernel-trace --stats -d $O/kt_$T -o kt --output-format csv -- python3 $R/bench.py --prec $P --clips $C "$@" $B > $O/kt_bench_$T.json 2> $O/kt_$T.err
  rocprofv3 --kernel-trace --stats -d $O/kl_$T -o kl --output-format csv -- python3 $R/bench.py --prec $P --clips $C "$@" $B --lanes 0 > $O/kl_bench_$T.json 2> $O/kl_$T.err
  if [ "$SQ" = "1" ]; then
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES -d $O/sa_$T -o sa --output-format csv -- python3 $R/bench.py --prec $P --clips $C "$@" $B --steps 3 --warmup 2 --windows 1 > /dev/null 2> $O/sa_$T.err
    rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES -d $O/sb_$T -o sb --output-format csv -- python3 $R/bench.py --prec $P --clips $C "$@" $B --steps 3 --warmup 2 --windows 1 > /dev/null 2> $O/sb_$T.err
  fi
  cd $R
  python3 tools/traffic_report.py $O/f_$T/f_counter_collection.csv $O/w_$T/w_counter_collection.csv $O/r5_hbm_traffic_$T.json "$L" > $O/r5_hbm_traffic_$T.txt
  python3 tools/kernel_stats_report.py $O/kt_$T/kt_kernel_stats.csv $O/r5_kernel_stats_$T.json "$L" > $O/r5_kernel_stats_$T.txt
  python3 tools/kernel_stats_report.py $O/kl_$T/kl_kernel_stats.csv $O/r5_kernel_stats_${T}_lanes0.json "$L --lanes 0" > $O/r5_kernel_stats_${T}_lanes0.txt
  cp $O/kt_$T/kt_kernel_stats.csv $O/r5_bench_${T}_kernel_stats.csv
  cp $O/kl_$T/kl_kernel_stats.csv $O/r5_bench_${T}_lanes0_kernel_stats.csv
  python3 tools/step_timeline.py $O/kt_$T/kt_kernel_trace.csv 3 --list > $O/r5_step_timeline_$T.txt 2>/dev/null || true
  if [ "$SQ" = "1" ]; then
    python3 tools/pmc_by_kernel.py $O/sa_$T/sa_counter_collection.csv > $O/r5_sq_pmc_${T}_pass_a.txt
    python3 tools/pmc_by_kernel.py $O/sb_$T/sb_counter_collection.csv > $O/r5_sq_pmc_${T}_pass_b.txt
  fi
  cp $O/r5_hbm_traffic_$T.json $O/r5_kernel_stats_$T.json $O/r5_kernel_stats_${T}_lanes0.json profiles/     # so that the bench line below can quote them
  rm -rf $O/f_$T $O/w_$T $O/kt_$T/kt_kernel_trace.csv $O/kl_$T/kl_kernel_trace.csv $O/sa_$T $O/sb_$T
}
# ONLY_SET=<file tag>: re-collect one workload (its plan changed, the kernels did not) and its bench line
want() { [ -z "$ONLY_SET" ] || [ "$ONLY_SET" = "$1" ]; }
if [ -z "$ONLY_BENCH" ]; then
want f32_c1 && run_set f32 1 "bench.py (configs[1]: 360x640, 1 clip x 8 frames, f32)" f32_c1 1
want f32_c8 && run_set f32 8 "bench.py --clips 8 (one GPU's share of configs[3]: 360x640, 8 clips x 8 frames, f32)" f32_c8 0
want f16x3_c8 && run_set f16x3 8 "bench.py --prec f16x3 --clips 8 (configs[2]: 360x640, 8 clips x 8 frames, split-fp16 MFMA)" f16x3_c8 1
want f32_c4_720x1280_t16 && run_set f32 4 "bench.py --height 720 --width 1280 --frames 16 --clips 4 --persistent-state 1 (configs[4])" f32_c4_720x1280_t16 0 --height 720 --width 1280 --frames 16 --persistent-state 1 --steps 5 --warmup 2
fi
cd $R
want f32_c1 && python3 bench.py > $O/r5_bench_default.json 2> $O/r5_bench_default.err
want f32_c1 && python3 bench.py --inflight 2 --no-extra > $O/r5_bench_inflight2.json 2> $O/r5_bench_inflight2.err
want f32_c8 && python3 bench.py --clips 8 --no-extra --no-cpu-baseline > $O/r5_bench_f32_c8.json 2> $O/r5_bench_f32_c8.err
want f16x3_c8 && python3 bench.py --prec f16x3 --clips 8 --no-extra --no-cpu-baseline > $O/r5_bench_f16x3_c8.json 2> $O/r5_bench_f16x3_c8.err
want f16x3_c8 && python3 bench.py --prec f16x3 --no-extra --no-cpu-baseline > $O/r5_bench_f16x3_c1.json 2> $O/r5_bench_f16x3_c1.err
want f32_c4_720x1280_t16 && python3 bench.py --height 720 --width 1280 --frames 16 --clips 4 --persistent-state 1 --steps 5 --warmup 2 --no-extra --no-cpu-baseline > $O/r5_bench_720p_c4_t16.json 2> $O/r5_bench_720p_c4_t16.err
ls $O
