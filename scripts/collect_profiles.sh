#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box (run through gpurun from the repository root):
#   bash scripts/collect_profiles.sh r05
# 1. kernel-trace + stats of the default bench command, 2.-4. PMC passes (FETCH_SIZE, WRITE_SIZE, MFMA busy) in runs of their own,
# 5. a clean kernel trace of plain steps (scripts/run_steps.py: no event brackets) for the step timeline / gaps, 6. the same counters and
# stats at local batch 4 (BASELINE configs[2] names a rocprof roofline capture), 7. bench lines at the other configurations.
# Raw output under gpurun_out/<tag>p/, reduced files are copied into profiles/ by hand.
set -o pipefail
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${TAG}p
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PMC_MFMA="SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
# (--no_also: the profiled process must hold the headline configuration's kernels only; the default line with its `also` block -- what the driver
# runs -- is taken without the profiler at the end)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $R/bench.py --steps 10 --warmup 3 --no_also > $OUT/bench_b8.json 2> $OUT/bench_b8.err || exit 1
echo "stats done"
for LB in 8 4 2; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch$LB -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_also --local_batch_size $LB > $OUT/fetch$LB.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write$LB -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_also --local_batch_size $LB > $OUT/write$LB.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc $PMC_MFMA --output-format csv -d $OUT/mfma$LB -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_also --local_batch_size $LB > $OUT/mfma$LB.log 2>&1 || exit 1
  echo "pmc passes at local batch $LB done"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats4 -o run -- python3 $R/bench.py --steps 10 --warmup 3 --local_batch_size 4 --no_cpu_baseline > $OUT/bench_b4.json 2> $OUT/bench_b4.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats2 -o run -- python3 $R/bench.py --steps 20 --warmup 5 --local_batch_size 2 --no_cpu_baseline > $OUT/bench_b2_prof.json 2> $OUT/bench_b2.err || exit 1
rocprofv3 --kernel-trace --output-format csv -d $OUT/plain -o run -- python3 $R/scripts/run_steps.py 8 16 > $OUT/plain.log 2>&1 || exit 1
DC_SIDE_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d $OUT/serial -o run -- python3 $R/scripts/run_steps.py 8 16 > $OUT/serial.log 2>&1 || exit 1
echo "traces done"
cd $R
first() { ls $1/*/*$2 $1/*$2 2>/dev/null | head -1; }
for LB in 8 4 2; do
  python3 scripts/pmc_traffic.py $OUT/fetch$LB $OUT/write$LB $OUT/pmc_traffic_b$LB.json $LB > $OUT/pmc_traffic_b$LB.txt 2>&1
  python3 scripts/pmc_mfma.py $OUT/mfma$LB $OUT/pmc_mfma_b$LB.json > $OUT/pmc_mfma_b$LB.txt 2>&1
  mkdir -p $OUT/rf$LB && ln -sfn $OUT/fetch$LB $OUT/rf$LB/FETCH_SIZE && ln -sfn $OUT/write$LB $OUT/rf$LB/WRITE_SIZE && ln -sfn $OUT/mfma$LB $OUT/rf$LB/SQ_INSTS_VALU_MFMA_MOPS_BF16
  python3 scripts/roofline_report.py $OUT/rf$LB $OUT/roofline_table_b$LB.md $LB > $OUT/roofline_table_b$LB.txt 2>&1
done
python3 scripts/step_timeline.py $(first $OUT/plain kernel_trace.csv) 12 > $OUT/step_timeline.txt 2>&1
python3 scripts/trace_gaps.py $(first $OUT/plain kernel_trace.csv) > $OUT/trace_gaps.txt 2>&1
python3 scripts/step_dump.py $(first $OUT/plain kernel_trace.csv) $OUT/step_both.tsv 12 > /dev/null 2>&1
python3 scripts/step_dump.py $(first $OUT/serial kernel_trace.csv) $OUT/step_serial.tsv 12 > /dev/null 2>&1
cp $(first $OUT/stats kernel_stats.csv) $OUT/kernel_stats_b8.csv
cp $(first $OUT/stats4 kernel_stats.csv) $OUT/kernel_stats_b4.csv
cp $(first $OUT/stats2 kernel_stats.csv) $OUT/kernel_stats_b2.csv
python3 scripts/serial_table.py $OUT/step_serial.tsv $OUT/step_both.tsv > $OUT/step_serial_table.txt 2>&1
python3 bench.py --steps 10 --warmup 3 --local_batch_size 2 --dtype fp32 --optimizer Adam --no_cpu_baseline > $OUT/bench_b2_fp32.json 2>> $OUT/bench_b8.err
python3 bench.py --steps 20 --warmup 5 --local_batch_size 4 --optimizer Adam --no_cpu_baseline > $OUT/bench_b4_adam.json 2>> $OUT/bench_b8.err
python3 bench.py --steps 20 --warmup 5 --local_batch_size 2 --no_cpu_baseline > $OUT/bench_b2.json 2>> $OUT/bench_b8.err
python3 bench.py > $OUT/bench_b8_plain.json 2>> $OUT/bench_b8.err          # the driver's command: headline + `also` + cpu_baseline
# the raw traces are large: keep the reduced files only
rm -rf $OUT/fetch8 $OUT/write8 $OUT/mfma8 $OUT/fetch4 $OUT/write4 $OUT/mfma4 $OUT/fetch2 $OUT/write2 $OUT/mfma2 $OUT/stats $OUT/stats4 $OUT/stats2 $OUT/rf8 $OUT/rf4 $OUT/rf2 $OUT/plain $OUT/serial
ls -la $OUT
