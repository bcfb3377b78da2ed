#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box (run through gpurun from the repository root):
#   bash scripts/collect_profiles.sh r02
# 1. kernel-trace + stats of the default bench command, 2.-4. PMC passes (FETCH_SIZE, WRITE_SIZE, MFMA busy) in runs of their own,
# 5. bench lines at local batch 4 and 2.  Raw output under gpurun_out/<tag>p/, reduced files are copied into profiles/ by hand.
set -o pipefail
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${TAG}p
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $R/bench.py --steps 10 --warmup 3 > $OUT/bench_b8.json 2> $OUT/bench_b8.err || exit 1
echo "stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline > $OUT/fetch.log 2>&1 || exit 1
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline > $OUT/write.log 2>&1 || exit 1
echo "write done"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline > $OUT/mfma.log 2>&1 || exit 1
echo "mfma done"
cd $R
python3 scripts/pmc_traffic.py $OUT/fetch $OUT/write $OUT/pmc_traffic.json > $OUT/pmc_traffic.txt 2>&1
python3 scripts/pmc_mfma.py $OUT/mfma $OUT/pmc_mfma.json > $OUT/pmc_mfma.txt 2>&1
mkdir -p $OUT/rf && ln -sfn $OUT/fetch $OUT/rf/FETCH_SIZE && ln -sfn $OUT/write $OUT/rf/WRITE_SIZE && ln -sfn $OUT/mfma $OUT/rf/SQ_INSTS_VALU_MFMA_MOPS_BF16
python3 scripts/roofline_report.py $OUT/rf $OUT/roofline_table.md > $OUT/roofline_table.txt 2>&1
python3 scripts/step_timeline.py $(ls $OUT/stats/*/*kernel_trace.csv $OUT/stats/*kernel_trace.csv 2>/dev/null | head -1) > $OUT/step_timeline.txt 2>&1
python3 scripts/trace_gaps.py $(ls $OUT/stats/*/*kernel_trace.csv $OUT/stats/*kernel_trace.csv 2>/dev/null | head -1) > $OUT/trace_gaps.txt 2>&1
cp $(ls $OUT/stats/*/*kernel_stats.csv $OUT/stats/*kernel_stats.csv 2>/dev/null | head -1) $OUT/kernel_stats.csv
python3 bench.py --steps 20 --warmup 5 --local_batch_size 4 --no_cpu_baseline > $OUT/bench_b4.json 2>> $OUT/bench_b8.err
python3 bench.py --steps 20 --warmup 5 --local_batch_size 2 --no_cpu_baseline > $OUT/bench_b2.json 2>> $OUT/bench_b8.err
# the raw traces are large: keep the reduced files only
rm -rf $OUT/fetch $OUT/write $OUT/mfma $OUT/stats $OUT/rf
ls -la $OUT
