#!/bin/bash
# One steady-state step's kernels alone (DC_SIDE_STREAM=0) and in the two-stream step, at local batch $1 (default 8): kernel traces of
# scripts/run_steps.py reduced by step_dump.py + serial_table.py into gpurun_out/$2/serial_table_b$1.txt.   bash scripts/serial_pair.sh 8 r5h
B=${1:-8}; TAG=${2:-r5h}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/plain$B -o run -- python3 $R/scripts/run_steps.py $B 16 > $OUT/plain$B.log 2>&1 || exit 1
DC_SIDE_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d $OUT/serial$B -o run -- python3 $R/scripts/run_steps.py $B 16 > $OUT/serial$B.log 2>&1 || exit 1
cd $R
first() { ls $1/*/*$2 $1/*$2 2>/dev/null | head -1; }
python3 scripts/step_dump.py $(first $OUT/plain$B kernel_trace.csv) $OUT/step_both_b$B.tsv 12 > /dev/null 2>&1
python3 scripts/step_dump.py $(first $OUT/serial$B kernel_trace.csv) $OUT/step_serial_b$B.tsv 12 > /dev/null 2>&1
python3 scripts/serial_table.py $OUT/step_serial_b$B.tsv $OUT/step_both_b$B.tsv > $OUT/serial_table_b$B.txt 2>&1
rm -rf $OUT/plain$B $OUT/serial$B
