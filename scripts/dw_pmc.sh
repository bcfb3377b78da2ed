#!/bin/bash
# Counter passes over scripts/dw_pmc.py (one rocprofv3 run per set: the TCC block holds 4 counters, SQ 8).   bash scripts/dw_pmc.sh <out dir>
set +e
OUT=${1:-gpurun_out/dw_pmc}
mkdir -p $OUT
export TMPDIR=/tmp
pass() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o run --output-format csv -- python3 scripts/dw_pmc.py > $OUT/$name.log 2>&1; echo "pass $name done"; }
pass sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES
pass tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
pass tcc2 TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_LEVEL_sum
pass tcc3 TCC_TAG_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_BUSY_sum TCC_EA0_WRREQ_LEVEL_sum
# (a TA_* pass aborted inside the profiler on this image and the job sat silent until it was killed: SQ and TCC only)
python3 scripts/dw_pmc.py reduce $OUT $OUT/table.txt > /dev/null
