#!/bin/bash
# Kernel traces of the step with the BatchNorm finalize in the dependent chain (default) and off it (DC_DEBUG_SKIP_BN_FINALIZE=async: the
# same kernels, on a stream of their own, the chain reading the previous step's coefficients): which kernels get shorter.
#   bash scripts/fin_trace.sh <out dir>      (each run: rocprofv3 --kernel-trace over scripts/run_steps.py 8 6)
OUT=${1:-gpurun_out/fin_trace}
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/default -o run -- python3 scripts/run_steps.py 8 6 > $OUT/default.json 2> $OUT/default.err
echo "default done"
export DC_DEBUG_SKIP_BN_FINALIZE=async
rocprofv3 --kernel-trace --output-format csv -d $OUT/async -o run -- python3 scripts/run_steps.py 8 6 > $OUT/async.json 2> $OUT/async.err
echo "async done"
