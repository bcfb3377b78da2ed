#!/bin/bash
# bf16 and fp32 time-to-target runs of the driver from the same seed on learnable synthetic data:
#   bash scripts/convergence_pair.sh gpurun_out/conv   -> <dir>/bf16.log, <dir>/fp32.log (:::MLLOG), then scripts/mllog_curves.py
set -e -o pipefail
out=${1:-gpurun_out/conv}
mkdir -p "$out"
common="--wireup_method single --synthetic_samples 48 --synthetic_learnable --local_batch_size 4 --optimizer LAMB --start_lr 2e-3
 --weight_decay 1e-2 --validation_frequency 24 --target_iou 0.82 --logging_frequency 1 --save_frequency 0 --max_epochs 40 --max_steps 400
 --training_visualization_frequency 0 --validation_visualization_frequency 0"
python -m mlperf_deepcam_amd.train $common --amp_opt_level O0 --run_tag fp32 --output_dir "$out/run_fp32" > "$out/fp32.stdout" 2>&1
cp "$out/run_fp32/logs/fp32.log" "$out/fp32.log"
python -m mlperf_deepcam_amd.train $common --amp_opt_level O1 --run_tag bf16 --output_dir "$out/run_bf16" > "$out/bf16.stdout" 2>&1
cp "$out/run_bf16/logs/bf16.log" "$out/bf16.log"
python scripts/mllog_curves.py "$out/fp32.log" "$out/bf16.log" --every 24 --csv "$out/curves.csv" | tee "$out/curves.txt"
