#!/bin/bash
# Snapshot of a source tree's package (Python + built library) under ab/<name>/ for same-box A/B runs of whole trees:
#   bash scripts/snapshot_tree.sh r3 /tmp/base        # a git worktree of the other revision, library built
#   python scripts/ab_step.py "" "AB_ROOT=ab/r3"
# ab/ is git-ignored but travels to the GPU box with gpurun.
set -e
NAME=$1; SRC=${2:-.}
DST=$(dirname "$0")/../ab/$NAME
rm -rf "$DST"; mkdir -p "$DST/mlperf-deepcam_amd" "$DST/tests"
cp "$SRC"/mlperf-deepcam_amd/*.py "$SRC"/mlperf-deepcam_amd/libdeepcam_hip.so "$DST/mlperf-deepcam_amd/"
cp -r "$SRC"/mlperf_deepcam_amd "$DST/"
cp "$SRC"/tests/util_inputs.py "$DST/tests/"
echo "snapshot $NAME: $(du -sh "$DST" | cut -f1)"
