#!/bin/bash
# Build container helper: export <git-rev> into ./_ab (git-ignored, travels with gpurun) and build its libldx.so, so that
# `cd _ab && python bench.py ...` and `python bench.py ...` can be compared on the SAME GPU box in one gpurun call
# (box-to-box spread is +-3 %, larger than most kernel-level gains).
set -e
REV=${1:-HEAD~1}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
rm -rf "$ROOT/_ab" && mkdir -p "$ROOT/_ab"
git -C "$ROOT" archive "$REV" | tar -x -C "$ROOT/_ab"
make -C "$ROOT/_ab/lightdiffusion-next_amd/csrc" -j8 > /dev/null
echo "built $REV in _ab/"
