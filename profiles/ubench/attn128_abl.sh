#!/bin/bash
# Timing ablations of attn128p_kernel (wrong results by construction): profiles/ubench/libldx_abl.so is libldx built with -DLDX_ATTN_ABLATE.
# Bits: 1 no s_barrier, 2 no maximum, 4 no MFMAs, 8 no staging, 16 no fragment reads, 32 no softmax half-pieces.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT/profiles/ubench
mkdir -p abl_lib && cp libldx_abl.so abl_lib/libldx.so
for abl in 0 1 2 4 8 16 32 34 24 58 62; do
  echo "== ABL $abl"; LD_LIBRARY_PATH=$ROOT/profiles/ubench/abl_lib LDX_ATTN_PIPE_ABL=$abl timeout 100 ./attn_pipe_test 4352 20 1 128 2>&1 | grep timing | tail -1
done
