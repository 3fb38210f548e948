#!/bin/bash
# Timing ablations of attn40p_kernel (wrong results by construction): profiles/ubench/libldx_abl.so is libldx built with -DLDX_ATTN_ABLATE.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT/profiles/ubench
mkdir -p abl_lib && cp libldx_abl.so abl_lib/libldx.so
for abl in 0 1 2 8 16 32 9 25 27 59; do
  echo "== ABL $abl"; LD_LIBRARY_PATH=$ROOT/profiles/ubench/abl_lib LDX_ATTN_PIPE_ABL=$abl ./attn_pipe_test 16384 10 1 2>&1 | grep timing | tail -1
done
