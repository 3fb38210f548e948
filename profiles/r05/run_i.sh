TAG=r05i KERNEL=attn_mx_kernel PROBE="python $GRAFT_REPO_ROOT/profiles/attn_mx_probe.py" bash profiles/pmc_attn512.sh
