"""profiles/r06/pmc_kernels.txt (isolated attn1 passes) + pmc_step_kernels.json (in-situ passes over one eager CFG evaluation) -> profiles/r06/traffic.json,
the file bench.py reads for roofline.traffic.  bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE / WRITE_SIZE are KiB and the x2 is the gfx950 read-side
correction of MI355X_MICROARCH.md (HBM section).  Usage: python profiles/r06/make_traffic.py"""
import ast, json, os, re

here = os.path.dirname(os.path.abspath(__file__))
iso = {}
for line in open(os.path.join(here, "pmc_kernels.txt")):
    m = re.match(r"attn1 (\{.*?\}) dispatches (\d+) kernel (\S+)", line)
    if m:
        iso.update(ast.literal_eval(m.group(1))); iso["device_kernel"] = m.group(3)
step = json.load(open(os.path.join(here, "pmc_step_kernels.json")))


def row(c, alg=None, note=None, extra=None):
    wc = c.get("SQ_WAVE_CYCLES") or 0
    hit, mis = c.get("TCC_HIT_sum", 0), c.get("TCC_MISS_sum", 0)
    r = {"fetch_kib": c.get("FETCH_SIZE"), "write_kib": c.get("WRITE_SIZE"),
         "bytes": int((2 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024),
         "MfmaUtil_pct": c.get("MfmaUtil"), "VALUBusy_pct": c.get("VALUBusy"),
         "issue_frac_of_wave_cycles": round(c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3) if wc else None,
         "wait_inst_frac": round(c.get("SQ_WAIT_INST_ANY", 0) / wc, 3) if wc else None,
         "wait_any_frac": round(c.get("SQ_WAIT_ANY", 0) / wc, 3) if wc else None,
         "l2_hit_rate": round(hit / (hit + mis), 3) if hit + mis else None,
         "lds": {k: c[k] for k in ("SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_MFMA", "SQ_INSTS_VALU") if k in c}}
    if alg: r["algorithmic_bytes"] = alg
    if note: r["note"] = note
    if extra: r.update(extra)
    return r


out = {"_comment": "HBM traffic and utilisation per launch, round 6, FINAL-TREE kernels only (no carried-over row): rocprofv3 --pmc passes of profiles/pmc_r06.sh, one counter set per "
                   "run, never combined with trace domains.  The dominant kernel is measured twice: in isolation (profiles/kprobe.py attn1, as in rounds 4-5) and in situ; every other "
                   "row is IN SITU: one eager CFG evaluation of the headline workload (profiles/r06/one_forward.py), averaged per (device kernel, grid) over the launches of that "
                   "class in the step (pmc_step_kernels.json holds all 64 classes).  bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 read-side correction, MI355X_MICROARCH.md). "
                   "MfmaUtil / VALUBusy: rocprofv3 derived metrics (gfx94x formulas); issue / wait fractions over SQ_WAVE_CYCLES; l2_hit_rate = TCC_HIT / (TCC_HIT + TCC_MISS).",
       "_meta": {"round": 6, "collected": "2026-10-01", "kernel_build": "every kernel of the round-6 step (final tree)", "raw": "profiles/r06/pmc_kernels.txt, profiles/r06/pmc_step_kernels.json"}}
dom = "attn40p_kernel<bf16>self B2 H8 N16384 M16384 D40"
insitu_dom = next(v for k, v in step.items() if "attn40p" in k and "grid 1024" in k)
out[dom] = row(iso, alg=83886080, note="isolated launch (kprobe attn1); fetch is 2.9x the algorithmic bytes: each 256-query workgroup streams its head's K / V (0.36 TB/s: not a bound)",
               extra={"shape": "self-attention level 0: B2 H8 N16384 M16384 D40", "device_kernel": iso.get("device_kernel"),
                      "in_situ": row(insitu_dom), "launches_per_step": insitu_dom["dispatches_per_forward"]})
names = [("attn40p", "grid 512 x", "attn40p_kernel<bf16>self B1 H8 N16384 M16384 D40 (the shared CFG prefix's self-attention: one image)"),
         ("gemm_pp_kernel<bool _Accum, int, E, 160", "grid 256 x", "3x3 convs of the 128^2 level, M32768 N320 K2880..8640 (gemm_pp_kernel<bf16,1,160>, 256 x 160 tiles)"),
         ("gemm_pp_kernel<bool _Accum, int, E, 128", "grid 240 x", "3x3 convs of the 64^2 level, M8192 N640, split-K 2 (gemm_pp_kernel<bf16,1,128>)"),
         ("ff_block", "grid 256 x", "ff_block<bf16> M32768 C320 (LayerNorm + GEGLU projection + down projection + residual)"),
         ("attn32g_kernel<bool _Accum, int, ELi3E", "grid 512 x", "self-attention of the 64^2 level: B2 H8 N4096 D80 (attn32g_kernel)"),
         ("splitk_reduce_gn", "grid 512 x", "split-K reduce + GroupNorm statistics (splitk_reduce_gn_kernel; 21 per step)"),
         ("gemm_pp_kernelIDF16bLi0ELi256ELb0", "grid 640 x", "GEGLU up-projection of the 64^2 level, M8192 N5120 K640 on 256 x 256 tiles (round 6)"),
         ("gemm_kernel<bool _Accum, int, E, 128, 128, 2", "grid 440 x", "weight-streaming 3x3 convs of the 16^2 level, M512 N1280 K11520..23040, split-K 11 (gemm_kernel<bf16,1,128,128>)"),
         ("gemm_ring", "grid 256 x", "N = K = 1280 projections of the 32^2 level, M2048 (gemm_ring_kernel, 64 x 160 tiles; 13.7 MB algorithmic)"),
         ("rowgemm_kernelIDF16bLi0ELi640", "grid 256 x", "rowgemm<bf16,0> C = 640 (to_out / proj_out + residual at the 64^2 level)"),
         ("rowgemm_kernelIDF16bLi0ELi320", "grid 256 x", "rowgemm<bf16,0> C = 320, M32768 (to_out / proj_out + residual at the 128^2 level)"),
         ("rowgemm_kernel<bool _Accum, int, E, 320", "grid 256 x", "rowgemm<bf16,1> C = 320, M32768, N 960 (LayerNorm + q|k|v)"),
         ("xattn_block", "grid 256 x", "xattn_block<bf16> M32768 C320, 77 keys"),
         ("gn_apply_kernel", "grid 1024 x", "gn_apply (GroupNorm apply + SiLU), largest class"),
         ("attn32g_kernel<bool _Accum, int, ELi6E", "grid 128 x", "self-attention of the 32^2 level: B2 H8 N1024 D160 (attn32g_kernel)")]
for sub, grid, label in names:
    hit = [(k, v) for k, v in step.items() if sub in k and grid in k]
    if not hit:
        continue
    k, v = hit[0]
    out[label] = row(v, extra={"device_kernel_and_grid": k, "launches_per_step": v["dispatches_per_forward"], "in_situ": True})
json.dump(out, open(os.path.join(here, "traffic.json"), "w"), indent=1)
print("traffic.json:", len(out) - 2, "kernel rows; dominant", dom, "bytes", out[dom]["bytes"])
