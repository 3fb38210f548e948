"""Per-level time table of one SD1.5 1024^2 CFG evaluation from per-shape probes (profiles/shape_probe.py output files).
Level = where the op runs: 0 = 128^2 (M 32768 / 16384 rows, HW 16384), 1 = 64^2 (M 8192, HW 4096), 2 = 32^2 (M 2048, HW 1024), 3 + mid = 16^2 (M 512, HW 256).
Usage: python profiles/r06/level_table.py <before.txt> <after.txt>"""
import re, sys


def level(key):
    m = re.search(r"HW(\d+)", key)
    if m: hw = int(m.group(1))
    elif re.search(r"N(\d+) M\d+ D\d+", key): hw = int(re.search(r"N(\d+) M\d+ D\d+", key).group(1))      # attention: "B2 H8 N16384 M16384 D40"
    elif re.search(r" R(\d+) C", key): hw = int(re.search(r" R(\d+) C", key).group(1)) // 2                  # LayerNorm rows (CFG batch 2; a prefix op has B = 1)
    elif re.search(r" M(\d+)", key):
        rows = int(re.search(r" M(\d+)", key).group(1))
        if rows in (154, 77): return "ctx"
        hw = 16384 if rows == 16384 else rows // 2
    elif key.startswith(("rowgemm", "xattn_block", "ff_block")): return "0-1 row-block (no shape in this probe)"
    else: return "boundary"
    return {16384: "0 (128^2)", 4096: "1 (64^2)", 1024: "2 (32^2)", 256: "3+mid (16^2)"}.get(hw, "other")


def kind(key):
    if key.startswith("attn"): return "attention"
    if key.startswith(("gn_", "ln_")): return "norm"
    if "<bf16,1>" in key and key.startswith("gemm"): return "conv3x3"
    return "linear / row-block"


def load(path):
    t = {}
    for line in open(path):
        m = re.match(r"\s*([\d.]+) ms\s+[\d.]+%\s+n=\s*(\d+)\s+[\d.]+ us/op(.*)$", line.rstrip())
        if not m: continue
        ms = float(m.group(1))
        key = re.sub(r"^\s*(\d+ TF)?\s*(\d+ GB/s)?\s*", "", m.group(3))
        lv = level(key)
        e = t.setdefault(lv, {}); e[kind(key)] = e.get(kind(key), 0.0) + ms
    return t


a, b = load(sys.argv[1]), load(sys.argv[2])
kinds = ["attention", "conv3x3", "linear / row-block", "norm"]
print(f"per-level ms of one CFG evaluation (sum of per-op HIP-event times; BEFORE = {sys.argv[1]}, AFTER = {sys.argv[2]})")
print(f"{'level':14s} " + " ".join(f"{k:>26s}" for k in kinds) + f" {'total':>18s}")
ta = tb = 0.0
for lv in ["0 (128^2)", "1 (64^2)", "2 (32^2)", "3+mid (16^2)", "0-1 row-block (no shape in this probe)", "ctx", "boundary", "other"]:
    if lv not in a and lv not in b: continue
    row = []
    sa = sb = 0.0
    for k in kinds:
        x, y = a.get(lv, {}).get(k, 0.0), b.get(lv, {}).get(k, 0.0)
        sa += x; sb += y
        row.append(f"{x:10.3f} -> {y:8.3f}" + " " * 4)
    ta += sa; tb += sb
    print(f"{lv[:14]:14s} " + " ".join(f"{r:>26s}" for r in row) + f" {sa:7.3f} -> {sb:7.3f}")
# the start-of-round probe carried no shape for the row-block kernels (levels 0 and 1): levels 0 + 1 + that row, comparable in both files
c01 = lambda t: sum(sum(t.get(lv, {}).values()) for lv in ("0 (128^2)", "1 (64^2)", "0-1 row-block (no shape in this probe)"))
print(f"{'0 + 1 incl. row-block kernels':70s}" + " " * 52 + f" {c01(a):7.3f} -> {c01(b):7.3f}")
print(f"{'sum':14s} " + " " * (27 * len(kinds)) + f" {ta:7.3f} -> {tb:7.3f}")
