"""Resource usage of every device kernel in the shipped libldx.so, read from the code objects' own metadata (no recompilation):
`llvm-objdump --offloading` unbundles the gfx950 code objects, `llvm-readelf --notes` prints each kernel's amdhsa metadata.

Round 4's accident: an opt-in experiment compiled into every GEMM instantiation cost 16-73 VGPRs, spilled the 128 x 160 tiles and took
a wave of occupancy from the 64 x 64 ones; nobody noticed until a 3 ms drift in the VAE.  tests/test_kernel_resources_cpu.py compares this table
with the committed one (tests/golden/kernel_resources.json) on every CPU run.

    python tests/tools/kernel_resources.py            # print the table of the current build
    python tests/tools/kernel_resources.py --write    # accept it as the new baseline (after LOOKING at the diff)
"""
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB = os.path.join(ROOT, "lightdiffusion-next_amd", "libldx.so")
TABLE = os.path.join(ROOT, "tests", "golden", "kernel_resources.json")
LLVM = "/opt/rocm/lib/llvm/bin"
FIELDS = {"vgpr_count": "vgpr", "agpr_count": "agpr", "sgpr_count": "sgpr", "private_segment_fixed_size": "scratch",
          "group_segment_fixed_size": "lds_static", "vgpr_spill_count": "vgpr_spill", "sgpr_spill_count": "sgpr_spill",
          "max_flat_workgroup_size": "threads"}


def waves_per_simd(vgpr: int, agpr: int) -> int:
    """MI355X_MICROARCH.md "Register files": unified 512-entry file, allocation granule 8, waves = min(8, 512 // alloc).
    The unified budget is vgpr (rounded up to 8: accum_offset is 4-aligned, the granule is 8) + agpr."""
    # .vgpr_count of the kernel metadata is the TOTAL of the unified file (arch VGPRs + accumulator registers; attn40p: 443 = 256 + 187)
    alloc = (vgpr + 7) // 8 * 8
    return max(1, min(8, 512 // max(alloc, 8)))


def collect(lib_path: str = LIB) -> dict:
    tmp = tempfile.mkdtemp(prefix="ldx_res_")
    try:
        so = os.path.join(tmp, "libldx.so")
        shutil.copy(lib_path, so)                   # llvm-objdump writes the unbundled objects NEXT to its input
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], check=True, capture_output=True, cwd=tmp)
        table = {}
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
            cur = None

            def close(rec):
                # one kernel record = the keys between two "  - .<first key>:" lines of amdhsa.kernels (keys are sorted: .vgpr_count comes AFTER .symbol)
                if rec and "name" in rec and "vgpr" in rec:
                    name = rec.pop("name")
                    rec["waves_per_simd"] = waves_per_simd(rec.get("vgpr", 0), rec.get("agpr", 0))
                    table[name] = rec

            in_kernels = False
            for line in notes.splitlines():
                if "amdhsa.kernels:" in line:
                    in_kernels = True
                    continue
                if not in_kernels:
                    continue
                if re.match(r"\s*amdhsa\.\w+:", line):          # next top-level section (amdhsa.target, amdhsa.version)
                    close(cur); cur = None; in_kernels = False
                    continue
                m0 = re.match(r"^  - \.(\w+):", line)             # a new kernel record (two-space indent: argument records are indented deeper)
                if m0:
                    close(cur); cur = {}
                m = re.match(r"\s*(?:- )?\.(\w+):\s+(\S+)\s*$", line)
                if not m or cur is None:
                    continue
                key, val = m.group(1), m.group(2)
                if key == "name" and re.match(r"^    \.name:", line):     # the kernel's own .name (4-space indent), not an argument's
                    cur["name"] = val
                elif key in FIELDS and re.match(r"^(  - |    )\.", line):
                    try:
                        cur[FIELDS[key]] = int(val)
                    except ValueError:
                        pass
            close(cur)
        return table
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def demangle(names):
    try:
        out = subprocess.run([os.path.join(LLVM, "llvm-cxxfilt")], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def diff(old: dict, new: dict):
    """[(kernel, what)] for everything that must not drift silently: scratch, spills, occupancy class, kernels appearing / disappearing."""
    out = []
    for k in sorted(set(old) | set(new)):
        if k not in new:
            out.append((k, "kernel gone from the build")); continue
        if k not in old:
            out.append((k, f"new kernel {new[k]}")); continue
        o, n = old[k], new[k]
        for f in ("scratch", "vgpr_spill", "sgpr_spill", "waves_per_simd"):
            if o.get(f, 0) != n.get(f, 0):
                out.append((k, f"{f} {o.get(f, 0)} -> {n.get(f, 0)}  (vgpr {o.get('vgpr')} -> {n.get('vgpr')}, agpr {o.get('agpr')} -> {n.get('agpr')})"))
    return out


if __name__ == "__main__":
    t = collect()
    if "--write" in sys.argv:
        with open(TABLE, "w") as f:
            json.dump(t, f, indent=0, sort_keys=True)
        print(f"wrote {len(t)} kernels to {TABLE}")
    else:
        dm = demangle(list(t))
        for k, v in sorted(t.items(), key=lambda kv: dm[kv[0]]):
            print(f"{v['vgpr']:4d}v {v.get('agpr', 0):4d}a {v['sgpr']:4d}s scratch {v['scratch']:5d} spill {v['vgpr_spill']:3d}/{v['sgpr_spill']:3d} occ {v['waves_per_simd']}  {dm[k][:150]}")
        print(len(t), "kernels;", sum(1 for v in t.values() if v["scratch"]), "with scratch")
