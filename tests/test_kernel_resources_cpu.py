"""Guard rail: the resource usage of EVERY kernel instantiation in the shipped libldx.so against a committed table.

Read from the code objects' own amdhsa metadata (tests/tools/kernel_resources.py: llvm-objdump --offloading + llvm-readelf --notes; no GPU, no
recompilation, ~2 s).  Fails when a kernel gains scratch or spills, changes its occupancy class (waves per SIMD from the unified 512-entry
register file, 8-register granule), appears or disappears.  Round 4's split-K fix-up experiment cost every GEMM instantiation 16-73 VGPRs
and was found by luck; with this test the same change fails the CPU suite and names the kernels.

An INTENDED change is accepted with `python tests/tools/kernel_resources.py --write` after reading the diff this test prints."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import kernel_resources as KR  # noqa: E402


def test_no_kernel_drifted_in_scratch_spills_or_occupancy(ldx_lib):
    new = KR.collect()
    with open(KR.TABLE) as f:
        old = json.load(f)
    assert len(new) > 200, len(new)                       # the extraction itself must have worked
    d = KR.diff(old, new)
    msg = "\n".join(f"  {k}: {what}" for k, what in d)
    assert not d, f"kernel resource usage drifted against tests/golden/kernel_resources.json:\n{msg}\n(accept with: python tests/tools/kernel_resources.py --write)"


def test_no_kernel_uses_scratch(ldx_lib):
    """Stronger than 'unchanged': the shipped build has NO kernel with a private segment or a VGPR spill (SGPR spills go to VGPR lanes and are
    tolerated where the table records them: outside the K loops, profiles/ubench/README.md)."""
    new = KR.collect()
    bad = {k: v for k, v in new.items() if v["scratch"] or v["vgpr_spill"]}
    assert not bad, bad


def test_occupancy_model_matches_the_guide():
    # MI355X_MICROARCH.md "Register files" table: allocated VGPR+AGPR per lane -> waves/SIMD
    for alloc, waves in ((64, 8), (72, 7), (80, 6), (96, 5), (128, 4), (168, 3), (256, 2), (264, 1), (512, 1)):
        assert KR.waves_per_simd(alloc, 0) == waves, alloc
    assert KR.waves_per_simd(443, 187) == 1 and KR.waves_per_simd(128, 8) == 4 and KR.waves_per_simd(129, 8) == 3      # .vgpr_count already includes the accumulator registers
