"""CPU: every gfx950 kernel of the library compiles without register spills or private-segment (scratch) use.
hipcc cross-compiles without a GPU; the code-object metadata of the device-only assembly is the evidence
(`.vgpr_spill_count`, `.private_segment_fixed_size`).  A spill in a hot loop is a silent 2x: round 1's weight-gradient
engine carried 14 spilled VGPRs under its occupancy cap."""
import os
import re
import shutil
import subprocess

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "literalkg_amd", "csrc")
HIP_SOURCES = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


@pytest.mark.parametrize("src", HIP_SOURCES)
def test_no_kernel_spills_or_uses_scratch(src):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    asm = subprocess.run([hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "--cuda-device-only", "-S", "-o", "-",
                          os.path.join(CSRC, src)], check=True, capture_output=True, text=True).stdout
    kernels = re.findall(r"\.name:\s+(\S+)", asm)
    spills = [int(x) for x in re.findall(r"\.vgpr_spill_count:\s+(\d+)", asm)]
    scratch = [int(x) for x in re.findall(r"\.private_segment_fixed_size:\s+(\d+)", asm)]
    assert kernels and len(kernels) == len(spills) == len(scratch), (len(kernels), len(spills), len(scratch))
    bad = [(k, s, p) for k, s, p in zip(kernels, spills, scratch) if s or p]
    assert not bad, bad


def test_tall_gemm_keeps_two_workgroups_per_cu():
    """The one-accumulator 256-column tall kernels (plain AND gate epilogue) are built for two workgroups per CU: the 8-wave
    form (64 x 64 wave tiles) at most 128 VGPRs (4 waves per SIMD), the 4-wave form (64 x 128 wave tiles) at most 256
    (2 waves per SIMD) -- with one workgroup per CU the k loop of the gate left the matrix pipe 38 % busy.
    Their LDS (76-79 KB each) is dynamic and set by the launcher."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    asm = subprocess.run([hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "--cuda-device-only", "-S", "-o", "-",
                          os.path.join(CSRC, "lkg_gemm_tall.hip")], check=True, capture_output=True, text=True).stdout
    found = {}
    for name, vgpr in re.findall(r"\.name:\s+(\S*gemm_tall_kernelILi256ELi[01]ELb1ELi(?:64|128)ELi(?:128|256)E\S*)\n(?:.*\n)*?.*?\.vgpr_count:\s+(\d+)", asm):
        found[name] = int(vgpr)
    narrow = {k: v for k, v in found.items() if "ELb1ELi64ELi128E" in k}
    wide = {k: v for k, v in found.items() if "ELb1ELi128ELi" in k}      # 64 x 128 wave tiles: 128- and 256-row tiles
    assert len(narrow) == 2 and len(wide) == 4, found
    assert all(v <= 128 for v in narrow.values()), narrow
    assert all(v <= 256 for v in wide.values()), wide
