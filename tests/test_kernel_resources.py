"""Build-time guard (no GPU needed: hipcc cross-compiles): register use / occupancy of the hot conv instantiations.

An epilogue feature compiled into every instantiation once raised all of them by ~40 VGPRs and silently took one wave
per SIMD away (-8 % on the benchmark); occupancy is what lets several workgroups share a CU and hide each other's
load / multiply / store phases, so it is pinned here."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

# (BM, BN, WM, WN, stages, addend) -> minimum waves per SIMD
EXPECTED = {
    (128, 128, 2, 2, 1, 0): 3,   # large grids of 128x128 tiles: 3 workgroups per CU
    (128, 128, 2, 2, 1, 1): 3,   # ... with the residual-gradient addend
    (128, 128, 2, 2, 4, 0): 2,
    (128, 64, 4, 1, 1, 0): 5,
    (128, 64, 4, 1, 4, 0): 4,
    (256, 256, 4, 2, 2, 0): 2,   # 8 waves per workgroup = 2 per SIMD: must not drop to 1
    (256, 256, 4, 2, 2, 1): 2,
}


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_conv_kernel_occupancy():
    src = os.path.join(ROOT, "sihl_amd", "csrc", "conv_igemm.hip")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", src, "-o", os.devnull,
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    found, name = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"Occupancy \[waves/SIMD\]: (\d+)", line)
        if m and name and "conv_igemm_dma_kernelI6bf16_t" in name:
            k = re.search(r"Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb0ELi(\d+)ELb(\d)E", name)
            if k:
                found[tuple(int(v) for v in k.groups())] = int(m.group(1))
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name and "conv_igemm_dma_kernelI6bf16_t" in name:
            assert int(m.group(1)) == 0, f"{name} spills to scratch"
    for key, want in EXPECTED.items():
        assert key in found, (key, sorted(found))
        assert found[key] >= want, f"conv_igemm_dma_kernel{key}: {found[key]} waves/SIMD, expected >= {want}"
