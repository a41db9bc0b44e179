"""Build-time guard (no GPU needed: hipcc cross-compiles): register use / occupancy of the hot conv instantiations.

An epilogue feature compiled into every instantiation once raised all of them by ~40 VGPRs and silently took one wave
per SIMD away (-8 % on the benchmark); occupancy is what lets several workgroups share a CU and hide each other's
load / multiply / store phases, so it is pinned here (and, since round 2, stated to the compiler through __launch_bounds__).

The same compile's assembly is scanned for the two hazards the hardware does not interlock and hipcc cannot see inside
inline asm (tools/scan_sgpr_hazard.py): a descriptor SGPR reloaded by v_readlane fewer than 5 wait states before an
inline buffer instruction, and the data registers of a 16-byte inline store overwritten in the next instruction."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

# (BM, BN, WM, WN, stages, addend) -> minimum waves per SIMD
EXPECTED = {
    (128, 128, 2, 2, 1, 0): 4,   # large grids of 128x128 tiles: 4 workgroups per CU (37 KB of LDS each)
    (128, 128, 2, 2, 1, 1): 3,   # ... with the residual-gradient addend
    (128, 128, 2, 2, 4, 0): 2,
    (128, 64, 4, 1, 1, 0): 5,
    (128, 64, 4, 1, 4, 0): 4,
    (256, 256, 4, 2, 2, 0): 2,   # 8 waves per workgroup = 2 per SIMD: must not drop to 1
    (256, 256, 4, 2, 2, 1): 2,
}


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_conv_kernel_occupancy():
    src = os.path.join(ROOT, "sihl_amd", "csrc", "conv_igemm_bf16.hip")  # the bf16 instantiations of conv_igemm_impl.h
    asm = os.path.join(tempfile.mkdtemp(), "conv_igemm.s")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", src, "-o", asm,
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from scan_sgpr_hazard import scan
    total, hits = scan(open(asm).read())
    os.remove(asm)
    assert total > 100 and not hits, hits[:5]
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


def test_hazard_scanner_on_synthetic_isa():
    """The scanner itself, on hand-written assembly: it must flag a descriptor reloaded by v_readlane right before an
    inline buffer instruction and a 16-byte store whose data register is overwritten next, and accept the padded forms."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from scan_sgpr_hazard import scan

    bad_sgpr = """_Zkernel_a:
\tv_readlane_b32 s6, v186, 2
\tv_readlane_b32 s7, v186, 3
\t;;#ASMSTART
\tbuffer_store_dword v0, v1, s[4:7], 0 offen
\t;;#ASMEND
"""
    ok_sgpr = bad_sgpr.replace("\tbuffer_store_dword", "\ts_nop 4\n\tbuffer_store_dword")
    bad_data = """_Zkernel_b:
\t;;#ASMSTART
\ts_nop 4
\tbuffer_store_dwordx4 v[0:3], v32, s[80:83], 0 offen
\t;;#ASMEND
\tv_add_u32_e32 v0, s1, v32
"""
    ok_data = bad_data.replace("0 offen\n", "0 offen\n\ts_nop 1\n", 1)
    dma = """_Zkernel_c:
\tv_readlane_b32 s11, v248, 14
\t;;#ASMSTART
\ts_mov_b32 m0, s40
\ts_nop 0
\tbuffer_load_dwordx4 v2, s[8:11], 0 offen lds
\t;;#ASMEND
"""
    assert len(scan(bad_sgpr)[1]) == 1 and not scan(ok_sgpr)[1]
    assert len(scan(bad_data)[1]) == 1 and not scan(ok_data)[1]
    assert len(scan(dma)[1]) == 1 and not scan(dma.replace("s_nop 0", "s_nop 3"))[1]
