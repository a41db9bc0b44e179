"""Static check of the gfx950 ISA of the inline-asm buffer instructions (LDS-DMA loads, conv_pw's stores): "VALU writes
SGPR -> VMEM reads that SGPR" needs 5 wait states, and the compiler's hazard recogniser does not see inside inline asm.
The kernels spill scalars to VGPR lanes, so a v_readlane / v_readfirstlane into a descriptor register can sit right in
front of the statement.  Usage:  python tools/scan_sgpr_hazard.py [file.hip ...]   (default: every translation unit that issues inline-asm buffer instructions)
Compiles each file to assembly (device only) and reports every buffer instruction whose descriptor registers were
written by a VALU lane read fewer than 5 wait states earlier, and every 12/16-byte store whose data registers the next
instruction overwrites ("VMEM store > 8 bytes -> VALU write of its data": 1 wait state).  Exit code 1 if any is found."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sihl_amd", "csrc")


def scan(asm_text):
    lines = asm_text.split("\n")
    kern, hits, total = None, [], 0
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\S+):", l)
        if m:
            kern = m.group(1)
        if not re.search(r"\bbuffer_(load|store)_dword", l):
            continue
        rs = re.search(r"s\[(\d+):(\d+)\]", l)
        if not rs:
            continue
        total += 1
        lo, hi = int(rs.group(1)), int(rs.group(2))
        st = re.search(r"buffer_store_dwordx[34]\s+v\[(\d+):(\d+)\]", l)
        if st:  # store of > 8 bytes: no VALU write of its data registers in the next wait state
            dlo, dhi = int(st.group(1)), int(st.group(2))
            k = i + 1
            while k < len(lines) and (not lines[k].strip() or lines[k].strip().startswith(";")):
                k += 1
            t = lines[k].strip()
            mm = re.match(r"v_\w+\s+v(\d+)", t) or re.match(r"v_\w+\s+v\[(\d+):", t)
            if mm and dlo <= int(mm.group(1)) <= dhi and not t.startswith("s_nop"):
                hits.append((kern, i + 1, "store data overwritten", t))
        k, states = i - 1, 0
        while k > 0 and states < 5:
            t = lines[k].strip()
            k -= 1
            if not t or t.startswith(";") or t.startswith("."):
                continue
            if t.startswith("s_nop"):
                states += int(t.split()[1]) + 1
                continue
            mm = re.match(r"v_read(?:first)?lane_b32 s(\d+),", t)
            if mm and lo <= int(mm.group(1)) <= hi:
                hits.append((kern, i + 1, states, t))
                break
            states += 1
    return total, hits


def main():
    files = sys.argv[1:] or [os.path.join(CSRC, f) for f in ("conv_igemm_bf16.hip", "conv_igemm_f32.hip", "conv_wgrad.hip",
                                                             "mlp_fused.hip", "mlp_rows.hip")]  # every user of dma.h / inline asm
    bad = 0
    for f in files:
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "k.s")
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "--cuda-device-only",
                            "-S", f, "-o", out], check=True, stderr=subprocess.DEVNULL)
            total, hits = scan(open(out).read())
        print(f"{os.path.basename(f)}: {total} inline buffer instructions, {len(hits)} inside the 5-wait-state window")
        for h in hits[:10]:
            print("   ", h)
        bad += len(hits)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
