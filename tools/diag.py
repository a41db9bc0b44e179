"""Developer diagnostic (GPU box): per-tensor error of the HIP path vs the golden vectors (fp32) and vs
the fp32 oracle evaluated at bf16-rounded operands (bf16).  Usage: python tools/diag.py case..."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, ROOT)
from cases import CASES  # noqa: E402
from util import golden_results, load_npz, namespace_of, quantized_copy, replay  # noqa: E402

import oracle.heads  # noqa: E402
import oracle.layers  # noqa: E402
import sihl_amd.heads  # noqa: E402
import sihl_amd.layers  # noqa: E402

HIP = namespace_of(sihl_amd.layers, sihl_amd.heads)
ORA = namespace_of(oracle.layers, oracle.heads)


def show(tag, res, ref):
    for k, g in ref.items():
        if not g.is_floating_point():
            continue
        r, g = res[k].float(), g.float()
        e = (r - g).abs()
        print(f"{tag:34s} {k:44s} max|g|={float(g.abs().max()):.2e} rel2max={float(e.max()) / max(1e-9, float(g.abs().max())):.2e} "
              f"rms_rel={float(e.pow(2).mean().sqrt() / g.pow(2).mean().sqrt().clamp(min=1e-12)):.2e}")


for name in sys.argv[1:]:
    data = load_npz(name)
    _, res = replay(CASES[name], HIP, data, device="cuda", dtype=torch.float32)
    show(f"{name} fp32-vs-golden", res, golden_results(data))
    q = quantized_copy(data)
    _, ref = replay(CASES[name], ORA, q)
    _, res = replay(CASES[name], HIP, q, device="cuda", dtype=torch.bfloat16)
    show(f"{name} bf16-vs-q-oracle", res, ref)
