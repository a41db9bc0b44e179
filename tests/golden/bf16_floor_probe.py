"""Test-infrastructure probe (GPU box; lives under tests/ because it runs the oracle): bf16 parity of the deep golden cases against the noise floor of bf16 storage.
For every floating result of a case: rms-relative deviation from the fp32 oracle (same bf16-rounded operands) of
  floor = the largest deviation over fp32 oracle runs with bf16's rounding error at every module boundary (bf16 storage
          emulated, and three random rounding patterns: tests/golden/util.bf16_floor)
  hip   = the HIP bf16 path."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cases import CASES  # noqa: E402
from util import bf16_floor, load_npz, namespace_of, quantized_copy, replay  # noqa: E402
import oracle.heads  # noqa: E402
import oracle.layers  # noqa: E402
from test_gpu_golden import BF16_DEEP, _ns  # noqa: E402

names = sys.argv[1:] or BF16_DEEP


def rms(a, b):
    return float((a.float() - b.float()).pow(2).mean().sqrt() / b.float().pow(2).mean().sqrt().clamp(min=1e-12))


for name in names:
    q = quantized_copy(load_npz(name))
    ons = namespace_of(oracle.layers, oracle.heads)
    ref, floor = bf16_floor(CASES[name], ons, q)
    _, hip = replay(CASES[name], _ns(), q, device="cuda", dtype=torch.bfloat16)
    for k, g in ref.items():
        if g.is_floating_point() and True:
            err = float((hip[k].float() - g.float()).norm() / g.float().norm().clamp(min=1e-30))
            flag = "  <-- above 3 x floor + 3e-2" if err > 3 * floor[k] + 3e-2 else ""
            print(f"{name:24s} {k:44s} n={g.numel():8d}  floor {floor[k]:9.3e}  hip {err:9.3e}{flag}", flush=True)
