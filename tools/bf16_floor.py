"""Developer probe (GPU box): bf16 parity of the deep golden cases against the noise floor of bf16 storage.
For every floating result of a case: rms-relative deviation from the fp32 oracle (same bf16-rounded operands) of
  floor = the fp32 oracle with bf16 storage emulated at every module boundary (tests/golden/util.emulate_bf16_storage)
  hip   = the HIP bf16 path."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cases import CASES  # noqa: E402
from util import emulate_bf16_storage, load_npz, namespace_of, quantized_copy, replay  # noqa: E402
import oracle.heads  # noqa: E402
import oracle.layers  # noqa: E402
from test_gpu_golden import BF16_DEEP, _ns  # noqa: E402

names = sys.argv[1:] or BF16_DEEP + ["hybrid_3to6_train", "depth_training_step"]


def rms(a, b):
    return float((a.float() - b.float()).pow(2).mean().sqrt() / b.float().pow(2).mean().sqrt().clamp(min=1e-12))


for name in names:
    q = quantized_copy(load_npz(name))
    ons = namespace_of(oracle.layers, oracle.heads)
    _, ref = replay(CASES[name], ons, q)
    _, emu = replay(CASES[name], ons, q, prepare=lambda m: (emulate_bf16_storage(m), m)[1])
    _, hip = replay(CASES[name], _ns(), q, device="cuda", dtype=torch.bfloat16)
    for k, g in ref.items():
        if g.is_floating_point():
            print(f"{name:24s} {k:28s} n={g.numel():8d}  floor {rms(emu[k], g):9.3e}  hip {rms(hip[k], g):9.3e}  "
                  f"hip-vs-floor-run {rms(hip[k], emu[k]):9.3e}", flush=True)
