"""torchvision.ops.MLP mirror (Linear -> LayerNorm -> SiLU -> Dropout, ..., Linear -> Dropout) whose
forward runs the HIP linear (matrix-core 1x1) and LayerNorm+SiLU kernels.  Same 18-module layout and
state_dict keys as the reference's heads (src/sihl/heads/object_detection.py:51-61)."""
from typing import List

import torch
from torch import Tensor, nn

from sihl_amd import ops


FUSE_WHOLE_MLP = True  # test / A-B switch: False = layer by layer (Linear kernel + LayerNorm kernel) in inference too


def _parts(mlp):
    mods = list(mlp)
    linears = [m for m in mods if isinstance(m, nn.Linear)]
    norms = [m for m in mods if isinstance(m, nn.LayerNorm)]
    rest = [m for m in mods if not isinstance(m, (nn.Linear, nn.LayerNorm, nn.Dropout))]
    act = "silu" if rest and all(isinstance(m, nn.SiLU) for m in rest) else (None if not rest else "?")
    return linears, norms, rest, act


def _drops(mlp) -> bool:
    """True when this MLP would really drop activations now (training mode with a Dropout of p > 0): the fused kernels have
    no dropout, and the layer-by-layer path refuses that case loudly instead of skipping it."""
    return mlp.training and any(isinstance(m, nn.Dropout) and m.p > 0 for m in mlp)


def forward_many(mlps, x: Tensor):
    """[m(x) for m in mlps] for MLPs that read the same rows - in inference as ONE launch when the register kernel covers
    them all (ops.mlp_fused_multi), else one by one."""
    if FUSE_WHOLE_MLP and not torch.is_grad_enabled() and len(mlps) > 1 and not any(_drops(m) for m in mlps):
        lead = x.shape[:-1]
        h = x.reshape(-1, x.shape[-1])
        parts = [_parts(m) for m in mlps]
        acts = {p[3] for p in parts}
        if len(acts) == 1 and "?" not in acts and all(len(p[2]) == len(p[1]) for p in parts):
            outs = ops.mlp_fused_multi(h, [(p[0], p[1]) for p in parts], parts[0][3])
            if outs is not None:
                return [o.reshape(*lead, o.shape[-1]) for o in outs]
    return [m(x) for m in mlps]


class MLP(nn.Sequential):
    def __init__(self, in_channels: int, hidden_channels: List[int], norm_layer=None, activation_layer=nn.ReLU,
                 inplace=None, bias=True, dropout=0.0):
        kw = {} if inplace is None else {"inplace": inplace}
        mods: List[nn.Module] = []
        d = in_channels
        for h in hidden_channels[:-1]:
            mods.append(nn.Linear(d, h, bias=bias))
            if norm_layer is not None:
                mods.append(norm_layer(h))
            mods.append(activation_layer(**kw))
            mods.append(nn.Dropout(dropout, **kw))
            d = h
        mods.append(nn.Linear(d, hidden_channels[-1], bias=bias))
        mods.append(nn.Dropout(dropout, **kw))
        super().__init__(*mods)

    def forward(self, x: Tensor) -> Tensor:
        """x: (..., C) -> (..., out); runs over flattened rows."""
        lead = x.shape[:-1]
        h = x.reshape(-1, x.shape[-1])
        mods = list(self)
        if FUSE_WHOLE_MLP and not torch.is_grad_enabled() and not _drops(self):
            # inference: the whole chain in one launch, activations resident in LDS between the layers (ops.mlp_fused)
            linears = [m for m in mods if isinstance(m, nn.Linear)]
            norms = [m for m in mods if isinstance(m, nn.LayerNorm)]
            rest = [m for m in mods if not isinstance(m, (nn.Linear, nn.LayerNorm, nn.Dropout))]
            act = "silu" if rest and all(isinstance(m, nn.SiLU) for m in rest) else (None if not rest else "?")
            if act != "?" and len(rest) == len(norms) and ops.mlp_fused_supported(h, linears, norms, act):
                out = ops.mlp_fused(h, linears, norms, act)
                return out.reshape(*lead, out.shape[-1])
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.Linear):
                nxt = mods[i + 1] if i + 1 < len(mods) else None
                if isinstance(nxt, nn.LayerNorm):
                    act = mods[i + 2]
                    if not isinstance(act, nn.SiLU):
                        raise NotImplementedError("LayerNorm is fused with SiLU only")
                    h = ops.layernorm_act(ops.linear(h, m.weight, m.bias), nxt.weight, nxt.bias, nxt.eps, "silu")
                    i += 3
                    continue
                h = ops.linear(h, m.weight, m.bias)
            elif isinstance(m, nn.Dropout):
                if m.p != 0.0 and self.training:
                    raise NotImplementedError("dropout > 0 is outside the HIP hot path")
            else:
                raise NotImplementedError(f"unsupported MLP layer {type(m).__name__}")
            i += 1
        return h.reshape(*lead, h.shape[-1])
