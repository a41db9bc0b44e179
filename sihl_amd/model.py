"""Model assembly for the hot path.

``SihlModel(backbone, neck, heads)`` has the constructor and the two entry points of the reference's class of the
same name (src/sihl/sihl_model.py:6-25): ``extract_features`` gives the level list every head reads, ``forward``
gives one inference output per head.  The level list is computed once per call whatever the number of heads.
On this path the parameters live channels-last (the conv kernels read KRSC weights directly), which
``to_device`` arranges in one place.
"""
from typing import Any, Iterable, List, Optional

import torch
from torch import Tensor, nn


class SihlModel(nn.Module):
    def __init__(self, backbone: nn.Module, neck: Optional[nn.Module], heads: Iterable[nn.Module]) -> None:
        super().__init__()
        self.backbone = backbone
        self.neck = neck  # None: heads read the backbone's levels directly
        self.heads = nn.ModuleList(heads)

    def extract_features(self, input: Tensor) -> List[Tensor]:
        stages = (self.backbone,) if self.neck is None else (self.backbone, self.neck)
        levels: Any = input
        for stage in stages:
            levels = stage(levels)
        return levels

    def forward(self, input: Tensor) -> List[Any]:
        levels = self.extract_features(input)
        return [head(levels) for head in self.heads]

    def to_device(self, device) -> "SihlModel":
        """Move to ``device`` with 4-d parameters channels-last, the layout the HIP conv kernels consume."""
        return self.to(device).to(memory_format=torch.channels_last)

    def prepare_inference(self, dtype: torch.dtype = torch.bfloat16):
        """bf16 inference: make the bf16 operand copies of every conv / linear weight once (one kernel) instead of
        casting per layer per call.  Returns the PreparedWeights handle; call its ``refresh()`` after changing weights
        (``load_state_dict`` bumps the version stamps, so stale copies are never used silently)."""
        from sihl_amd import ops

        self._prepared = ops.PreparedWeights(self, dtype)
        return self._prepared

    def num_parameters(self, trainable_only: bool = False) -> int:
        return sum(p.numel() for p in self.parameters() if p.requires_grad or not trainable_only)
