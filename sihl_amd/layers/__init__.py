"""Neck and building-block modules of the hot path; the public names match the reference's ``sihl.layers`` for the
modules in scope, so ``getattr(sihl_amd.layers, "BiFPN")`` selects a neck by name as the reference examples do."""
from sihl_amd.layers.convblocks import Conv2dNormActivation, ConvNormAct, SequentialConvBlocks  # noqa: F401
from sihl_amd.layers.hybrid_encoder import CSPRepLayer, HybridEncoder, RepVGGBlock  # noqa: F401
from sihl_amd.layers.necks import FPN, BiFPN, BiFPNLayer, FastNormalizedFusion  # noqa: F401
from sihl_amd.layers.scalers import AntialiasedDownscaler, BlurPool2d, Interpolate, SimpleUpscaler  # noqa: F401
