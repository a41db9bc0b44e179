# Same public names as the reference's ``sihl.layers`` for the hot-path modules
# (src/sihl/layers/__init__.py), so ``getattr(sihl_amd.layers, "BiFPN")`` works as in the examples.
from sihl_amd.layers.bifpn import BiFPN, BiFPNLayer, FastNormalizedFusion  # noqa: F401
from sihl_amd.layers.convblocks import Conv2dNormActivation, ConvNormAct, SequentialConvBlocks  # noqa: F401
from sihl_amd.layers.fpn import FPN  # noqa: F401
from sihl_amd.layers.scalers import AntialiasedDownscaler, BlurPool2d, Interpolate, SimpleUpscaler  # noqa: F401
