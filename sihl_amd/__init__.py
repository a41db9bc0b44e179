"""sihl_amd: MI355X-native (gfx950) drop-in for the backbone -> FPN/BiFPN -> dense-head hot path of
jonregef/sihl.  Public surface mirrors the reference: ``SihlModel``, ``layers``, ``heads``,
``TorchvisionBackbone``-style level-list backbones."""
from sihl_amd import heads, layers  # noqa: F401
from sihl_amd.backbone import ResNetBackbone, TimmBackbone, TorchvisionBackbone  # noqa: F401
from sihl_amd.model import SihlModel  # noqa: F401

__version__ = "0.1.0"
