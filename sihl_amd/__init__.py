"""sihl_amd: MI355X-native (gfx950) drop-in for the backbone -> FPN/BiFPN -> dense-head hot path of
jonregef/sihl.  Public surface mirrors the reference: ``SihlModel``, ``layers``, ``heads``,
``TorchvisionBackbone``-style level-list backbones."""
import os as _os

import torch as _torch

# HIP-graph replays: ROCm's "graph packet capture" (AQL packets and kernel arguments pre-built in device memory when a graph is
# instantiated, the default of this runtime) is not safe beside other device allocations - memory the process allocates
# between or before replays can land on the packets' kernel arguments, and the next replay then runs kernels with clobbered
# pointers: silently wrong results, or a GPU memory fault "on the second replay" (the graph-replay faults of rounds 1-4;
# DESIGN section 5d, profiles/r04_graph_replay_root_cause.txt).  With the optimisation off a replay re-issues its nodes from
# the host (as much host time as eager launches, the same device time) and is correct.  The variable is read when the HIP
# runtime initialises, i.e. at the first device call - importing torch does not do that.
if _os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") is None:
    _os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"
    # (set by this process: in time only if no device call has happened yet.  bench.py, tests/conftest.py and
    # __graft_entry__.py set the pair themselves, before they import torch.)
    _os.environ["SIHL_GRAPH_ENV_EARLY"] = "0" if _torch.cuda.is_initialized() else "1"


def graph_replay_safe() -> bool:
    """False when HIP graphs of this process may replay through the runtime's packet-capture path: the variable is set to
    something else, or this process set it only after the HIP runtime was up."""
    return _os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") == "0" and _os.environ.get("SIHL_GRAPH_ENV_EARLY", "1") != "0"


from sihl_amd import heads, layers  # noqa: F401,E402
from sihl_amd.backbone import ResNetBackbone, TimmBackbone, TorchvisionBackbone  # noqa: F401,E402
from sihl_amd.model import SihlModel  # noqa: F401,E402

__version__ = "0.1.0"
