"""build_network (basicsr/archs/__init__.py:18-24) over the HIP-backed architectures."""
from copy import deepcopy

from basicsr.utils import get_root_logger
from basicsr.utils.registry import ARCH_REGISTRY
from bem import archs as _a

for _cls in (_a.Network, _a.DecompDualBranchDDWavelet, _a.DecompSingleBranch, _a.DecompDualBranch2DD, _a.DecompDualBranch2,
             _a.DecompSingleBranchDD, _a.DecompDualBranch):
    if _cls.__name__ not in ARCH_REGISTRY:
        ARCH_REGISTRY.register(_cls)

__all__ = ["build_network"]


def build_network(opt):
    opt = deepcopy(opt)
    network_type = opt.pop("type")
    net = ARCH_REGISTRY.get(network_type)(**opt)
    get_root_logger().info(f"Network [{net.__class__.__name__}] is created.")
    return net
