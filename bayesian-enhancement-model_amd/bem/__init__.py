"""MI355X-native hot path of the Bayesian Enhancement Model: host side above the C ABI
(include/bem_hip.h).  PyTorch is used for device memory, streams and torch.distributed only;
every compute step of the path is a hand-written HIP kernel in csrc/ reached through ctypes."""
from . import native  # noqa: F401
