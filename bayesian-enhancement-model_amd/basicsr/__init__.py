"""Drop-in ``basicsr`` surface of the MI355X-native Bayesian Enhancement Model hot path.

Only what Enhancement/eval.py-style drivers import for this path is provided:
  basicsr.models.build_model, basicsr.archs.build_network, basicsr.utils.options.parse,
  basicsr.utils.registry.{ARCH,MODEL}_REGISTRY, basicsr.bayesian.{set_prediction_type,convert2bnn_selective},
  basicsr.vmamba.models.{vmamba,csms6s,csm_triton} operator seams, basicsr.QD.{model4,quaternion}.
Datasets, losses, metrics, training loops and the classifier stack are out of scope."""
import os as _os
import sys as _sys

_pkg_root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
if _pkg_root not in _sys.path:
    _sys.path.insert(0, _pkg_root)
