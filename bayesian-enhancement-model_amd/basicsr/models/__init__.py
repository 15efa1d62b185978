"""build_model (basicsr/models/__init__.py:18-29) for the two model types on the hot path."""
from copy import deepcopy

from basicsr.utils.registry import MODEL_REGISTRY
from . import condition_generator_model, image_enhancer_model  # noqa: F401

__all__ = ["build_model"]


def build_model(opt):
    opt = deepcopy(opt)
    return MODEL_REGISTRY.get(opt["model_type"])(opt)
