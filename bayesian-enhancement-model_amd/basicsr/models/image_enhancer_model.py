"""ImageEnhancer (basicsr/models/image_enhancer_model.py:28-63): Stage-II net wrapper."""
from basicsr.archs import build_network
from basicsr.models.base_model import BaseModel
from basicsr.utils.registry import MODEL_REGISTRY


@MODEL_REGISTRY.register()
class ImageEnhancer(BaseModel):
    def __init__(self, opt):
        super().__init__(opt)
        self.net_g = build_network(opt["network_g"])
        self.net_g = self.model_to_device(self.net_g)
        path = opt["path"].get("pretrain_network_g")
        if path is not None:
            self.load_network(self.net_g, path, opt["path"].get("strict_load_g", True), opt["path"].get("param_key", "params"))
        if self.is_train:
            raise NotImplementedError("the Stage-II training step (backward kernels) is SURVEY.md row A10, not built this round")
