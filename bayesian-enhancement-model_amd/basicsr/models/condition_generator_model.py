"""ConditionGenerator (basicsr/models/condition_generator_model.py:28-75): Stage-I net + BNN conversion."""
from basicsr.archs import build_network
from basicsr.bayesian import convert2bnn, convert2bnn_selective
from basicsr.models.base_model import BaseModel
from basicsr.utils.registry import MODEL_REGISTRY


@MODEL_REGISTRY.register()
class ConditionGenerator(BaseModel):
    def __init__(self, opt):
        super().__init__(opt)
        self.net_g = build_network(opt["network_g"])
        cfg = {"sigma_init": opt.get("sigma_init", 0.05), "decay": 0.998, "pretrain": False}
        (convert2bnn_selective if opt.get("selective", True) else convert2bnn)(self.net_g, cfg)
        self.net_g = self.model_to_device(self.net_g)
        path = opt["path"].get("pretrain_network_g")
        if path is not None:
            self.load_network(self.net_g, path, opt["path"].get("strict_load_g", True), opt["path"].get("param_key", "params"))
        if self.is_train:
            raise NotImplementedError("Stage-I training (KL + EMA prior) is a later row of SURVEY.md section 8f")
