"""Thin model wrapper: device placement + checkpoint loading (basicsr/models/base_model.py:89-103,283-343).
Optimisers, schedulers, validation and saving are outside the hot path."""
import torch


class BaseModel:
    def __init__(self, opt):
        self.opt = opt
        self.device = torch.device("cuda" if opt.get("num_gpu", 1) != 0 and torch.cuda.is_available() else "cpu")
        self.is_train = opt.get("is_train", False)

    def model_to_device(self, net):
        return net.to(self.device)

    def load_network(self, net, load_path, strict=True, param_key="params"):
        ck = torch.load(load_path, map_location="cpu", weights_only=True)
        sd = ck[param_key] if param_key is not None and param_key in ck else ck
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
        net.load_state_dict(sd, strict=strict)
