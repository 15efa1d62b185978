"""ImageEnhancer (basicsr/models/image_enhancer_model.py): Stage-II net wrapper -- build / load (:28-63) and the training step
(:64-128 settings, :130-148 feed_train_data, :165-216 optimize_parameters).

The step runs entirely on the HIP path: condition upsample + concat (bem.ops), forward / backward (bem.autograd), global-norm clip +
AdamW on one flat buffer (bem.train.BemAdamW), with no host synchronisation inside ``optimize_parameters`` -- loss values and the
gradient norm come back as device scalars (read them when logging).  AMP (`use_amp`) is not offered: the path is f32."""
from collections import OrderedDict

import torch

from basicsr.archs import build_network
from basicsr.losses import build_loss
from basicsr.models.base_model import BaseModel
from basicsr.utils.registry import MODEL_REGISTRY
from bem import ops
from bem.train import BemAdamW


@MODEL_REGISTRY.register()
class ImageEnhancer(BaseModel):
    def __init__(self, opt):
        super().__init__(opt)
        self.net_g = build_network(opt["network_g"])
        self.net_g = self.model_to_device(self.net_g)
        path = opt["path"].get("pretrain_network_g")
        if path is not None:
            self.load_network(self.net_g, path, opt["path"].get("strict_load_g", True), opt["path"].get("param_key", "params"))
        self.mixing_flag = False
        self.mask = None
        if self.is_train:
            if opt.get("use_amp") or opt.get("train", {}).get("use_amp"):
                raise NotImplementedError("ImageEnhancer: use_amp is not available on the f32 HIP path")
            if opt["train"].get("mixing_augs", {}).get("mixup", False):
                raise NotImplementedError("ImageEnhancer: mixup augmentation is host-side data preparation outside the hot path")
            self.init_training_settings()

    # ---------------------------------------------------------------------------------------------------------------
    def init_training_settings(self):
        self.net_g.train()
        train_opt = self.opt["train"]
        self.ema_decay = train_opt.get("ema_decay", 0)
        if self.ema_decay > 0:
            raise NotImplementedError("ImageEnhancer: ema_decay > 0 (a second, averaged copy of the net) is not used by the shipped option files")
        self.cri_pix = build_loss(train_opt["pixel_opt"]).to(self.device) if train_opt.get("pixel_opt") else None
        self.cri_perceptual = build_loss(train_opt["perceptual_opt"]).to(self.device) if train_opt.get("perceptual_opt") else None
        if self.cri_pix is None and self.cri_perceptual is None:
            raise ValueError("Both pixel and perceptual losses are None.")
        self.optimizers, self.schedulers = [], []
        self.setup_optimizers()
        self.setup_schedulers()

    def setup_optimizers(self):
        train_opt = self.opt["train"]
        normal, custom = [], []
        for k, v in self.net_g.named_parameters():
            if v.requires_grad:
                (custom if "impfusion" in k else normal).append(v)
        groups = [{"params": normal, "lr_mult": 1, "name": "normal_params"},
                  {"params": custom, "lr_mult": 1, "decay_mult": 0, "name": "custom_params"}]
        cfg = dict(train_opt["optim_g"])
        kind = cfg.pop("type")
        if kind != "AdamW":
            raise NotImplementedError(f"optimizer {kind} is not supperted yet.")      # the shipped option files use AdamW
        self.optimizer_g = BemAdamW(groups, **cfg)
        self.optimizers.append(self.optimizer_g)

    def feed_train_data(self, data):
        dev = self.device
        self.lq = data["lq"].to(dev)
        self.gt = data["gt"].to(dev) if "gt" in data else None
        self.mask = data["mask"].to(dev) if "mask" in data else None
        cond = self.opt["condition"]
        if cond["type"] == "histogram":
            raise NotImplementedError("condition type 'histogram' is not used by the shipped option files")
        gd = data["gt_down"].to(dev).contiguous()
        nl = cond.get("noise_level", 0)
        # conds = gt_down + randn_like(gt_down) * noise_level  (:143-148); the draw comes from the device Philox stream in the reserved
        # condition-noise key space (bit 62, as BEMPipeline.candidates): unique per rank and per iteration, so replicas never share a draw
        # and a resumed run continues the sequence instead of replaying it
        it = int(getattr(self, "current_iter_hint", 0)) or (getattr(self, "_cond_calls", 0) + 1)
        self._cond_calls = it
        key = (1 << 62) | (int(self.opt.get("rank", 0)) << 44) | (it & ((1 << 44) - 1))
        self.conds = ops.add(gd, ops.randn(tuple(gd.shape), dev, int(self.opt.get("manual_seed", 0) or 0), key), nl) if nl else gd

    feed_data = feed_train_data

    def optimize_parameters(self, current_iter):
        self.optimizer_g.zero_grad()
        s = self.opt["condition"].get("scale_down", 0) + self.opt["condition"].get("hist_patch_size", 0)
        B, _, H, W = self.lq.shape
        x = torch.empty(B, 6, H, W, device=self.lq.device, dtype=torch.float32)
        ops.copy_channels(self.lq.contiguous(), x, 0)
        ops.bilinear_up(self.conds, s, dst=x, dst_c0=3)                # F.interpolate(conds, scale_factor=s, 'bilinear') into channels 3..5
        _, preds = self.net_g(x, mask=None)                           # the Decomp* archs ignore the MIM mask (DDWavelet_arch.py:301)
        loss_dict = OrderedDict()
        if self.cri_pix is None:
            raise NotImplementedError("ImageEnhancer: the pixel loss is the only loss on the HIP path")
        l_total = l_pix = self.cri_pix(preds, self.gt)
        w = self.opt["train"]["pixel_opt"].get("loss_weight", 1)
        loss_dict["l_pix"] = l_pix.detach() if w == 1 else l_pix.detach() / w
        l_total.backward()
        self.sync_gradients(self.optimizer_g)
        mgn = self.opt["train"].get("max_grad_norm")
        total_norm = self.optimizer_g.clip_grad_norm_(mgn if mgn else float("inf"))
        self.optimizer_g.step()
        self.log_dict = self.reduce_loss_dict(loss_dict)
        return total_norm

    # -- validation (image_enhancer_model.py:259-325) -------------------------------------------------------------------------
    @torch.no_grad()
    def validation(self, dataloader, current_iter, tb_logger=None, save_img=False, rgb2bgr=True, use_image=True):
        """Mean PSNR of net_g over a validation loader, conditions derived from the ground truth as in training (feed_data: gt_down +
        noise).  The inference kernels run the forward; PSNR is the float form ``10 log10(1 / mse)`` on [0,1] tensors computed on the
        device (the reference's ``use_image: false`` branch; its uint8 / cv2 image branch and image dumps are host-side reporting)."""
        from bem import ops
        was_training = self.net_g.training
        self.net_g.eval()
        s = self.opt["condition"].get("scale_down", 0) + self.opt["condition"].get("hist_patch_size", 0)
        tot, cnt = 0.0, 0
        for data in dataloader:
            self.feed_train_data(data)
            B, _, H, W = self.lq.shape
            x = torch.empty(B, 6, H, W, device=self.lq.device, dtype=torch.float32)
            ops.copy_channels(self.lq.contiguous(), x, 0)
            ops.bilinear_up(self.conds, s, dst=x, dst_c0=3)
            pred = self.net_g(x)[-1]
            h, w = data.get("crop_hw", (H, W))
            gt = self.gt[..., :h, :w].contiguous()
            _, ps = ops.candidate_finalize(pred.contiguous(), gt, 1, h, w, False)
            tot += float(ps.sum())
            cnt += B
        if was_training:
            self.net_g.train()
        self.metric_results = {"psnr": tot / max(cnt, 1)}
        if self.opt.get("rank", 0) == 0:
            print(f"Validation {getattr(dataloader.dataset, 'opt', {}).get('name', 'val')},\t\t # psnr: {self.metric_results['psnr']:.4f}", flush=True)
        return self.metric_results["psnr"]

    # -- checkpoints (image_enhancer_model.py:340-370) --------------------------------------------------------------------
    def save(self, epoch, current_iter, **kwargs):
        self.save_network(self.net_g, "net_g", current_iter)
        self.save_training_state(epoch, current_iter, **kwargs)

    def save_best(self, best_metric, param_key="params"):
        """``<experiments_root>/best_psnr_<psnr>_<iter>.pth``, replacing any earlier best_* file."""
        import glob
        import os
        if self.opt.get("rank", 0) != 0:
            return None
        root = self.opt["path"]["experiments_root"]
        path = os.path.join(root, f"best_psnr_{best_metric['psnr']:.2f}_{best_metric['iter']}.pth")
        if not os.path.exists(path):
            for f in glob.glob(f"{root}/best_*"):
                os.remove(f)
            sd = {(k[7:] if k.startswith("module.") else k): v.detach().cpu() for k, v in self.get_bare_model(self.net_g).state_dict().items()}
            os.makedirs(root, exist_ok=True)
            torch.save({param_key: sd}, path)
        return path
