"""ConditionGenerator (basicsr/models/condition_generator_model.py): Stage-I net + BNN conversion (:28-75), training settings and
optimizer (:77-144), data feed (:146-174), the training step with the KL term (:176-218), checkpoints (:346-370)."""
import os
from collections import OrderedDict

import torch

from basicsr.archs import build_network
from basicsr.bayesian import convert2bnn, convert2bnn_selective, get_kl_loss
from basicsr.losses import build_loss
from basicsr.models.base_model import BaseModel
from basicsr.utils.registry import MODEL_REGISTRY
from bem import autograd as ag
from bem.train import BemAdamW


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
        self.mask = None
        if self.is_train:
            if opt.get("use_amp"):
                raise NotImplementedError("ConditionGenerator: use_amp is not available on the f32 HIP path")
            if opt["train"].get("mixing_augs", {}).get("mixup", False):
                raise NotImplementedError("ConditionGenerator: mixup augmentation is host-side data preparation outside the hot path")
            self.init_training_settings()

    def init_training_settings(self):
        self.net_g.train()
        train_opt = self.opt["train"]
        self.ema_decay = train_opt.get("ema_decay", 0)
        if self.ema_decay > 0:
            raise NotImplementedError("ConditionGenerator: ema_decay > 0 (a second, averaged copy of the net) is not used by the shipped option files")
        self.cri_pix = build_loss(train_opt["pixel_opt"]).to(self.device) if train_opt.get("pixel_opt") else None
        self.cri_perceptual = build_loss(train_opt["perceptual_opt"]).to(self.device) if train_opt.get("perceptual_opt") else None
        if self.cri_pix is None and self.cri_perceptual is None:
            raise ValueError("Both pixel and perceptual losses are None.")
        self.optimizers, self.schedulers = [], []
        self.setup_optimizers()
        self.setup_schedulers()

    def setup_optimizers(self):
        normal, custom = [], []
        for k, v in self.net_g.named_parameters():
            if v.requires_grad:
                (custom if "impfusion" in k else normal).append(v)
        groups = [{"params": normal, "lr_mult": 1, "name": "normal_params"},
                  {"params": custom, "lr_mult": 1, "decay_mult": 0, "name": "custom_params"}]
        cfg = dict(self.opt["train"]["optim_g"])
        kind = cfg.pop("type")
        if kind != "AdamW":
            raise NotImplementedError(f"optimizer {kind} is not supperted yet.")
        self.optimizer_g = BemAdamW(groups, **cfg)
        self.optimizers.append(self.optimizer_g)

    def feed_train_data(self, data):
        kind = self.opt["condition"]["type"]
        if kind == "mean":
            self.lq = data["lq_down"].to(self.device)
            if "gt" in data:
                self.gt = data["gt_down"].to(self.device)
        elif kind == "histogram":
            self.lq = data["hist_lq"].to(self.device)
            if "gt" in data:
                self.gt = data["hist_gt"].to(self.device)
        self.mask = data["mask"].to(self.device) if "mask" in data else None

    def feed_data(self, data):
        kind = self.opt["condition"]["type"]
        if kind not in ("mean", "histogram"):
            raise NotImplementedError(f"{kind} is not supported yet.")
        self.feed_train_data({k: v for k, v in data.items() if k != "mask"})

    def optimize_parameters(self, current_iter):
        """zero_grad -> net_g(lq, mask) (mask dropped after the first scheduler period) -> l_total = 0.01 * l_kl / mini_batch + l_pix ->
        backward -> clip_grad_norm_ -> AdamW (:176-218).  The train.png dumps every 100 iterations are host-side logging, not done here.

        The step is ~2000 launches on 8x8 .. 2x2 planes, i.e. launch-bound: after one ordinary run per input geometry it is captured into a
        HIP graph and replayed, with everything that differs between iterations read from device memory
        (bem.train.StepState).  BEM_STAGE1_GRAPH=0, injected eps, a distributed run or an active launch profile keep the ordinary form."""
        if current_iter > self.opt["train"]["scheduler"]["periods"][0]:
            self.mask = None
        from bem import ops
        from bem.modules import _SAMPLE_CTX
        skip = [self.net_g.mask_token] if self.mask is None else []
        if (os.environ.get("BEM_STAGE1_GRAPH", "1") != "0" and _SAMPLE_CTX[0] is None and not self.opt.get("dist") and ops._PROF is None
                and self.optimizer_g.graph_safe(skip)):
            return self._optimize_graphed(current_iter, skip)
        return self._step_body(current_iter, skip, None)

    def _step_body(self, current_iter, skip, state):
        from bem.modules import _SAMPLE_CTX, SampleCtx, sampling
        self.optimizer_g.zero_grad()
        # the weight draws of iteration i come from Philox streams keyed by (manual_seed, rank, i): a resumed run draws what the
        # uninterrupted run would have drawn (the reference's resumed run restarts torch's generator instead)
        seed, rank = int(self.opt.get("manual_seed") or 0) & 0xFFFFFFFF, int(self.opt.get("rank", 0))
        if state is not None:
            ctx = SampleCtx(1, None, seed=seed, rank=rank, epoch_dev=state.epoch_dev)
        else:
            ctx = _SAMPLE_CTX[0] or SampleCtx(1, None, seed=seed, rank=rank, epoch=int(current_iter) & 0xFFFFFF)   # an enclosing context (injected eps) wins
        with sampling(ctx):
            _, preds = self.net_g(self.lq.contiguous(), mask=self.mask)
        loss_dict = OrderedDict()
        l_kl = get_kl_loss(self.net_g)
        loss_dict["l_kl"] = l_kl.detach()
        if self.cri_pix is None:
            raise NotImplementedError("ConditionGenerator: the pixel loss is the only image loss on the HIP path")
        l_pix = self.cri_pix(preds, self.gt)
        w = self.opt["train"]["pixel_opt"].get("loss_weight", 1)
        loss_dict["l_pix"] = l_pix.detach() if w == 1 else l_pix.detach() / w
        l_total = ag.ScaledSumFn.apply(l_pix, l_kl, 0.01 / self.opt["datasets"]["train"]["mini_batch_sizes"][0])
        l_total.backward()
        self.sync_gradients(self.optimizer_g)
        mgn = self.opt["train"].get("max_grad_norm")
        total_norm = self.optimizer_g.clip_grad_norm_(mgn if mgn else float("inf"))
        # a parameter outside this iteration's graph has .grad None in the reference and is skipped by AdamW: the mask token without a mask
        if state is not None:
            self.optimizer_g.step_captured(state, skip=skip)
            return loss_dict, total_norm
        self.optimizer_g.step(skip=skip)
        self.log_dict = self.reduce_loss_dict(loss_dict)
        return total_norm

    def _optimize_graphed(self, current_iter, skip):
        """One captured step per input geometry: {"graph", "state", static inputs, the step's loss tensors}.  The first iteration with a
        geometry runs the ordinary way (it is also the warm-up that leaves every lazily created buffer in place); the second records the
        step -- recording launches nothing -- and is then replayed like all later ones."""
        import bem.train as bt
        key = (tuple(self.lq.shape), tuple(self.gt.shape), None if self.mask is None else tuple(self.mask.shape))
        graphs = self.__dict__.setdefault("_graphs", {})
        g = graphs.get(key)
        bank = self.get_bare_model(self.net_g).__dict__.get("_bayes_bank")
        if isinstance(g, dict) and g["bank_sig"] is not (bank.sig if bank is not None else None):
            g = None                 # the bank rebuilt its tables (a parameter / gradient / prior buffer moved): the recorded launches point at the old ones
        if g is None:
            graphs[key] = "warm"
            return self._step_body(current_iter, skip, None)
        if g == "warm":
            g = {"state": bt.StepState(self.lq.device), "lq": self.lq.clone(), "gt": self.gt.clone(),
                 "mask": None if self.mask is None else self.mask.clone(), "graph": torch.cuda.CUDAGraph()}
            feed = self.lq, self.gt, self.mask
            self.lq, self.gt, self.mask = g["lq"], g["gt"], g["mask"]
            torch.cuda.synchronize()
            bt.STEP_STATE[0] = g["state"]
            try:
                with torch.cuda.graph(g["graph"]):
                    g["losses"], g["norm"] = self._step_body(current_iter, skip, g["state"])
            finally:
                bt.STEP_STATE[0] = None
                self.lq, self.gt, self.mask = feed
            bank = self.get_bare_model(self.net_g).__dict__.get("_bayes_bank")
            g["bank_sig"] = bank.sig if bank is not None else None
            graphs[key] = g
        for dst, src in ((g["lq"], self.lq), (g["gt"], self.gt), (g["mask"], self.mask)):
            if dst is not None and dst.data_ptr() != src.data_ptr():
                dst.copy_(src)
        g["state"].upload(current_iter)
        g["graph"].replay()
        g["state"].advance()
        self.log_dict = self.reduce_loss_dict(g["losses"])
        return g["norm"]

    # -- validation (:236-334) ----------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def validation(self, dataloader, current_iter, tb_logger=None, save_img=False, rgb2bgr=True, use_image=True):
        """Mean PSNR of the predicted condition planes against the down-sampled ground truth, deterministic weights (mu), float form on
        the device."""
        from basicsr.bayesian import set_prediction_type
        from bem import ops
        was_training = self.net_g.training
        self.net_g.eval()
        set_prediction_type(self.net_g, True)
        tot, cnt = 0.0, 0
        for data in dataloader:
            self.feed_data(data)
            pred = self.net_g(self.lq.contiguous())[-1]
            h, w = pred.shape[-2:]
            _, ps = ops.candidate_finalize(pred.contiguous(), self.gt.contiguous(), 1, h, w, False)
            tot += float(ps.sum())
            cnt += pred.shape[0]
        set_prediction_type(self.net_g, False)
        if was_training:
            self.net_g.train()
        self.metric_results = {"psnr": tot / max(cnt, 1)}
        if self.opt.get("rank", 0) == 0:
            print(f"Validation,\t\t # psnr: {self.metric_results['psnr']:.4f}", flush=True)
        return self.metric_results["psnr"]

    def save_best(self, best_metric, param_key="params"):
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

    # -- checkpoints (:346-370) ---------------------------------------------------------------------------------------------
    def save(self, epoch, current_iter, **kwargs):
        self.save_network(self.net_g, "net_g", current_iter)
        self.save_training_state(epoch, current_iter, **kwargs)
