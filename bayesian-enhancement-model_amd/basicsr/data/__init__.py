"""Tensor dataset shim of the training driver (stands in for basicsr/data/__init__.py:21-82 build_dataset / build_dataloader).

The reference's datasets (``Dataset_PairedImage_Mask`` and twelve others: cv2 / lmdb readers, augmentation, worker pools) are host I/O
outside the hot path.  What the training step consumes from them is a dict of tensors per iteration
(paired_image_dataset.py:235-412): ``lq``, ``gt`` (B,3,S,S) in [0,1], ``lq_down``, ``gt_down`` (the x1/scale_down INTER_LINEAR
resize), ``mask`` (B,S/16,S/16) for the Stage-I MIM token, ``lq_path``.  Two sources produce exactly that:

  type: TensorPairs     ``pairs: file.pt`` -- a dict(lq=(N,3,H,W), gt=(N,3,H,W)) saved with torch.save (loaded weights_only);
  type: Synthetic       ``num_images`` seeded LOL-like pairs, ``gt_size`` square.

Random crops of ``gt_size``, optional flips (``geometric_augs``), the condition planes and the mask are made on the device by the
HIP kernels of bem.ops / torch index arithmetic on the batch -- no worker processes.  A reference option file that names
``Dataset_PairedImage_Mask`` is redirected here only when the driver is given ``--synthetic`` or ``--pairs``; otherwise it raises."""
import math

import torch

__all__ = ["build_dataset", "build_dataloader", "TensorPairDataset", "TensorBatchLoader"]


class TensorPairDataset:
    def __init__(self, opt):
        self.opt = opt
        kind = opt.get("type")
        if kind == "TensorPairs":
            blob = torch.load(opt["pairs"], map_location="cpu", weights_only=True)
            self.lq, self.gt = blob["lq"].float(), blob["gt"].float()
        elif kind == "Synthetic":
            from bem.pipeline import synthetic_pair
            s = int(opt.get("image_size", opt.get("gt_size", 128)))
            self.lq, self.gt = synthetic_pair((int(opt.get("num_images", 32)), 3, s, s), seed=int(opt.get("seed", 287128)))
        else:
            raise NotImplementedError(
                f"dataset type {kind}: file-backed datasets are host I/O outside the HIP path; run the driver with --synthetic N or "
                f"--pairs file.pt (basicsr.data), or set datasets.<phase>.type to TensorPairs / Synthetic")
        if self.lq.shape != self.gt.shape or self.lq.dim() != 4 or self.lq.shape[1] != 3:
            raise ValueError("TensorPairDataset: lq and gt must both be (N,3,H,W)")

    def __len__(self):
        return self.lq.shape[0]


class TensorBatchLoader:
    """Iterates one epoch of device batches.  Order and crops come from a CPU generator seeded by (seed, epoch, rank): a resumed run
    that restarts an epoch draws the same batches as the uninterrupted one."""

    def __init__(self, dataset, dataset_opt, device, seed=0, rank=0, world=1, condition=None, train=True):
        self.dataset, self.opt, self.device, self.seed, self.rank, self.world, self.train = dataset, dataset_opt, device, int(seed or 0), rank, world, train
        self.batch = int(dataset_opt.get("batch_size_per_gpu", 1)) if train else 1
        self.crop = int(dataset_opt.get("gt_size", 0) or 0) if train else 0
        self.cond = condition or dataset_opt.get("condition") or {}
        self.epoch = 0
        self.lq, self.gt = dataset.lq.to(device), dataset.gt.to(device)

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        n = len(self.dataset) * int(self.opt.get("dataset_enlarge_ratio", 1)) if self.train else len(self.dataset)
        return math.ceil(n / (self.batch * self.world)) if self.train else n

    def __iter__(self):
        from bem import ops
        g = torch.Generator().manual_seed(self.seed * 1000003 + self.epoch * 101 + self.rank)
        N = len(self.dataset)
        s = int(self.cond.get("scale_down", 16))
        if self.train:
            n = N * int(self.opt.get("dataset_enlarge_ratio", 1))
            order = (torch.randperm(n, generator=g) % N) if self.opt.get("use_shuffle", True) else torch.arange(n) % N
            order = order[self.rank::self.world]
        else:
            order = torch.arange(N)
        for i in range(0, len(order), self.batch):
            idx = order[i:i + self.batch]
            if self.train and len(idx) < self.batch:
                break
            lq, gt = self.lq[idx.to(self.device)], self.gt[idx.to(self.device)]
            H, W = lq.shape[-2:]
            if self.crop and (H > self.crop or W > self.crop):
                t, l = int(torch.randint(0, H - self.crop + 1, (1,), generator=g)), int(torch.randint(0, W - self.crop + 1, (1,), generator=g))
                lq, gt = lq[..., t:t + self.crop, l:l + self.crop], gt[..., t:t + self.crop, l:l + self.crop]
            if self.train and self.opt.get("geometric_augs", False):
                k = int(torch.randint(0, 4, (1,), generator=g))
                if k & 1:
                    lq, gt = lq.flip(-1), gt.flip(-1)
                if k & 2:
                    lq, gt = lq.flip(-2), gt.flip(-2)
            lq, gt = lq.contiguous(), gt.contiguous()
            f = 4 * s
            hp, wp = lq.shape[-2] % f, lq.shape[-1] % f
            if hp or wp:                               # validation on whole images: pad like eval.py:146-153 before the condition planes
                Hp, Wp = lq.shape[-2] + (f - hp) % f, lq.shape[-1] + (f - wp) % f
                lqp, gtp = ops.pad_reflect(lq, Hp, Wp), ops.pad_reflect(gt, Hp, Wp)
            else:
                lqp, gtp = lq, gt
            out = dict(lq=lqp, gt=gtp, lq_down=ops.resize_down(lqp, s), gt_down=ops.resize_down(gtp, s), crop_hw=tuple(lq.shape[-2:]),
                       lq_path=[f"tensor_{int(j):05d}" for j in idx])
            if self.train and self.opt.get("mask_ratio") is not None:
                hd, wd = lqp.shape[-2] // s, lqp.shape[-1] // s
                out["mask"] = (torch.rand(lq.shape[0], hd, wd, generator=g) < float(self.opt["mask_ratio"])).float().to(self.device)
            yield out


def build_dataset(dataset_opt):
    return TensorPairDataset(dataset_opt)


def build_dataloader(dataset, dataset_opt, num_gpu=1, dist=False, sampler=None, seed=None, device="cuda", rank=0, world=1, train=None):
    train = dataset_opt.get("phase", "train") == "train" if train is None else train
    return TensorBatchLoader(dataset, dataset_opt, device, seed=seed, rank=rank, world=world, train=train)
