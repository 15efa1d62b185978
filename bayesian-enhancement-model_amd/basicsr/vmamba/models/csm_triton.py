"""Operator seam of basicsr/vmamba/models/csm_triton.py:491-505 (channel-first, scans=0, not one-by-one), forward
and backward: like CrossScanF / CrossMergeF (csm_triton.py:190-273) each op's backward is the other op's forward."""
import torch

from bem import ops


def _only_default(in_channel_first, out_channel_first, one_by_one, scans):
    if not (in_channel_first and out_channel_first) or one_by_one or scans != 0:
        raise NotImplementedError("only the channel-first cross2d form (scans=0) is used on the BEM path")


class CrossScanHip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.shape = x.shape
        return ops.cross_scan(x.detach().float().contiguous())

    @staticmethod
    def backward(ctx, ys):
        B, C, H, W = ctx.shape
        return ops.cross_merge(ys.float().contiguous().view(B, 4, C, H, W)).view(B, C, H, W)


class CrossMergeHip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ys):
        ctx.shape = ys.shape
        return ops.cross_merge(ys.detach().float().contiguous())

    @staticmethod
    def backward(ctx, y):
        B, K, C, H, W = ctx.shape
        return ops.cross_scan(y.float().contiguous().view(B, C, H, W)).view(B, 4, C, H, W)


def cross_scan_fn(x, in_channel_first=True, out_channel_first=True, one_by_one=False, scans=0, force_torch=False):
    _only_default(in_channel_first, out_channel_first, one_by_one, scans)
    return CrossScanHip.apply(x)


def cross_merge_fn(y, in_channel_first=True, out_channel_first=True, one_by_one=False, scans=0, force_torch=False):
    _only_default(in_channel_first, out_channel_first, one_by_one, scans)
    return CrossMergeHip.apply(y)
