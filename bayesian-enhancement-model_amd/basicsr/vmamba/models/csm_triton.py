"""Operator seam of basicsr/vmamba/models/csm_triton.py:491-505 (channel-first, scans=0, not one-by-one)."""
from bem import ops


def _only_default(in_channel_first, out_channel_first, one_by_one, scans):
    if not (in_channel_first and out_channel_first) or one_by_one or scans != 0:
        raise NotImplementedError("only the channel-first cross2d form (scans=0) is used on the BEM path")


def cross_scan_fn(x, in_channel_first=True, out_channel_first=True, one_by_one=False, scans=0, force_torch=False):
    _only_default(in_channel_first, out_channel_first, one_by_one, scans)
    return ops.cross_scan(x.float().contiguous())


def cross_merge_fn(y, in_channel_first=True, out_channel_first=True, one_by_one=False, scans=0, force_torch=False):
    _only_default(in_channel_first, out_channel_first, one_by_one, scans)
    return ops.cross_merge(y.float().contiguous())
