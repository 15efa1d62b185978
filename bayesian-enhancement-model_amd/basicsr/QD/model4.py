"""Haar DWT / IWT modules with the reference's names (basicsr/QD/model4.py:7-49) on the HIP kernels."""
import torch.nn as nn

from bem import ops


def dwt_init(x):
    return ops.dwt(x.contiguous())


def iwt_init(x):
    return ops.iwt(x.float().contiguous())


class DWT(nn.Module):
    def forward(self, x):
        return dwt_init(x)


class IWT(nn.Module):
    def forward(self, x):
        return iwt_init(x)
