"""hamilton_product (basicsr/QD/quaternion.py:3-17).  The fused kernels drop the real part as the archs do;
this seam returns all four components by evaluating the (i,j,k) kernel and the real part separately is not
needed on the path, so only the 3-channel form is exposed."""
import torch

from bem import ops


def hamilton_product_ijk(q1, q2):
    """(B,4,H,W) x (B,4,H,W) -> (B,3,H,W): components i, j, k of the Hamilton product."""
    return ops.hamilton(torch.cat([q1, q2], 1).contiguous())
