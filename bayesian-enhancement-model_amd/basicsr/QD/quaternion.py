"""basicsr/QD/quaternion.py:3-17 on the HIP kernels: ``hamilton_product`` under the reference's name (all four components,
real part first), plus the 3-channel form the fused arch kernels use (the archs drop the real part)."""
from bem import ops


def hamilton_product(q1, q2):
    """(B,4,H,W) x (B,4,H,W) -> (B,4,H,W): [r, i, j, k] of the Hamilton product q1 * q2."""
    return ops.hamilton_full(q1.contiguous(), q2.contiguous())


def hamilton_product_ijk(q1, q2):
    """(B,4,H,W) x (B,4,H,W) -> (B,3,H,W): components i, j, k only."""
    return ops.hamilton_full(q1.contiguous(), q2.contiguous())[:, 1:].contiguous()
