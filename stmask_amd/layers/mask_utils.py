"""generate_mask (reference layers/mask_utils.py:111-128): tanh(coeff) -> proto @ coeff^T -> sigmoid -> crop -> [n,h,w],
as ONE fused gfx950 kernel (the reference: matmul + 2 activations + 8 element-wise kernels + permute copy)."""
from .. import ops


def generate_mask(proto_data, mask_coeff, bbox=None, use_sipmask=False):
    if use_sipmask:
        raise NotImplementedError("use_sipmask is False in every STMask config (config.py:704)")
    if mask_coeff.shape[0] == 0:
        return proto_data.new_zeros(0, proto_data.shape[0], proto_data.shape[1])
    return ops.lincomb_sigmoid_crop(proto_data, mask_coeff, bbox, apply_tanh=True)
