"""
oracle/torch_chain.py -- TEST INFRASTRUCTURE ONLY (CPU baseline leg of bench.py and tests).

The reference's torch-backend op chain for integer compression + packing, restated op for op with
torch CPU tensors, so that it runs multi-threaded on the host exactly as the reference's
`compress_weight(Tensor(torch_weight), ...)` + `pack_int4/pack_uint4` does:
  weight_lowering.py:451-508, 292-326, 629-684; fake_quantize.py:322-350;
  torch/quantization/quantize_functions.py:417-461.
Each line below is one ATen kernel with a full-tensor round trip, as in the reference
(nncf.tensor.functions.torch_numeric). Checked bit-exact against oracle/wc_oracle.py in tests/.
"""
import torch

EPS = torch.finfo(torch.float32).eps


def compress_int_torch(weight: torch.Tensor, mode: str, group_size: int, pack: bool = True):
    """weight [out, in] (any float dtype, CPU). Returns (codes or packed, scale, zero_point|None)."""
    num_bits = 8 if mode.startswith("int8") else 4
    asym = mode.endswith("asym")
    w = weight
    if group_size != -1:
        if w.shape[1] % group_size != 0:
            raise ValueError(f"Channel size {w.shape[1]} should be divisible by size of group {group_size}.")
        w = w.reshape(w.shape[0], w.shape[1] // group_size, group_size)
    axis = w.ndim - 1
    if w.dtype != torch.float32:
        w = w.to(torch.float32)
    zero_point = None
    if asym:
        level_low, level_high = 0, 2 ** num_bits - 1
        lo = torch.amin(w, dim=axis, keepdim=True)
        hi = torch.amax(w, dim=axis, keepdim=True)
        scale = ((hi - lo) / (level_high - level_low)).to(torch.float32)
        scale = torch.where(torch.abs(scale) < EPS, EPS, scale)
        zero_point = level_low - torch.round(lo / scale)
        zero_point = torch.clip(zero_point.to(torch.int32), level_low, level_high)
    else:
        level_low, level_high = -(2 ** (num_bits - 1)), 2 ** (num_bits - 1) - 1
        w_abs_min = torch.abs(torch.amin(w, dim=axis, keepdim=True))
        w_max = torch.amax(w, dim=axis, keepdim=True)
        scale = torch.where(w_abs_min >= w_max, w_abs_min, -w_max)
        scale /= 2 ** (num_bits - 1)
        scale = torch.where(torch.abs(scale) < EPS, EPS, scale)
    cw = w / scale
    if zero_point is not None:
        cw += zero_point.to(w.dtype)
    cw = torch.round(cw)
    cw = torch.clip(cw, level_low, level_high).to(torch.uint8 if asym else torch.int8)
    if pack and num_bits == 4:
        t = cw
        if not asym:
            t = (t + 8).type(torch.uint8)
        t = t.contiguous().reshape(-1, 2)
        cw = torch.bitwise_and(t[..., ::2], 15) | t[..., 1::2] << 4
    return cw, scale, zero_point
