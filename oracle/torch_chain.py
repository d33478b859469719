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


def compress_float_torch(weight: torch.Tensor, mode: str, group_size: int):
    """NF4 / MXFP4 torch-CPU op chain (weight_lowering.py:84-128 scale, :613-626 normalise, :557-586 nearest quantile
    with the e2m1 tie rule).  Returns (fp32 quantile values, scale) like the reference's torch path does (:585-586).
    Checked against oracle/wc_oracle.py in tests/test_abi_cpu.py."""
    from oracle import wc_oracle as O

    w = weight
    if group_size != -1:
        if w.shape[1] % group_size != 0:
            raise ValueError(f"Channel size {w.shape[1]} should be divisible by size of group {group_size}.")
        w = w.reshape(w.shape[0], w.shape[1] // group_size, group_size)
    if w.dtype != torch.float32:
        w = w.to(torch.float32)
    nf4 = mode == "nf4"
    scale = torch.amax(torch.abs(w), dim=w.ndim - 1, keepdim=True)
    if not nf4:
        scale = scale / 6.0
    scale = torch.where(torch.abs(scale) < EPS, EPS, scale)
    if mode == "mxfp4":
        scale = torch.log2(scale)
        scale = torch.ceil(scale)
        scale = torch.clip(scale, -127, 127)
        scale = 2 ** scale
    norm = w / scale
    quantiles = torch.from_numpy(O.NF4_QUANTILES if nf4 else O.F4E2M1_QUANTILES)
    centers = torch.from_numpy(O.CENTER_OF_NF4_QUANTILES if nf4 else O.CENTER_OF_F4E2M1_QUANTILES)
    idx = torch.searchsorted(centers, norm)
    if not nf4:
        shifted = torch.clip(idx + 1, 0, quantiles.numel() - 1)
        dist_left = torch.abs(norm - quantiles[idx])
        dist_right = torch.abs(norm - quantiles[shifted])
        right = (dist_right < dist_left) | ((dist_left == dist_right) & ((shifted + 1) % 2 == 0))
        idx = torch.where(right, shifted, idx)
    return quantiles[idx], scale
