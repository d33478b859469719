"""Export of INT8-compressed Linear layers to the GPTQ v2 tensor layout (optional export).

Mirror of src/nncf/experimental/torch/gptqmodel/convertor.py (`convert_linear_module` :35-84, `convert_model` :87-132).
The reference decompresses every weight, moves it to the CPU, lets gptqmodel's `TritonV2QuantLinear.pack` re-derive
the integer codes and assemble the int32 words in numpy, and moves the module back.  gptqmodel is not part of this
build (third-party, absent from the image): `GPTQQuantLinear` holds the same tensors under the same names
(`qweight`, `qzeros`, `scales`, `g_idx`, `bias`; bits = 8, group_size = -1, desc_act = False, v2 zero points), the
codes never leave HBM (`nncf_amd_gptq_pack8`: one 32-bit transpose), and its forward dequantizes with
`nncf_amd_gptq_dequant8`.  Parity of the LAYOUT with gptqmodel is unpinned (nothing in the reference holds packed
tensors); what the reference's own test checks -- the converted layer computes the same linear
(tests/integration/gptq_model/test_converter.py:25-57) -- is what the tests here check, bit-exactly.

Symmetric layers, a DELIBERATE and unpinned choice: INT8_SYM codes are stored as code + 128 with qzeros = 128 (the
GPTQ convention for signed codes in unsigned words), where the reference's convertor hands `pack` a zero point of 0.
The forward is numerically the decompressor's linear either way, but the stored qweight / qzeros words of a symmetric
layer must not be assumed interchangeable with a file gptqmodel wrote from the reference's convertor.
"""
from typing import Union

import torch
import torch.nn.functional as F
from torch import nn
from torch.nn.utils import parametrize

from nncf_amd import _lib
from nncf_amd.decompressors import INT8AsymmetricWeightsDecompressor, INT8SymmetricWeightsDecompressor
from nncf_amd.errors import InternalError

T_INT8_DECOMPRESSOR = Union[INT8AsymmetricWeightsDecompressor, INT8SymmetricWeightsDecompressor]
_OUT = {torch.float32: _lib.F32, torch.float16: _lib.F16, torch.bfloat16: _lib.BF16}


def gptq_pack8(codes: torch.Tensor) -> torch.Tensor:
    """[out, in] uint8 / int8 codes -> qweight [in / 4, out] int32 (int8 codes are stored as code + 128)."""
    _lib.require_gpu(codes, "gptq_pack8")
    if codes.dtype not in (torch.uint8, torch.int8) or codes.dim() != 2:
        raise InternalError(f"gptq_pack8: expected 2-D uint8 / int8 codes, got {codes.dtype} {tuple(codes.shape)}")
    codes = codes.contiguous()
    out_f, in_f = codes.shape
    qweight = torch.empty((in_f // 4, out_f), dtype=torch.int32, device=codes.device)
    with torch.cuda.device(codes.device):
        _lib.check(_lib.lib().nncf_amd_gptq_pack8(_lib.ptr(codes), int(codes.dtype == torch.int8), out_f, in_f,
                                                  _lib.ptr(qweight), _lib.stream_of(codes)))
    return qweight


def gptq_dequant8(qweight: torch.Tensor, qzeros: torch.Tensor, scales: torch.Tensor,
                  out_dtype: torch.dtype = torch.float16) -> torch.Tensor:
    """qweight [in / 4, out] int32, qzeros [1, out / 4] int32, scales [1, out] fp16 -> weight [out, in]."""
    _lib.require_gpu(qweight, "gptq_dequant8")
    if out_dtype not in _OUT:
        raise InternalError(f"gptq_dequant8: unsupported result dtype {out_dtype}")
    in4, out_f = qweight.shape
    if qweight.dtype != torch.int32 or qzeros.dtype != torch.int32 or scales.dtype != torch.float16 \
            or qzeros.numel() * 4 != out_f or scales.numel() != out_f:
        raise InternalError("gptq_dequant8: tensors are not a bits = 8, group_size = -1 GPTQ layer")
    weight = torch.empty((out_f, in4 * 4), dtype=out_dtype, device=qweight.device)
    with torch.cuda.device(qweight.device):
        _lib.check(_lib.lib().nncf_amd_gptq_dequant8(_lib.ptr(qweight.contiguous()), _lib.ptr(qzeros.contiguous()),
                                                     _lib.ptr(scales.contiguous()), in4 * 4, out_f, _lib.ptr(weight),
                                                     _OUT[out_dtype], _lib.stream_of(qweight)))
    return weight


class GPTQQuantLinear(nn.Module):
    """The tensors of a gptqmodel quantized linear (bits = 8, group_size = -1, sym as given, desc_act = False)."""

    bits = 8
    group_size = -1
    desc_act = False

    def __init__(self, in_features: int, out_features: int, sym: bool, bias: bool, device=None):
        super().__init__()
        self.in_features, self.out_features, self.sym = in_features, out_features, sym
        self.register_buffer("qweight", torch.zeros((in_features // 4, out_features), dtype=torch.int32, device=device))
        self.register_buffer("qzeros", torch.zeros((1, out_features // 4), dtype=torch.int32, device=device))
        self.register_buffer("scales", torch.zeros((1, out_features), dtype=torch.float16, device=device))
        self.register_buffer("g_idx", torch.zeros((in_features,), dtype=torch.int32, device=device))
        self.register_buffer("bias", torch.zeros((out_features,), dtype=torch.float16, device=device) if bias else None)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        w = gptq_dequant8(self.qweight, self.qzeros, self.scales, x.dtype)
        return F.linear(x, w, None if self.bias is None else self.bias.to(x.dtype))


def _codes_of(linear: nn.Linear) -> torch.Tensor:
    if parametrize.is_parametrized(linear, "weight"):
        return linear.parametrizations.weight.original.data
    return linear.weight.data


def convert_linear_module(linear: nn.Linear, decompressor: T_INT8_DECOMPRESSOR) -> GPTQQuantLinear:
    """convertor.py:35-84: the compressed INT8 weight of `linear` in the GPTQ layout."""
    if isinstance(decompressor, INT8AsymmetricWeightsDecompressor):
        is_symmetric = False
    elif isinstance(decompressor, INT8SymmetricWeightsDecompressor):
        is_symmetric = True
    else:
        msg = f"Unsupported decompressor module type: {type(decompressor)}"
        raise InternalError(msg)
    codes = _codes_of(linear)
    _lib.require_gpu(codes, "convert_linear_module")
    out_f, in_f = linear.out_features, linear.in_features
    if tuple(codes.shape) != (out_f, in_f) or decompressor._scale.numel() != out_f:
        msg = f"Expected a per-channel INT8 weight of shape {(out_f, in_f)}, got codes {tuple(codes.shape)} " \
              f"with {decompressor._scale.numel()} scales"
        raise InternalError(msg)
    if is_symmetric:
        codes = codes.to(torch.int8) if codes.dtype != torch.int8 else codes
        zeros = torch.full((out_f,), 128, dtype=torch.uint8, device=codes.device)
    else:
        zeros = decompressor._zero_point.reshape(-1).to(torch.uint8)
    module = GPTQQuantLinear(in_f, out_f, is_symmetric, linear.bias is not None, device=codes.device)
    module.qweight = gptq_pack8(codes)
    module.qzeros = zeros.contiguous().view(torch.int32).reshape(1, out_f // 4)
    module.scales = decompressor._scale.reshape(1, out_f).to(torch.float16).contiguous()
    if linear.bias is not None:
        module.bias = linear.bias.detach().clone()
    return module


def convert_model(model: nn.Module) -> nn.Module:
    """convertor.py:87-132: every compressed Linear of a model returned by `nncf_amd.compress_weights` becomes a
    `GPTQQuantLinear`; nothing of the compression wrapper (parametrizations) is left on the model."""
    targets = [(name, mod) for name, mod in model.named_modules() if parametrize.is_parametrized(mod, "weight")]
    if not targets:
        msg = "Expected the model returned by nncf_amd.compress_weights()"
        raise InternalError(msg)
    for name, mod in targets:  # validate everything before the first replacement
        hook = mod.parametrizations.weight[0]
        if not isinstance(hook, (INT8AsymmetricWeightsDecompressor, INT8SymmetricWeightsDecompressor)):
            msg = f"Unsupported decompressor module type {type(hook)}, module {name!r}"
            raise InternalError(msg)
        if not isinstance(mod, nn.Linear):
            msg = f"Unsupported module type {type(mod).__name__}, module {name!r}"
            raise InternalError(msg)
    for name, mod in targets:
        new_linear = convert_linear_module(mod, mod.parametrizations.weight[0])
        if name.count(".") == 0:
            setattr(model, name, new_linear)
        else:
            parent_name, child_name = name.rsplit(".", 1)
            setattr(model.get_submodule(parent_name), child_name, new_linear)
    return model
