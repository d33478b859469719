"""Weight decompressor modules with a fused HIP forward.

Mirrors `INT8AsymmetricWeightsDecompressor`, `INT8SymmetricWeightsDecompressor`,
`INT4AsymmetricWeightsDecompressor`, `INT4SymmetricWeightsDecompressor`
(src/nncf/torch/quantization/layers.py:1303-1460): same constructor arguments, buffers (`_scale` fp16, packed
`_zero_point`), `pack_weight`, `quantization_mode` and forward result -- the forward is ONE launch of
`nncf_amd_decompress` instead of unpack + cast + subtract + multiply + reshape + cast.  `NF4WeightsDecompressor`
and `MXFP4WeightsDecompressor` are this build's extension for the modes the reference only stores through OpenVINO
(openvino_backend.py:241-299): 4-bit code book indices / OCP e2m1 bit patterns, fp16 scale or E8M0 scale bytes.
"""
from __future__ import annotations

from functools import reduce
from operator import mul
from typing import Optional, Tuple

import torch

from nncf_amd import _lib
from nncf_amd.errors import InternalError
from nncf_amd.weight_lowering import pack_int4, pack_uint4

_OUT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}
_SCALE_F32, _SCALE_F16, _SCALE_E8M0 = 0, 1, 3


def decompress(codes: torch.Tensor, mode: str, n_groups: int, group_size: int, scale: torch.Tensor,
               zero_point: Optional[torch.Tensor], out_dtype: torch.dtype, scale_kind: Optional[int] = None) -> torch.Tensor:
    """Flat fused decompression: returns a 1-D tensor of n_groups * group_size elements of `out_dtype`."""
    _lib.require_gpu(codes, "decompress")
    if out_dtype not in _OUT:
        raise InternalError(f"decompress: unsupported result dtype {out_dtype}")
    if scale_kind is None:
        if scale.dtype == torch.float16:
            scale_kind = _SCALE_F16
        elif scale.dtype == torch.float32:
            scale_kind = _SCALE_F32
        elif scale.dtype == torch.uint8:
            scale_kind = _SCALE_E8M0
        else:
            raise InternalError(f"decompress: unsupported scale dtype {scale.dtype}")
    codes = codes.contiguous()
    scale = scale.contiguous()
    if scale.numel() != n_groups:
        raise InternalError(f"decompress: {scale.numel()} scales for {n_groups} groups")
    zp = zero_point.contiguous() if zero_point is not None else None
    out = torch.empty(n_groups * group_size, dtype=out_dtype, device=codes.device)
    with torch.cuda.device(codes.device):
        _lib.check(_lib.lib().nncf_amd_decompress(_lib.ptr(codes), _lib.MODE_ID[mode], n_groups, group_size, _lib.ptr(scale),
                                                  scale_kind, _lib.ptr(zp), _lib.ptr(out), _OUT[out_dtype],
                                                  _lib.stream_of(codes)))
    return out


class BaseWeightsDecompressor(torch.nn.Module):
    """layers.py:1270-1300."""

    @property
    def quantization_mode(self) -> str:
        raise NotImplementedError

    def pack_weight(self, weight: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError


def _groups(shape: Tuple[int, ...], scale: torch.Tensor) -> Tuple[int, int]:
    """(n_groups, group_size) of a weight of `shape` whose scale has one entry per group along ONE axis."""
    n = reduce(mul, shape, 1)
    n_groups = scale.numel()
    if n_groups == 0 or n % n_groups:
        raise InternalError(f"decompressor: scale of {n_groups} entries does not divide a weight of shape {shape}")
    gs = n // n_groups
    # groups must be runs of consecutive elements: the scale's only broadcast axis is the last one of the weight
    if tuple(scale.shape[:-1]) != tuple(shape[:-1]) or scale.shape[-1] != 1:
        raise InternalError(f"decompressor: scale shape {tuple(scale.shape)} is not [..., 1] over weight shape {shape}")
    return n_groups, gs


class INT8AsymmetricWeightsDecompressor(BaseWeightsDecompressor):
    """layers.py:1303-1337."""

    def __init__(self, scale: torch.Tensor, zero_point: torch.Tensor, result_dtype: Optional[torch.dtype] = None):
        super().__init__()
        self.register_buffer("_scale", scale.type(dtype=torch.float16))
        self.register_buffer("_zero_point", self.pack_weight(zero_point))
        self.result_dtype = result_dtype

    @property
    def quantization_mode(self) -> str:
        return "asymmetric"

    def pack_weight(self, weight: torch.Tensor) -> torch.Tensor:
        if torch.is_floating_point(weight):
            raise ValueError(f"Invalid weight dtype {weight.type}. Integer types are supported.")
        if torch.any((weight < 0) | (weight > 255)):
            raise ValueError("Weight values are not in [0, 255].")
        return weight.type(dtype=torch.uint8)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        n_groups, gs = _groups(tuple(x.shape), self._scale)
        dt = self.result_dtype if self.result_dtype is not None else self._scale.dtype
        return decompress(x, "int8_asym", n_groups, gs, self._scale, self._zero_point, dt).reshape(x.shape)


class INT8SymmetricWeightsDecompressor(BaseWeightsDecompressor):
    """layers.py:1340-1368."""

    def __init__(self, scale: torch.Tensor, result_dtype: Optional[torch.dtype] = None):
        super().__init__()
        self.register_buffer("_scale", scale.type(dtype=torch.float16))
        self.result_dtype = result_dtype

    @property
    def quantization_mode(self) -> str:
        return "symmetric"

    def pack_weight(self, weight: torch.Tensor) -> torch.Tensor:
        if torch.any((weight < -128) | (weight > 127)):
            raise ValueError("Weight values are not in [-128, 127].")
        return weight.type(dtype=torch.int8)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        n_groups, gs = _groups(tuple(x.shape), self._scale)
        dt = self.result_dtype if self.result_dtype is not None else self._scale.dtype
        return decompress(x, "int8_sym", n_groups, gs, self._scale, None, dt).reshape(x.shape)


class _FourBit(BaseWeightsDecompressor):
    _mode = ""

    def _run(self, x: torch.Tensor, scale: torch.Tensor, zp: Optional[torch.Tensor], scale_kind=None) -> torch.Tensor:
        n_groups, gs = _groups(tuple(self.compressed_weight_shape), scale)
        dt = self.result_dtype if self.result_dtype is not None else (scale.dtype if scale.is_floating_point() else torch.float16)
        out = decompress(x, self._mode, n_groups, gs, scale, zp, dt, scale_kind)
        return out.reshape(self.result_shape if self.result_shape is not None else self.compressed_weight_shape)


class INT4AsymmetricWeightsDecompressor(_FourBit):
    """layers.py:1371-1416."""
    _mode = "int4_asym"

    def __init__(self, scale: torch.Tensor, zero_point: torch.Tensor, compressed_weight_shape: Tuple[int, ...],
                 result_shape: Optional[Tuple[int, ...]] = None, result_dtype: Optional[torch.dtype] = None):
        super().__init__()
        self.register_buffer("_scale", scale.type(dtype=torch.float16))
        self.zero_point_shape = zero_point.shape
        self.register_buffer("_zero_point", self.pack_weight(zero_point))
        self.compressed_weight_shape = compressed_weight_shape
        self.result_shape = result_shape
        self.result_dtype = result_dtype

    @property
    def quantization_mode(self) -> str:
        return "asymmetric"

    def pack_weight(self, weight: torch.Tensor) -> torch.Tensor:
        if torch.any((weight < 0) | (weight > 15)):
            raise ValueError("Weight values are not in [0, 15].")
        return pack_uint4(weight.type(dtype=torch.uint8))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._run(x, self._scale, self._zero_point)


class INT4SymmetricWeightsDecompressor(_FourBit):
    """layers.py:1419-1460."""
    _mode = "int4_sym"

    def __init__(self, scale: torch.Tensor, compressed_weight_shape: Tuple[int, ...],
                 result_shape: Optional[Tuple[int, ...]] = None, result_dtype: Optional[torch.dtype] = None):
        super().__init__()
        self.register_buffer("_scale", scale.type(dtype=torch.float16))
        self.compressed_weight_shape = compressed_weight_shape
        self.result_shape = result_shape
        self.result_dtype = result_dtype

    @property
    def quantization_mode(self) -> str:
        return "symmetric"

    def pack_weight(self, weight: torch.Tensor) -> torch.Tensor:
        if torch.is_floating_point(weight):
            raise ValueError(f"Invalid weight dtype {weight.type}. Integer types are supported.")
        if torch.any((weight < -8) | (weight > 7)):
            raise ValueError("Tensor values are not in [-8, 7].")
        return pack_int4(weight.type(dtype=torch.int8))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._run(x, self._scale, None)


class NF4WeightsDecompressor(_FourBit):
    """Build extension (f1): x = nibble-packed NF4 indices, scale fp16 [..., groups, 1]."""
    _mode = "nf4"

    def __init__(self, scale: torch.Tensor, compressed_weight_shape: Tuple[int, ...],
                 result_shape: Optional[Tuple[int, ...]] = None, result_dtype: Optional[torch.dtype] = None):
        super().__init__()
        self.register_buffer("_scale", scale.type(dtype=torch.float16))
        self.compressed_weight_shape = compressed_weight_shape
        self.result_shape = result_shape
        self.result_dtype = result_dtype

    @property
    def quantization_mode(self) -> str:
        return "symmetric"

    def pack_weight(self, weight: torch.Tensor) -> torch.Tensor:
        if torch.any((weight < 0) | (weight > 15)):
            raise ValueError("Weight values are not in [0, 15].")
        return pack_uint4(weight.type(dtype=torch.uint8))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._run(x, self._scale, None)


class MXFP4WeightsDecompressor(_FourBit):
    """Build extension (f1): x = nibble-packed OCP e2m1 codes, scale = E8M0 bytes (uint8, 2^(b-127)) [..., groups, 1]."""
    _mode = "mxfp4"

    def __init__(self, scale_e8m0: torch.Tensor, compressed_weight_shape: Tuple[int, ...],
                 result_shape: Optional[Tuple[int, ...]] = None, result_dtype: Optional[torch.dtype] = None):
        super().__init__()
        if scale_e8m0.dtype != torch.uint8:
            raise ValueError("MXFP4 scales are E8M0 bytes (torch.uint8)")
        self.register_buffer("_scale", scale_e8m0)
        self.compressed_weight_shape = compressed_weight_shape
        self.result_shape = result_shape
        self.result_dtype = result_dtype

    @property
    def quantization_mode(self) -> str:
        return "symmetric"

    def pack_weight(self, weight: torch.Tensor) -> torch.Tensor:
        if torch.any((weight < 0) | (weight > 15)):
            raise ValueError("Weight values are not in [0, 15].")
        return pack_uint4(weight.type(dtype=torch.uint8))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._run(x, self._scale, None, _SCALE_E8M0)
