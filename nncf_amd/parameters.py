"""Boundary data types of the lowering path, same names and semantics as the reference:
CompressWeightsMode (src/nncf/parameters.py:73-120), WeightCompressionConfig
(.../weight_compression/config.py:26-135), CompressedWeight (.../weight_compression/parameters.py:18-49)."""
from dataclasses import dataclass
from enum import Enum
from typing import Optional

import torch

from nncf_amd.errors import ValidationError


class CompressWeightsMode(Enum):
    INT8_SYM = "int8_sym"
    INT8_ASYM = "int8_asym"
    INT4_SYM = "int4_sym"
    INT4_ASYM = "int4_asym"
    NF4 = "nf4"
    MXFP4 = "mxfp4"
    FP4 = "fp4"


def _mode_of(mode) -> CompressWeightsMode:
    """Accepts this package's enum, the reference's enum (same .value strings) or a string."""
    if isinstance(mode, CompressWeightsMode):
        return mode
    value = getattr(mode, "value", mode)
    return CompressWeightsMode(value)


@dataclass
class WeightCompressionConfig:
    mode: CompressWeightsMode = CompressWeightsMode.INT8_ASYM
    group_size: int = -1

    def __post_init__(self) -> None:
        self.mode = _mode_of(self.mode)
        if self.group_size == 0 or self.group_size < -1:
            msg = f"Invalid group_size={self.group_size}. Group size must be a positive integer or -1."
            raise ValidationError(msg)

    @property
    def num_bits(self) -> int:
        return 8 if self.mode in (CompressWeightsMode.INT8_SYM, CompressWeightsMode.INT8_ASYM) else 4

    @property
    def is_asym_mode(self) -> bool:
        return self.mode in (CompressWeightsMode.INT4_ASYM, CompressWeightsMode.INT8_ASYM)

    @property
    def is_integer(self) -> bool:
        return self.mode not in (CompressWeightsMode.NF4, CompressWeightsMode.MXFP4, CompressWeightsMode.FP4)

    def __hash__(self) -> int:
        return hash((self.mode.value, self.group_size))


def as_config(config) -> WeightCompressionConfig:
    """Accepts this package's config or the reference's WeightCompressionConfig (duck-typed)."""
    if isinstance(config, WeightCompressionConfig):
        return config
    return WeightCompressionConfig(mode=_mode_of(config.mode), group_size=config.group_size)


@dataclass
class CompressedWeight:
    tensor: Optional[torch.Tensor] = None
    scale: Optional[torch.Tensor] = None
    zero_point: Optional[torch.Tensor] = None
    codebook: Optional[torch.Tensor] = None
    global_scale: Optional[torch.Tensor] = None

    def get_unscaled_tensor(self):
        if self.tensor is not None and self.codebook is not None:
            return self.codebook[self.tensor]
        return self.tensor
