"""Helpers for the -m gpu parity tests: numpy <-> device tensors in the three weight dtypes."""
import numpy as np
import torch

TORCH_DT = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}


def to_dev(w32: np.ndarray, dtype: str) -> torch.Tensor:
    """float32 numpy values -> device tensor of `dtype` (values must already be representable for bit parity)."""
    return torch.from_numpy(np.ascontiguousarray(w32)).to("cuda").to(TORCH_DT[dtype])


def representable(w32: np.ndarray, dtype: str) -> np.ndarray:
    """Rounds float32 values to `dtype` and returns them as float32 (the oracle's input)."""
    t = torch.from_numpy(np.ascontiguousarray(w32)).to(TORCH_DT[dtype])
    return t.to(torch.float32).numpy()


def host(t):
    if t is None:
        return None
    if t.dtype == torch.bfloat16:
        t = t.float()
    return t.detach().cpu().numpy()
