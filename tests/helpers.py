"""Shared helpers for the test-suite (fixture decoding, dtype plumbing)."""
import numpy as np


def bf16_bits_to_f32(bits: np.ndarray) -> np.ndarray:
    return (bits.astype(np.uint32) << 16).view(np.float32)


def golden_weight(g: dict, dtype: str) -> np.ndarray:
    """The fixture's input weight as float32 values exactly representable in `dtype`."""
    if dtype == "bf16":
        return bf16_bits_to_f32(g["w_bf16_bits"])
    return g[f"w_{dtype}"].astype(np.float32)


def assert_equal_nan(a, b, msg=""):
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, f"{msg} shape {a.shape} vs {b.shape}"
    ok = (a == b) | (np.isnan(a) & np.isnan(b)) if a.dtype.kind == "f" else (a == b)
    assert ok.all(), f"{msg}: {(~ok).sum()} mismatches, first at {np.argwhere(~ok)[:5].tolist()}"
