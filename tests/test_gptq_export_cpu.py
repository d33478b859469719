"""CPU checks of the oracle's GPTQ-layout restatement (oracle/wc_oracle.py: gptq_pack8_ref / gptq_dequant8_ref):
the re-derived codes are the stored codes, the words are the little-endian byte image of 4 consecutive codes along
the input axis, and pack -> dequant gives back the decompressed weight."""
import numpy as np
import pytest

from oracle import wc_oracle as O


@pytest.mark.parametrize("shape", [(8, 16), (12, 4), (64, 128)])
def test_pack_is_the_byte_image_and_round_trips(shape):
    rng = np.random.default_rng(shape[0])
    out_f, in_f = shape
    codes = rng.integers(0, 256, shape, dtype=np.uint8)
    zero = rng.integers(0, 256, (out_f,), dtype=np.uint8)
    scale = (rng.random((out_f,), dtype=np.float32) * 0.01 + 1e-4).astype(np.float16)
    w = (codes.astype(np.float32) - zero.astype(np.float32)[:, None]) * scale.astype(np.float32)[:, None]
    qw, qz = O.gptq_pack8_ref(w, scale, zero.astype(np.float32))
    assert qw.shape == (in_f // 4, out_f) and qz.shape == (1, out_f // 4)
    assert np.array_equal(qw.view(np.uint32), np.ascontiguousarray(codes).view(np.uint32).T)
    assert np.array_equal(qz.view(np.uint32).reshape(-1), zero.view(np.uint32))
    dq = O.gptq_dequant8_ref(qw, qz, scale)
    want = ((codes.astype(np.int32) - zero.astype(np.int32)[:, None]).astype(np.float16) * scale[:, None]).astype(np.float16)
    assert np.array_equal(dq.view(np.uint16), want.view(np.uint16))
