"""GPU parity of the GPTQ-layout export (nncf_amd/gptq_export.py, csrc/wc_export.hip) against the oracle's
restatement of the third-party pack loop, and against what the reference's own test checks: the converted layer
computes the same linear as the INT8 decompressor (tests/integration/gptq_model/test_converter.py:25-83).
The layout itself is parity-unpinned (gptqmodel is absent, the reference holds no packed tensors)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F
from torch import nn

from oracle import wc_oracle as O

pytestmark = pytest.mark.gpu


def _decompressor(is_sym, scale, zp=None):
    from nncf_amd import decompressors as D
    if is_sym:
        return D.INT8SymmetricWeightsDecompressor(scale=scale).cuda()
    return D.INT8AsymmetricWeightsDecompressor(scale=scale, zero_point=zp).cuda()


@pytest.mark.parametrize("shape", [(128, 128), (64, 4), (4, 64), (200, 328), (1028, 516)])
@pytest.mark.parametrize("is_sym", [True, False], ids=["sym", "asym"])
def test_pack_and_dequant_match_the_oracle(shape, is_sym):
    from nncf_amd import gptq_export as E
    g = torch.Generator().manual_seed(shape[0] * 7 + shape[1])
    out_f, in_f = shape
    scale = (torch.rand((out_f, 1), generator=g) * 0.01 + 1e-4).to(torch.float16)
    if is_sym:
        codes = torch.randint(-128, 128, shape, generator=g, dtype=torch.int8)
        zero = np.full((out_f,), 128.0, dtype=np.float32)
        w_dq = codes.to(torch.float32) * scale.to(torch.float32)
    else:
        codes = torch.randint(0, 256, shape, generator=g, dtype=torch.uint8)
        zp = torch.randint(0, 256, (out_f, 1), generator=g, dtype=torch.uint8)
        zero = zp.reshape(-1).numpy().astype(np.float32)
        w_dq = (codes.to(torch.float32) - zp.to(torch.float32)) * scale.to(torch.float32)
    # the oracle re-derives the codes from the decompressed weight, as the third-party pack does (fp32 here so that
    # the re-derivation is exact; the HIP path never leaves the integer codes)
    qw_ref, qz_ref = O.gptq_pack8_ref(w_dq.numpy(), scale.numpy(), zero)
    qw = E.gptq_pack8(codes.cuda())
    assert qw.dtype == torch.int32 and tuple(qw.shape) == (in_f // 4, out_f)
    assert np.array_equal(qw.cpu().numpy(), qw_ref)
    qz = torch.from_numpy(zero.astype(np.uint8)).cuda().view(torch.int32).reshape(1, -1)
    assert np.array_equal(qz.cpu().numpy(), qz_ref)
    sc = scale.reshape(1, -1).cuda()
    w_ref = O.gptq_dequant8_ref(qw_ref, qz_ref, scale.numpy())
    got = E.gptq_dequant8(qw, qz, sc, torch.float16)
    assert np.array_equal(got.cpu().numpy().view(np.uint16), w_ref.view(np.uint16))
    for dt in (torch.float32, torch.bfloat16):
        assert torch.equal(E.gptq_dequant8(qw, qz, sc, dt), got.to(dt))


@torch.no_grad()
@pytest.mark.parametrize("with_bias", [True, False], ids=["with_bias", "without_bias"])
@pytest.mark.parametrize("is_sym", [True, False], ids=["sym", "asym"])
def test_convert_int8_linear(with_bias, is_sym):
    """test_converter.py:25-57 with the reference's inputs (seed 42); here the two results are bit-equal."""
    from nncf_amd.gptq_export import GPTQQuantLinear, convert_linear_module
    torch.manual_seed(42)
    example_input = torch.randn(1, 128, dtype=torch.float16)
    scale = torch.rand((128, 1), dtype=torch.float16) * 0.01
    compressed_weight = torch.randint(0, 255, (128, 128), dtype=torch.uint8)
    bias = torch.randn((128,), dtype=torch.float16) if with_bias else None
    if is_sym:
        compressed_weight = (compressed_weight.to(torch.int16) - 128).to(torch.int8)
        decompressor = _decompressor(True, scale)
    else:
        zp = torch.randint(100, 140, (128, 1), dtype=torch.uint8)
        decompressor = _decompressor(False, scale, zp)
    compressed_weight = compressed_weight.cuda()
    bias_d = bias.cuda() if with_bias else None
    result = F.linear(example_input.cuda(), decompressor(compressed_weight), bias_d)

    linear = nn.Linear(128, 128, bias=with_bias)
    linear.weight = nn.Parameter(compressed_weight, requires_grad=False)
    if with_bias:
        linear.bias = nn.Parameter(bias_d, requires_grad=False)
    gptq_module = convert_linear_module(linear, decompressor)
    assert isinstance(gptq_module, GPTQQuantLinear) and gptq_module.sym == is_sym
    assert gptq_module.qweight.shape == (32, 128) and gptq_module.qzeros.shape == (1, 32)
    assert gptq_module.scales.shape == (1, 128) and not gptq_module.g_idx.any()
    gptq_result = gptq_module(example_input.cuda())
    assert torch.equal(result, gptq_result)


class _LinearModel(nn.Module):
    def __init__(self):
        super().__init__()
        self.linear = nn.Linear(128, 128)
        self.block = nn.Sequential(nn.Linear(128, 64), nn.ReLU(), nn.Linear(64, 128, bias=False))

    def forward(self, x):
        return self.block(self.linear(x))


@torch.no_grad()
@pytest.mark.parametrize("mode", ["int8_sym", "int8_asym"])
def test_convert_model(mode):
    """test_converter.py:69-83 + the outputs before and after conversion."""
    import nncf_amd
    from nncf_amd.gptq_export import GPTQQuantLinear, convert_model
    from torch.nn.utils import parametrize
    torch.manual_seed(0)
    example_input = torch.randn(3, 128).cuda()
    model = _LinearModel().cuda()
    q_model = nncf_amd.compress_weights(model, mode=mode, dataset=nncf_amd.Dataset([example_input]))
    before = q_model(example_input)
    convert_model(q_model)
    for mod in (q_model.linear, q_model.block[0], q_model.block[2]):
        assert isinstance(mod, GPTQQuantLinear) and not isinstance(mod, nn.Linear)
    assert not any(parametrize.is_parametrized(m) for m in q_model.modules())
    after = q_model(example_input)
    # the decompressor rounds to fp16 and casts to the model's fp32, as the converted layer does
    assert torch.equal(before, after)
    assert set(q_model.state_dict()) >= {"linear.qweight", "linear.qzeros", "linear.scales", "linear.g_idx", "linear.bias"}


def test_convert_errors():
    import nncf_amd
    from nncf_amd.errors import InternalError
    from nncf_amd.gptq_export import convert_linear_module, convert_model, gptq_pack8
    with pytest.raises(InternalError, match="Expected the model returned by"):
        convert_model(_LinearModel().cuda())
    model = nn.Sequential(nn.Linear(128, 128)).cuda()
    q = nncf_amd.compress_weights(model, mode="int4_sym", group_size=32, all_layers=True,
                                  dataset=nncf_amd.Dataset([torch.randn(1, 128).cuda()]))
    with pytest.raises(InternalError, match="Unsupported decompressor module type"):
        convert_model(q)
    with pytest.raises(InternalError, match="Unsupported decompressor module type"):
        convert_linear_module(nn.Linear(8, 8), nn.Identity())
    with pytest.raises(InternalError):  # in_features not a multiple of 4
        gptq_pack8(torch.zeros((8, 6), dtype=torch.uint8, device="cuda"))
    assert gptq_pack8(torch.zeros((0, 8), dtype=torch.uint8, device="cuda")).shape == (2, 0)
