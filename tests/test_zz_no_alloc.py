"""nncf_amd_int_quant_error_static: the error metric with its per-channel partial sums in the library's own device
array (no caller workspace, no allocation), through the C ABI, against nncf_amd_int_quant_error (the existing entry
point with a caller workspace; same kernels, so the bar is bit equality) on two calls in a row, the second with a
smaller input, and against the oracle.  Collected last on purpose (file name)."""
import numpy as np
import pytest
import torch

from oracle import wc_oracle as O

pytestmark = pytest.mark.gpu


def _both(w, gs, mode, reduction):
    from nncf_amd import _lib
    L = _lib.lib()
    outer, r = w.shape
    res_ws = torch.full((1,), -1.0, device="cuda")
    res_st = torch.full((1,), -2.0, device="cuda")
    ws = torch.empty(int(L.nncf_amd_int_quant_error_workspace(outer, 1)), dtype=torch.uint8, device="cuda")
    red = {"max_mean": 0, "frobenius": 1}[reduction]
    _lib.check(L.nncf_amd_int_quant_error(_lib.ptr(w), _lib.dtype_id(w), outer, r, 1, gs, _lib.MODE_ID[mode], red,
                                          _lib.ptr(res_ws), _lib.ptr(ws), ws.numel(), _lib.stream_of(w)))
    def static_call():
        _lib.check(L.nncf_amd_int_quant_error_static(_lib.ptr(w), _lib.dtype_id(w), outer, r, 1, gs, _lib.MODE_ID[mode],
                                                     red, _lib.ptr(res_st), _lib.stream_of(w)))

    static_call()  # warm-up: lazy code-object / symbol loading happens here, not in the measured call
    torch.cuda.synchronize()
    # torch's counter only sees torch's caching allocator; a hipMalloc inside the library would show in the DRIVER's
    # free-memory figure (hipMemGetInfo), so both are compared around the call
    before, free_before = torch.cuda.memory_allocated(), torch.cuda.mem_get_info()[0]
    static_call()
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before
    assert torch.cuda.mem_get_info()[0] == free_before
    return float(res_ws.item()), float(res_st.item())


@pytest.mark.parametrize("reduction", ["max_mean", "frobenius"])
@pytest.mark.parametrize("mode,gs", [("int8_asym", -1), ("int4_sym", 64)])
def test_static_error_metric_two_calls_second_smaller(mode, gs, reduction):
    torch.manual_seed(7)
    big = (torch.randn(3000, 512, device="cuda") * 0.05).to(torch.bfloat16)
    small = (torch.randn(40, 256, device="cuda") * 0.3).to(torch.float32)
    small[3] = 0.0
    for w in (big, small):  # the second call sees the first call's leftovers in the library's array
        a, b = _both(w, gs, mode, reduction)
        assert a == b, (a, b)
        ref = O.get_integer_quantization_error(w.float().cpu().numpy(), (1,), mode, gs, reduction)
        # fp32 sums in another order than numpy's: the reference's own alignment test takes rtol 1e-4 here
        # (tests/openvino/optimized_functions/test_compression_functions.py:284-324)
        np.testing.assert_allclose(b, ref, rtol=1e-4)


def test_static_error_metric_rejects_beyond_its_maximum():
    from nncf_amd import _lib
    L = _lib.lib()
    w = torch.zeros(8, 8, device="cuda")
    res = torch.zeros(1, device="cuda")
    st = L.nncf_amd_int_quant_error_static(_lib.ptr(w), 0, 262145, 8, 1, -1, 0, 0, _lib.ptr(res), None)
    assert st == _lib.ERR_WORKSPACE and "262144" in _lib.last_error()
    st = L.nncf_amd_int_quant_error_static(_lib.ptr(w), 0, 1 << 40, 8, 1 << 40, -1, 0, 0, _lib.ptr(res), None)
    assert st == _lib.ERR_WORKSPACE
