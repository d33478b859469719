"""CPU-only checks of the boundary: the shared library loads, exports exactly the symbols include/nncf_amd_wc.h
declares, and argument validation answers with the reference's error conventions without touching a GPU."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "nncf_amd_wc.h")


def _declared():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nncf_amd_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from nncf_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run python -m nncf_amd.build"
    declared = _declared()
    assert len(declared) >= 18
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = sorted(set(re.findall(r" T (nncf_amd_[a-z0-9_]+)", out)))
    assert exported == declared, (set(declared) ^ set(exported))
    assert sorted(_lib.SIGNATURES) == declared  # the ctypes table binds each of them
    L = _lib.lib()
    assert L.nncf_amd_version() >= 100
    assert L.nncf_amd_status_string(-2).decode().startswith("channel size")


def test_argument_validation_without_gpu():
    from nncf_amd import _lib, errors
    L = _lib.lib()
    # InvalidGroupSizeError message of weight_lowering.py:72-74
    st = L.nncf_amd_int_quantize(None, 0, 2, 10, 1, 3, 3, None, None, 0, None, None, None, None, None, None)
    assert st == _lib.ERR_GROUP_SIZE and "Channel size 10 should be divisible by size of group 3." == _lib.last_error()
    with pytest.raises(errors.InvalidGroupSizeError):
        _lib.check(st)
    assert L.nncf_amd_int_quantize(None, 0, 2, 10, 1, 5, 9, None, None, 0, None, None, None, None, None, None) == _lib.ERR_MODE
    assert L.nncf_amd_int_quantize(None, 0, 0, 10, 1, 5, 3, None, None, 0, None, None, None, None, None, None) == 0  # empty
    assert L.nncf_amd_int_quantize(None, 0, 2, 10, 1, 5, 3, None, None, 0, None, None, None, None, None, None) == _lib.ERR_BAD_ARG
    assert L.nncf_amd_float_quantize(None, 0, 2, 10, 1, 5, 3, None, None, None, None, None, None, None, None) == _lib.ERR_MODE
    assert L.nncf_amd_pack_u4(None, 3, None, None) == _lib.ERR_BAD_ARG
    assert L.nncf_amd_pack_u4(None, 0, None, None) == 0
    assert L.nncf_amd_gptq_block(None, None, None, None, 1, 1, 0, 1, 1, 3, None, None, 0, None) == _lib.ERR_BAD_ARG
    assert L.nncf_amd_hessian_accumulate(None, None, 0, 1, 1, 1.0, 1.0, None) == _lib.ERR_BAD_ARG
    assert L.nncf_amd_cholinv_block(None, 1, None, 1, 1, None, None) == _lib.ERR_BAD_ARG
    assert L.nncf_amd_reduce_workspace(1, 2048, 4096) > 0
    assert L.nncf_amd_reduce_activation_stats(None, 0, 1, 1, 1, None, None, None, 0, None) == _lib.ERR_BAD_ARG


def test_host_mirror_rejects_cpu_tensors_and_bad_configs():
    import torch

    from nncf_amd import errors
    from nncf_amd import weight_lowering as wl
    from nncf_amd.parameters import CompressWeightsMode, WeightCompressionConfig
    w = torch.randn(4, 32)
    with pytest.raises(errors.UnsupportedBackendError):
        wl.compress_weight(w, (1,), WeightCompressionConfig(CompressWeightsMode.INT4_SYM, 16))
    with pytest.raises(errors.ValidationError):
        WeightCompressionConfig(CompressWeightsMode.INT4_SYM, 0)
    with pytest.raises(errors.InvalidGroupSizeError):
        wl.reshape_weight_for_grouped_quantization(w, 1, 5)
    with pytest.raises(errors.UnsupportedModelError):
        wl.reshape_weight_for_grouped_quantization(w, (0, 1), 4)
    cfg = WeightCompressionConfig("int8_asym")
    assert cfg.num_bits == 8 and cfg.is_asym_mode and cfg.is_integer and cfg.group_size == -1
    assert not WeightCompressionConfig("mxfp4", 32).is_integer
    r, ax = wl.reshape_weight_for_grouped_quantization(w, (1,), 8)
    assert tuple(r.shape) == (4, 4, 8) and ax == 2
    assert tuple(wl.ungroup_weights(r, 1).shape) == (4, 32)


def test_torch_chain_matches_numpy_oracle():
    import numpy as np
    import torch

    from oracle import wc_oracle as O
    from oracle.torch_chain import compress_int_torch
    rng = np.random.default_rng(0)
    wt = torch.from_numpy((rng.standard_normal((64, 512)) * 0.02).astype(np.float32)).to(torch.bfloat16)
    w32 = wt.float().numpy()
    for mode in O.INT_MODES:
        for gs in (-1, 128):
            c, s, z = compress_int_torch(wt, mode, gs, pack=True)
            rc, rs, rz = O.compress_weight(w32, (1,), mode, gs)
            if mode.startswith("int4"):
                rc = O.pack_int4(rc) if mode == "int4_sym" else O.pack_uint4(rc)
            assert np.array_equal(c.numpy(), rc) and np.array_equal(s.numpy(), rs)
            if rz is not None:
                assert np.array_equal(z.numpy(), rz)
