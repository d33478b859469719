#!/usr/bin/env python3
"""Developer bench of the non-RTN rows on one MI355X: reducers (HBM), Hessian SYRK (MFMA), GPTQ layer, QDQ loop."""
import json
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nncf_amd import gptq, statistics as S, weight_lowering as wl  # noqa: E402
from nncf_amd.parameters import WeightCompressionConfig  # noqa: E402


def timeit(fn, reps=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    out = {}
    dev = "cuda"
    # reducers: one calibration sample [1, 2048, 4096] bf16 .. and a large batch to leave the caches
    for shape in ((1, 2048, 4096), (64, 2048, 4096)):
        x = torch.randn(shape, device=dev, dtype=torch.bfloat16)
        t = timeit(lambda: S.activation_stats(x))
        out[f"reduce_{shape}"] = {"us": t * 1e6, "GBps": x.numel() * 2 / t / 1e9}
        t = timeit(lambda: (x.float().mean((0, 1)), x.float().var(1, correction=0).mean(), x.abs().amax(1).float().mean()))
        out[f"reduce_torch_{shape}"] = {"us": t * 1e6, "GBps": x.numel() * 2 / t / 1e9}
    # hessian
    for K, T, dt in ((4096, 16384, torch.float32), (4096, 16384, torch.bfloat16), (8192, 8192, torch.bfloat16)):
        x = torch.randn((1, T, K), device=dev, dtype=dt)
        t = timeit(lambda: gptq.calculate_hessian([x]), reps=5)
        out[f"hessian_K{K}_T{T}_{str(dt)[6:]}"] = {"ms": t * 1e3, "TFLOPs_full": 2 * T * K * K / t / 1e12,
                                                     "TFLOPs_sym": T * K * K / t / 1e12}
        xf = x[0].float()
        t = timeit(lambda: xf.T @ xf, reps=5)
        out[f"hessian_torch_K{K}_T{T}"] = {"ms": t * 1e3, "TFLOPs_full": 2 * T * K * K / t / 1e12}
    # GPTQ layer 4096x4096, g128
    K = N = 4096
    x = torch.randn((1, 8192, K), device=dev)
    H = gptq.calculate_hessian([x])[0]
    W = torch.randn((N, K), device=dev) * 0.02
    cfg = WeightCompressionConfig("int4_sym", 128)
    t = timeit(lambda: gptq.quantize_weights(H.clone(), W.clone(), cfg), reps=2, warm=1)
    out["gptq_quantize_4096x4096_g128"] = {"ms": t * 1e3}
    # QDQ (AWQ/SE inner loop): fp32 weight, precomputed scale
    w = torch.randn((4096, 32, 128), device=dev) * 0.02
    sc = w.abs().amax(-1, keepdim=True) / 8
    t = timeit(lambda: wl.integer_quantize_dequantize_weight(w, cfg, None, sc))
    out["qdq_fp32_4096x4096_g128"] = {"us": t * 1e6, "GBps_alg": w.numel() * 8 / t / 1e9}
    # decompressor forward (f2): packed int4 -> bf16, 14336 x 4096 g128
    from nncf_amd import decompressors as D
    wb = (torch.randn((14336, 4096), device=dev) * 0.02).to(torch.bfloat16)
    cw = wl.compress_weight(wb, (1,), cfg)
    fused = wl.compress_weight_packed(wb, (1,), cfg)
    dec = D.INT4SymmetricWeightsDecompressor(cw.scale, tuple(cw.tensor.shape), (14336, 4096), torch.bfloat16).cuda()
    t = timeit(lambda: dec(fused.tensor))
    n = wb.numel()
    out["decompress_int4sym_14336x4096_bf16"] = {"us": t * 1e6, "GBps_alg": n * (0.5 + 2 + 2 / 128) / t / 1e9}
    from nncf_amd.weight_lowering import unpack_int4
    s16 = cw.scale.half()
    t = timeit(lambda: (unpack_int4(fused.tensor).reshape(cw.tensor.shape).half() * s16).reshape(14336, 4096).to(torch.bfloat16))
    out["decompress_torch_chain"] = {"us": t * 1e6}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
