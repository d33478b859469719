"""Model-level parity (BASELINE configs[0] and [2]): nncf_amd.compress_weights on a torch model
on the GPU vs what the REFERENCE nncf.compress_weights stored for the same model (tests/golden/model_*.npz/json,
written by tools/gen_golden_model.py in the build container).

Bar: data-free configurations -- which weights are compressed, to what precision, packed codes, fp16 scales and
packed zero points all bit-exact.  Data-aware configurations (statistics come from a GPU forward here and a CPU
forward in the fixture, so activations differ in the last bits): the precision assignment must be identical, scales
within one fp16 ulp on >= 97 % of the groups, codes equal on >= 97 % of the elements, logits close."""
import hashlib
import json
import math
import os

import numpy as np
import pytest
import torch

from tests.tiny_models import TINY_CONFIGS, make_tiny_lm, tiny_lm_dataset

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _amd_kwargs(kw):
    from nncf_amd.parameters import AdvancedCompressionParameters, GroupSizeFallbackMode, IgnoredScope

    kw = dict(kw)
    data = kw.pop("data", False)
    fallback = kw.pop("fallback", None)
    ignored = kw.pop("ignored", None)
    if fallback is not None:
        kw["advanced_parameters"] = AdvancedCompressionParameters(
            group_size_fallback_mode=GroupSizeFallbackMode(fallback), min_adjusted_group_size=16)
    if ignored is not None:
        kw["ignored_scope"] = IgnoredScope(names=tuple(ignored))
    return kw, data


def _harvest(model):
    """{module name: {"w", "s", "z"}} of every parametrized (compressed) weight; tied modules reported once."""
    from torch.nn.utils import parametrize

    per, seen = {}, set()
    for name, mod in model.named_modules():
        if parametrize.is_parametrized(mod, "weight"):
            plist = mod.parametrizations.weight
            if id(plist.original) in seen:
                continue
            seen.add(id(plist.original))
            d = {"w": plist.original.detach().cpu().numpy(), "s": plist[0]._scale.cpu().numpy()}
            if hasattr(plist[0], "_zero_point"):
                d["z"] = plist[0]._zero_point.cpu().numpy()
            per[name] = d
    return per


@pytest.fixture(scope="module")
def tiny_golden():
    z = np.load(os.path.join(GOLDEN, "model_tiny.npz"))
    meta = json.loads(bytes(z["__meta__"]).decode())
    return z, meta


def _run_tiny(name):
    import nncf_amd

    mkw, kw = TINY_CONFIGS[name]
    kw, data = _amd_kwargs(kw)
    model = make_tiny_lm(**mkw).cuda()
    samples = tiny_lm_dataset()
    ds = samples if data else samples[:1]
    nncf_amd.compress_weights(model, dataset=nncf_amd.Dataset(ds), **kw)
    with torch.no_grad():
        logits = model(samples[0].cuda()).float().cpu().numpy()
    return model, _harvest(model), logits


DATA_FREE = [n for n, (_, kw) in TINY_CONFIGS.items()
             if not kw.get("data") or (not kw.get("awq") and not kw.get("scale_estimation"))]
DATA_AWARE = [n for n in TINY_CONFIGS if n not in DATA_FREE]


@pytest.mark.parametrize("name", DATA_FREE)
def test_tiny_lm_bit_exact(name, tiny_golden):
    """Data-free lowering, and data-aware MIXED PRECISION (the statistics only pick which layers go to 4 bit; the
    lowering itself is data-free): everything the reference stored must be reproduced bit for bit."""
    z, meta = tiny_golden
    model, per, logits = _run_tiny(name)
    assert sorted(per) == meta[name]["modules"], "set of compressed weights differs from the reference"
    for mod, d in per.items():
        for k, v in d.items():
            ref = z[f"{name}|{mod}|{k}"]
            assert v.dtype == ref.dtype and v.shape == ref.shape, (mod, k, v.dtype, ref.dtype, v.shape, ref.shape)
            assert np.array_equal(v.view(np.uint8), ref.view(np.uint8)), (name, mod, k)
        assert f"{name}|{mod}|z" not in z.files or "z" in d
    ref_logits = z[f"{name}|logits"]
    np.testing.assert_allclose(logits, ref_logits, atol=2e-3 * np.abs(ref_logits).max())


@pytest.mark.parametrize("name", DATA_AWARE)
def test_tiny_lm_data_aware(name, tiny_golden):
    z, meta = tiny_golden
    model, per, logits = _run_tiny(name)
    assert sorted(per) == meta[name]["modules"]
    n_scales = ok_scales = n_codes = ok_codes = 0
    for mod, d in per.items():
        ref_w, ref_s = z[f"{name}|{mod}|w"], z[f"{name}|{mod}|s"]
        assert d["w"].dtype == ref_w.dtype and d["w"].shape == ref_w.shape, (mod, "precision assignment differs")
        assert d["s"].shape == ref_s.shape
        s, rs = d["s"].astype(np.float32), ref_s.astype(np.float32)
        ok_scales += int(np.sum(np.abs(s - rs) <= 1.1e-3 * np.abs(rs)))
        n_scales += s.size
        if ref_w.shape[-1] == 1 and ref_w.dtype == np.uint8 and ref_w.ndim == 2 and mod not in ("embed", "lm_head"):
            a = np.stack([d["w"] & 15, d["w"] >> 4], -1)
            b = np.stack([ref_w & 15, ref_w >> 4], -1)
        else:
            a, b = d["w"], ref_w
        ok_codes += int(np.sum(a == b))
        n_codes += a.size
    assert ok_scales >= 0.97 * n_scales, (ok_scales, n_scales)
    assert ok_codes >= 0.97 * n_codes, (ok_codes, n_codes)
    # the AWQ activation scale in front of the Act -> MatMul Linear
    for key in meta[name]["awq_keys"]:
        ref = z[f"{name}|awq|{key}"]
        got = model.blocks[2].mlp.fc2.awq_input_scale._scale_value.cpu().numpy()
        np.testing.assert_allclose(got.reshape(-1), ref.reshape(-1), rtol=1e-4)
    ref_logits = z[f"{name}|logits"]
    np.testing.assert_allclose(logits, ref_logits, atol=3e-2 * np.abs(ref_logits).max())


def test_opt125m_int8_asym_digests():
    """BASELINE configs[0]: random-init OPT-125m-shaped model (transformers.OPTConfig() defaults, seed 0), INT8_ASYM
    per-channel, data-free: all 74 compressed weights, scales and zero points hash to what the reference stored."""
    transformers = pytest.importorskip("transformers")
    import nncf_amd

    with open(os.path.join(GOLDEN, "model_opt125m_int8_asym.json")) as fh:
        gold = json.load(fh)
    torch.manual_seed(0)
    model = transformers.OPTForCausalLM(transformers.OPTConfig()).eval()
    ids = torch.tensor(gold["input_ids"])
    model = model.cuda()
    nncf_amd.compress_weights(model, dataset=nncf_amd.Dataset([{"input_ids": ids}]))
    per = _harvest(model)
    assert len(per) == gold["n_compressed"] == 74
    assert sorted(per) == sorted(gold["digests"])
    for mod, d in per.items():
        g = gold["digests"][mod]
        assert list(d["w"].shape) == g["shape"], mod
        for k in ("w", "s", "z"):
            assert hashlib.sha256(np.ascontiguousarray(d[k]).tobytes()).hexdigest() == g[k], (mod, k)
    with torch.no_grad():
        logits = model(input_ids=ids.cuda()).logits[0, :, :64].float().cpu().numpy()
    ref = np.asarray(gold["logits_0_64"], dtype=np.float32)
    np.testing.assert_allclose(logits, ref, atol=2e-3 * np.abs(ref).max())


@pytest.mark.parametrize("mode,kw", [("nf4", dict(group_size=32)),
                                     ("mxfp4", dict(backup_mode="int8_asym")),
                                     ("nf4", dict(group_size=32, all_layers=True, ratio=0.8))])
def test_tiny_lm_float_modes(mode, kw):
    """NF4 / MXFP4 at model level (the reference's Torch backend rejects them, quantize_model.py:504-525): every
    decompressed weight must equal the tensor-level oracle's dequantized weight (fp16 scale as stored)."""
    import nncf_amd
    from oracle import wc_oracle as O

    ref_model = make_tiny_lm()
    model = make_tiny_lm().cuda()
    samples = tiny_lm_dataset()
    nncf_amd.compress_weights(model, mode=mode, dataset=nncf_amd.Dataset(samples[:1]), **kw)
    report = model._nncf_amd_weight_compression
    n4 = 0
    for name, mod in model.named_modules():
        cfg = report.get(name)
        if cfg is None:
            continue
        w_ref = dict(ref_model.named_modules())[name].weight.detach().numpy()
        got = mod.weight.detach().float().cpu().numpy()  # parametrization -> decompressed
        codes, scale, zp = O.compress_weight(w_ref, (1,), cfg.mode.value, cfg.group_size)
        s16 = scale.astype(np.float16).astype(np.float32)  # scales are stored in fp16 (layers.py:1317-1434)
        if cfg.mode.value == "mxfp4":
            n4 += 1
            want = codes * scale  # E8M0 scale: exact
        elif cfg.mode.value == "nf4":
            n4 += 1
            want = (codes * s16).astype(np.float16)  # the product is taken in the scale's dtype
        elif zp is not None:
            want = ((codes.astype(np.int32) - zp) * s16).astype(np.float16)
        else:
            want = (codes.astype(np.float32) * s16).astype(np.float16)
        want = want.astype(np.float32).reshape(w_ref.shape)
        np.testing.assert_allclose(got, want, rtol=1.1e-3, atol=1e-9)
    assert n4 >= 15
    with torch.no_grad():
        out = model(samples[0].cuda())
    assert torch.isfinite(out).all()


def test_mxfp4_default_backup_is_out_of_scope():
    import nncf_amd
    from nncf_amd.errors import ParameterNotSupportedError

    model = make_tiny_lm().cuda()
    with pytest.raises(ParameterNotSupportedError):
        nncf_amd.compress_weights(model, mode="mxfp4", dataset=nncf_amd.Dataset(tiny_lm_dataset()[:1]))


@pytest.mark.parametrize("se", [False, True])
def test_tiny_lm_gptq(se):
    """GPTQ at model level ("parity unpinned" at this level: the reference rejects gptq on Torch; the tensor math is
    pinned in test_gpu_data_aware.py).  Checks here: the feeder's Hessian of the first group equals the tensor-level
    Hessian of the captured activations, the short-circuited feeder gives the same result as the plain one, and the
    layer-wise output error is below round-to-nearest's."""
    import nncf_amd
    from nncf_amd import gptq as G
    from nncf_amd import layerwise
    from nncf_amd.parameters import AdvancedCompressionParameters, AdvancedGPTQParameters

    import threading

    n_threads = threading.active_count()
    samples = tiny_lm_dataset()
    adv = AdvancedCompressionParameters(gptq_params=AdvancedGPTQParameters(subset_size=6))

    def run(gptq):
        model = make_tiny_lm().cuda()
        nncf_amd.compress_weights(model, mode="int4_sym", group_size=32, dataset=nncf_amd.Dataset(samples),
                                  gptq=gptq or None, scale_estimation=(se and gptq) or None, advanced_parameters=adv)
        return model

    ref = make_tiny_lm().cuda()
    m_gptq, m_rtn = run(True), run(False)
    with torch.no_grad():
        err_g = err_r = 0.0
        for s in samples:
            y = ref(s.cuda())
            err_g += float(((m_gptq(s.cuda()) - y) ** 2).mean())
            err_r += float(((m_rtn(s.cuda()) - y) ** 2).mean())
    assert err_g < err_r, (err_g, err_r)

    # feeder Hessian of the first group == tensor-level Hessian of the same activations
    model = make_tiny_lm().cuda()
    from nncf_amd.quantize_model import trace_model

    tr = trace_model(model, samples[0].cuda())
    feeder = layerwise.LayerwiseFeeder(model, nncf_amd.Dataset(samples), 6, tr, ["blocks.0.attn.q_proj",
                                                                               "blocks.0.attn.k_proj"])
    it = iter(feeder)
    group, H, stats = next(it)
    it.close()
    assert group == ["blocks.0.attn.q_proj", "blocks.0.attn.k_proj"]
    with torch.no_grad():
        xs = [model.blocks[0].norm1(model.embed(s.cuda())) for s in samples]
    H_ref = G.calculate_hessian(xs)[0]
    torch.testing.assert_close(H, H_ref, rtol=1e-5, atol=1e-7)
    assert len(stats.mean_values) == 6
    torch.testing.assert_close(stats.mean_values[0], xs[0].mean(dim=(0, 1)), rtol=1e-5, atol=1e-7)

    names = [n for n in tr.order if n != "embed"]
    mk = lambda **kw: list(layerwise.LayerwiseFeeder(model, nncf_amd.Dataset(samples), 6, tr, names, **kw))
    susp = mk()  # default: every sample's forward runs once, suspended in front of each group
    fast = mk(suspend=False, short_circuit=True)
    assert "forward" not in model.blocks[0].__dict__  # patches are removed when the feeder is exhausted
    plain = mk(suspend=False, short_circuit=False)
    assert [g for g, _, _ in fast] == [g for g, _, _ in plain] == [g for g, _, _ in susp] and len(fast) == 13
    for (g, h1, s1), (_, h2, s2), (_, h3, s3) in zip(fast, plain, susp):
        torch.testing.assert_close(h1, h2, rtol=1e-5, atol=1e-7, msg=str(g))
        torch.testing.assert_close(h3, h2, rtol=1e-5, atol=1e-7, msg=str(g))  # same launches, same order
        assert len(s3.mean_values) == len(s2.mean_values) == 6
        for a, b, c in zip(s1.mean_values, s2.mean_values, s3.mean_values):
            torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-8)
            torch.testing.assert_close(c, b, rtol=1e-6, atol=1e-8)
    import threading
    assert threading.active_count() == n_threads  # every suspended forward has ended

    # whole algorithm: both schemes feed GPTQ the same activations -> the same compressed model
    old = layerwise.DEFAULT_SUSPEND
    layerwise.DEFAULT_SUSPEND = False
    try:
        m_rerun = run(True)
    finally:
        layerwise.DEFAULT_SUSPEND = old
    sd_a, sd_b = m_gptq.state_dict(), m_rerun.state_dict()
    assert sd_a.keys() == sd_b.keys()
    for key in sd_a:  # (the Hessian's token-split atomics may reorder sums: allow isolated last-bit flips)
        a, b = sd_a[key], sd_b[key]
        if a.dtype in (torch.uint8, torch.int8):
            assert float((a != b).float().mean()) < 0.01, key
        else:
            torch.testing.assert_close(a.float(), b.float(), rtol=1e-2, atol=1e-3 * float(b.float().abs().max()), msg=key)


def test_suspended_feeder_cleans_up_on_errors_and_early_stop():
    """The suspended forwards are threads: they must all have ended when the consumer stops early, when a sample's
    forward raises, and a module that runs twice per forward must be seen once."""
    import threading

    import nncf_amd
    from nncf_amd import layerwise
    from nncf_amd.quantize_model import trace_model

    samples = tiny_lm_dataset()
    model = make_tiny_lm().cuda()
    tr = trace_model(model, samples[0].cuda())
    names = [n for n in tr.order if n != "embed"]
    n0 = threading.active_count()
    it = iter(layerwise.LayerwiseFeeder(model, nncf_amd.Dataset(samples), 6, tr, names))
    next(it), next(it)
    assert threading.active_count() == n0 + 6
    it.close()
    assert threading.active_count() == n0

    calls = {"n": 0}
    target = model.blocks[1].mlp if hasattr(model.blocks[1], "mlp") else model.blocks[1]

    def boom(mod, args):
        calls["n"] += 1
        if calls["n"] == 3:
            raise RuntimeError("boom")

    h = target.register_forward_pre_hook(boom)
    try:
        with pytest.raises(RuntimeError, match="boom"):
            list(layerwise.LayerwiseFeeder(model, nncf_amd.Dataset(samples), 6, tr, names))
    finally:
        h.remove()
    assert threading.active_count() == n0


@pytest.mark.parametrize("name", ["int4_asym_g32_r08_awq_se", "int8_asym_tied", "int4_sym_g32"])
def test_save_and_load_compressed(name, tmp_path):
    """On-disk export: safetensors round trip into a freshly built model gives the same tensors and the
    same logits bit for bit."""
    from nncf_amd.quantize_model import load_compressed, save_compressed

    model, per, logits = _run_tiny(name)
    path = str(tmp_path / "m.safetensors")
    save_compressed(model, path)
    mkw, _ = TINY_CONFIGS[name]
    fresh = make_tiny_lm(seed=7, **mkw).cuda()
    load_compressed(fresh, path)
    sd1, sd2 = model.state_dict(), fresh.state_dict()
    assert sorted(sd1) == sorted(sd2)
    for k in sd1:
        assert torch.equal(sd1[k], sd2[k]), k
    with torch.no_grad():
        again = fresh(tiny_lm_dataset()[0].cuda()).float().cpu().numpy()
    assert np.array_equal(again, logits)


@pytest.mark.parametrize("layout", ["native", "nncf"])
def test_save_and_load_compressed_layouts(layout, tmp_path):
    from nncf_amd.quantize_model import load_compressed, save_compressed

    name = "int4_asym_g32_r08_awq_se"
    model, per, logits = _run_tiny(name)
    path = str(tmp_path / "m.safetensors")
    save_compressed(model, path, layout=layout)
    mkw, _ = TINY_CONFIGS[name]
    fresh = make_tiny_lm(seed=7, **mkw).cuda()
    load_compressed(fresh, path)
    with torch.no_grad():
        again = fresh(tiny_lm_dataset()[0].cuda()).float().cpu().numpy()
    assert np.array_equal(again, logits)
    with pytest.raises(ValueError):
        save_compressed(model, path, layout="gguf")


@pytest.mark.parametrize("name", DATA_FREE + ["int4_asym_g32_awq", "int4_asym_g32_r08_awq_se"])
def test_export_in_the_reference_state_dict_layout(name, tiny_golden, tmp_path):
    """SURVEY 8 f4 [probed layout]: `nncf_state_dict` / `save_compressed(layout="nncf")` emit the keys the reference's
    compressed model holds -- packed `<module>.weight`, `__nncf_hooks.post_hooks.<mod:with:colons>:weight__0.0._scale /
    ._zero_point`, `__nncf_hooks.pre_hooks.<node>__0.0._scale_value` -- parsed here exactly as tools/gen_golden_model.py
    parsed the reference's state_dict when it wrote the fixture.  Data-free configurations: same key set, same bytes;
    AWQ configurations (statistics from a GPU forward): same key set incl. the pre-hook's node name, scales close."""
    from safetensors import safe_open

    from nncf_amd.quantize_model import nncf_state_dict, save_compressed

    z, meta = tiny_golden
    model, per, _ = _run_tiny(name)
    sd = nncf_state_dict(model)
    path = str(tmp_path / "ref_layout.safetensors")
    save_compressed(model, path, layout="nncf")
    with safe_open(path, framework="pt", device="cpu") as fh:
        on_disk = set(fh.keys())
        aliases = json.loads(fh.metadata()["nncf_amd_weight_compression"])["aliases"]
    assert on_disk | set(aliases) == set(sd)
    assert not any(".parametrizations." in k or "awq_input_scale" in k for k in sd)
    found, awq = {}, {}
    for k, v in sd.items():  # the harvest of tools/gen_golden_model.py, on OUR dict
        if k.startswith("__nncf_hooks.post_hooks.") and (k.endswith("._scale") or k.endswith("._zero_point")):
            mid = k.split(".")[2]
            node = mid.rsplit("__", 1)[0]
            assert node.endswith(":weight"), k
            mod = node[: -len(":weight")].replace(":", ".")
            found.setdefault(mod, {})["s" if k.endswith("._scale") else "z"] = v.cpu().numpy()
        elif k.startswith("__nncf_hooks.pre_hooks.") and k.endswith("._scale_value"):
            awq[k.split(".")[2]] = v.cpu().numpy()
    for mod in found:
        found[mod]["w"] = sd[mod + ".weight"].cpu().numpy()
    assert sorted(found) == meta[name]["modules"]
    assert sorted(awq) == meta[name]["awq_keys"]
    exact = name in DATA_FREE
    for mod, d in found.items():
        for kind, v in d.items():
            ref = z[f"{name}|{mod}|{kind}"]
            assert v.dtype == ref.dtype and v.shape == ref.shape, (mod, kind, v.dtype, ref.dtype, v.shape, ref.shape)
            if exact:
                assert np.array_equal(v.view(np.uint8), ref.view(np.uint8)), (name, mod, kind)
        assert f"{name}|{mod}|z" not in z.files or "z" in d
    for key, v in awq.items():
        ref = z[f"{name}|awq|{key}"]
        assert v.shape == ref.shape and v.dtype == ref.dtype
        np.testing.assert_allclose(v.astype(np.float32), ref.astype(np.float32), rtol=1e-4)
    # everything that is not a compressed weight keeps its own key and value
    own = model.state_dict()
    plain = {k for k in sd if k in own}
    assert plain and all(torch.equal(sd[k], own[k]) for k in plain)
    assert all(k.startswith("__nncf_hooks.") or k.endswith(".weight") for k in sd if k not in own)


def test_statistics_cache_is_reused(tmp_path):
    """with `statistics_path` the calibration statistics are written once and a second compression
    (here: another algorithm mix over the same statistics) does not run the calibration set again."""
    import nncf_amd
    from nncf_amd.parameters import AdvancedCompressionParameters

    path = str(tmp_path / "stats.safetensors")
    adv = AdvancedCompressionParameters(statistics_path=path)
    samples = tiny_lm_dataset()
    m1 = make_tiny_lm().cuda()
    nncf_amd.compress_weights(m1, mode="int4_asym", group_size=32, scale_estimation=True,
                              dataset=nncf_amd.Dataset(samples), advanced_parameters=adv)
    assert os.path.isfile(path)

    class OnlyFirst:
        """Yields the tracing sample; touching a second item means the statistics were collected again."""
        def __iter__(self):
            yield samples[0]
            raise AssertionError("calibration data was read although a statistics cache exists")

    m2 = make_tiny_lm().cuda()
    nncf_amd.compress_weights(m2, mode="int4_asym", group_size=32, scale_estimation=True,
                              dataset=nncf_amd.Dataset(OnlyFirst()), advanced_parameters=adv)
    for (k1, v1), (k2, v2) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2), k1


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_tiny_lm_16bit_models(dtype):
    """Models usually live in bf16 / fp16: data-free codes must equal the oracle's lowering of the same 16-bit
    weights (upcast first, weight_lowering.py:493-494), and the data-aware pipeline must run end to end."""
    import nncf_amd
    from oracle import wc_oracle as O

    samples = tiny_lm_dataset()
    ref = make_tiny_lm().to(dtype)
    model = make_tiny_lm().to(dtype).cuda()
    nncf_amd.compress_weights(model, mode="int4_sym", group_size=32, dataset=nncf_amd.Dataset(samples[:1]))
    per = _harvest(model)
    assert len(per) == 22
    for name, d in per.items():
        w = dict(ref.named_modules())[name].weight.detach().float().numpy()
        four = name not in ("embed", "lm_head")
        codes, scale, zp = O.compress_weight(w, (1,), "int4_sym" if four else "int8_asym", 32 if four else -1)
        want = O.pack_int4(codes) if four else codes
        assert np.array_equal(d["w"].reshape(-1), np.asarray(want).reshape(-1)), name
        assert np.array_equal(d["s"].reshape(-1), scale.astype(np.float16).reshape(-1)), name
    assert model.blocks[0].attn.q_proj.weight.dtype == dtype  # decompressed in the layer's dtype

    model = make_tiny_lm().to(dtype).cuda()
    base = make_tiny_lm().to(dtype).cuda()
    nncf_amd.compress_weights(model, mode="int4_asym", group_size=32, ratio=0.8, awq=True, scale_estimation=True,
                              dataset=nncf_amd.Dataset(samples))
    with torch.no_grad():
        y, y0 = model(samples[0].cuda()).float(), base(samples[0].cuda()).float()
    assert torch.isfinite(y).all()
    assert float((y - y0).abs().max()) < 0.35 * float(y0.abs().max())
    model = make_tiny_lm().to(dtype).cuda()
    nncf_amd.compress_weights(model, mode="nf4", group_size=32, gptq=True, dataset=nncf_amd.Dataset(samples))
    with torch.no_grad():
        y = model(samples[0].cuda()).float()
    assert torch.isfinite(y).all() and float((y - y0).abs().max()) < 0.35 * float(y0.abs().max())


def _tiny_llama(seed=0):
    transformers = pytest.importorskip("transformers")
    cfg = transformers.LlamaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=3, num_attention_heads=4,
                                   num_key_value_heads=2, vocab_size=1000, max_position_embeddings=128,
                                   tie_word_embeddings=False)
    torch.manual_seed(seed)
    return transformers.LlamaForCausalLM(cfg).eval()


def _llama_dataset(n=5):
    g = torch.Generator().manual_seed(5)
    return [{"input_ids": torch.randint(0, 1000, (1, 24 + i), generator=g),
             "attention_mask": torch.ones(1, 24 + i, dtype=torch.long)} for i in range(n)]


@pytest.mark.parametrize("algo", ["data_free", "awq_se", "gptq", "mixed"])
def test_hf_llama_architecture(algo):
    """The driver on a real transformers architecture (LlamaForCausalLM, random init, 3 layers): keyword inputs,
    rotary embeddings, grouped-query attention, decoder layers called with keyword arguments -- tracing, the AWQ
    pattern of the gated MLP, the short-circuited layer-wise GPTQ feeder and the parametrized forward all have to
    work on it."""
    import nncf_amd
    from nncf_amd.parameters import AdvancedCompressionParameters, AdvancedGPTQParameters

    data = _llama_dataset()
    base = _tiny_llama().cuda()
    model = _tiny_llama().cuda()
    ids = {k: v.cuda() for k, v in data[0].items()}
    with torch.no_grad():
        y0 = base(**ids).logits.float()
    if algo == "data_free":
        nncf_amd.compress_weights(model, mode="int4_sym", group_size=64, dataset=nncf_amd.Dataset(data[:1]))
    elif algo == "awq_se":
        up_before = model.model.layers[0].mlp.up_proj.weight.detach().clone()
        nncf_amd.compress_weights(model, mode="int4_asym", group_size=64, awq=True, scale_estimation=True,
                                  dataset=nncf_amd.Dataset(data))
    elif algo == "gptq":
        adv = AdvancedCompressionParameters(gptq_params=AdvancedGPTQParameters(subset_size=5))
        nncf_amd.compress_weights(model, mode="int4_sym", group_size=64, gptq=True, dataset=nncf_amd.Dataset(data),
                                  advanced_parameters=adv)
    else:
        nncf_amd.compress_weights(model, mode="nf4", group_size=64, ratio=0.7, all_layers=True,
                                  sensitivity_metric="max_activation_variance", dataset=nncf_amd.Dataset(data))
    report = model._nncf_amd_weight_compression
    assert len(report) == 3 * 7 + 2
    if algo != "mixed":
        assert report["lm_head"].num_bits == 8 and report["model.embed_tokens"].num_bits == 8
        assert all(cfg.num_bits == 4 for name, cfg in report.items() if ".layers." in name)
    else:
        bits = [cfg.num_bits for cfg in report.values()]
        assert 4 in bits and 8 in bits
    with torch.no_grad():
        y = model(**ids).logits.float()
    assert torch.isfinite(y).all()
    assert float((y - y0).abs().max()) < 0.5 * float(y0.abs().max())
    assert not any("forward" in blk.__dict__ for blk in model.model.layers)  # feeder patches are gone
    if algo == "awq_se":
        # MatMul -> Multiply -> MatMul: the activation scale of down_proj was merged into the rows of up_proj
        from nncf_amd.quantize_model import trace_model

        tr = trace_model(_tiny_llama(), data[0], with_provenance=True)
        assert tr.producer["model.layers.0.mlp.down_proj"] == ("mul_linear", "model.layers.0.mlp.up_proj")
        assert not hasattr(model.model.layers[0].mlp.down_proj, "awq_input_scale")


class _AwqLinearModel(torch.nn.Module):
    """The six-Linear model of the reference's AWQ scale test (tests/torch/function_hook/quantization/
    test_weights_compression.py:208-258), restated: every weight = 0.01 * arange(64).reshape(8, 8) + 0.05; mergeable:
    linear3 reads linear1(x) * linear2(x) (MatMul -> Multiply -> MatMul), non-mergeable: linear2 reads relu(linear1(x))."""

    def __init__(self, non_mergable_pattern):
        super().__init__()
        self.non_mergable_pattern = non_mergable_pattern
        for i in range(1, 5 if non_mergable_pattern else 7):
            lin = torch.nn.Linear(8, 8, bias=False)
            lin.weight = torch.nn.Parameter(0.01 * torch.arange(0, 64, dtype=torch.float32).reshape(8, 8) + 0.05)
            setattr(self, f"linear{i}", lin)

    def forward(self, x):
        if self.non_mergable_pattern:
            return self.linear4(torch.relu(self.linear3(self.linear2(torch.relu(self.linear1(x))))))
        n3 = self.linear3(self.linear1(x) * self.linear2(x))
        return self.linear6(self.linear4(n3) * self.linear5(n3))


@pytest.mark.parametrize("non_mergable_pattern", [True, False])
def test_awq_scale_reference_literals(non_mergable_pattern):
    """The reference's own known answers for the AWQ activation scales (template_test_weights_compression.py:766-803
    with the Torch literals of tests/torch/function_hook/quantization/test_weights_compression.py:856-884): INT4_SYM,
    ratio 1, per-channel, two copies of input 0.01 * arange + 0.02, rtol 1e-5 (fns.allclose).  The reference keys are
    its graph node names ("linear3/linear/0"), ours the module names."""
    ref = {"linear3": np.array([[1.226455], [1.205499], [1.141340], [1.097436], [1.064355], [1.037971], [1.016118],
                                [0.997526]], np.float32),
           "linear2": np.array([[[1.990990, 1.863296, 1.575980, 1.397459, 1.272275, 1.177998, 1.103558, 1.042768]]],
                               np.float32)}
    import nncf_amd
    model = _AwqLinearModel(non_mergable_pattern).cuda()
    x = torch.from_numpy(0.01 * np.arange(0, 32, dtype=np.float32).reshape(1, 4, 8) + 0.02)
    nncf_amd.compress_weights(model, mode="int4_sym", ratio=1.0, all_layers=False, group_size=-1,
                              dataset=nncf_amd.Dataset([x] * 2), awq=True)
    scales = model._nncf_amd_awq_scales
    assert set(scales) == {"linear2" if non_mergable_pattern else "linear3"}
    for name, sc in scales.items():
        assert tuple(sc.shape) == ref[name].shape
        np.testing.assert_allclose(sc.cpu().numpy(), ref[name], rtol=1e-5, atol=1e-8)


class _SequentialMatmulModel(torch.nn.Module):
    """Five 4x4 Linears in a row, weight = arange(16).reshape(4, 4) with the last element replaced by 10000, 1000, 1,
    10, 10000, stored transposed (tests/torch/function_hook/quantization/test_weights_compression.py:67-84, restated)."""

    def __init__(self):
        super().__init__()
        self.layers = torch.nn.ModuleList()
        for main_value in (10000, 1000, 1, 10, 10000):
            w = torch.arange(0, 16, dtype=torch.float32).reshape(4, 4)
            w[-1, -1] = main_value
            layer = torch.nn.Linear(4, 4, bias=False)
            layer.weight = torch.nn.Parameter(w.t().contiguous())
            self.layers.append(layer)

    def forward(self, x):
        for layer in self.layers:
            x = layer(x)
        return x


@pytest.mark.parametrize("metric,all_layers,ratio,ref_ids", [
    ("weight_quantization_error", True, 1, [0, 1, 2, 3, 4]), ("weight_quantization_error", True, 0.8, [0, 3, 4]),
    ("weight_quantization_error", True, 0.4, [0]), ("weight_quantization_error", True, 0.2, []),
    ("weight_quantization_error", False, 1, [0, 1, 2, 3]), ("weight_quantization_error", False, 0.8, [0, 1, 3]),
    ("weight_quantization_error", False, 0.4, [0]), ("weight_quantization_error", False, 0.2, []),
    ("hessian_input_activation", True, 0.8, [0, 1, 2]), ("hessian_input_activation", False, 0.8, [0, 1, 2]),
    ("mean_activation_variance", True, 0.8, [0, 1, 2]), ("mean_activation_variance", False, 0.8, [0, 1, 2]),
    ("max_activation_variance", True, 0.8, [0, 1, 2]), ("max_activation_variance", False, 0.8, [0, 1, 2]),
    ("mean_activation_magnitude", True, 0.8, [0, 1, 2]), ("mean_activation_magnitude", False, 0.8, [0, 1, 2])])
def test_mixed_precision_reference_ids(metric, all_layers, ratio, ref_ids):
    """The reference's known answers for which Linears end up in INT4 (template_test_weights_compression.py:207-246,
    transpose_a = False as on its Torch backend): INT4_SYM, group size 1, two samples (ones, arange(16))."""
    import nncf_amd
    from nncf_amd.decompressors import INT4SymmetricWeightsDecompressor
    model = _SequentialMatmulModel().cuda()
    first = torch.ones((1, 4, 4), dtype=torch.float32)
    second = torch.arange(16, dtype=torch.float32).reshape(1, 4, 4)
    nncf_amd.compress_weights(model, mode="int4_sym", ratio=ratio, group_size=1, all_layers=all_layers,
                              sensitivity_metric=metric, dataset=nncf_amd.Dataset([first, second]))
    got = [i for i, layer in enumerate(model.layers)
           if any(isinstance(m, INT4SymmetricWeightsDecompressor) for m in layer.modules())]
    assert got == ref_ids


@pytest.mark.parametrize("metric,ref_act_score,with_baseline", [
    ("hessian_input_activation", (16 + 1 + 4) * 2 / 9, False), ("mean_activation_magnitude", 2.333333, True),
    ("mean_activation_variance", 1.555555, True), ("max_activation_variance", 3.555555, True)])
def test_data_based_criterion_reference_scores(metric, ref_act_score, with_baseline):
    """template_test_weights_compression.py:147-178: activation [[[-4, 1, 2], [0, 0, 0], [0, 0, 0]]] in front of a
    255 * eye(3) weight, INT4_ASYM group size 1, ratio 0.5, all layers: the activation sensitivity equals the literal
    (rel 1e-5) and the weight's score is 0 for the Hessian metric and literal / eps(fp32) for the others."""
    import nncf_amd
    model = torch.nn.Linear(3, 3, bias=False)
    model.weight = torch.nn.Parameter(255 * torch.eye(3, dtype=torch.float32))
    model = torch.nn.Sequential(model).cuda()
    x = torch.tensor([[[-4, 1, 2], [0, 0, 0], [0, 0, 0]]], dtype=torch.float32)
    nncf_amd.compress_weights(model, mode="int4_asym", ratio=0.5, group_size=1, all_layers=True,
                              sensitivity_metric=metric, dataset=nncf_amd.Dataset([x]))
    dbg = model._nncf_amd_mixed_precision
    assert math.isclose(dbg["activation_scores"]["0"], ref_act_score, rel_tol=1e-5, abs_tol=1e-8)
    ref_score = ref_act_score / 1.1920928955078125e-07 if with_baseline else 0.0
    assert math.isclose(dbg["scores"][0], ref_score, rel_tol=1e-5, abs_tol=1e-8)


@pytest.mark.parametrize("sampled", [False, True], ids=["full", "sampled"])
def test_scale_estimation_reference_literals(sampled):
    """The reference's known answers for scale estimation at model level (template_test_weights_compression.py:281-326
    with the Torch literals of tests/torch/function_hook/quantization/test_weights_compression.py:657-700): one
    Linear(8 -> 16) with weight arange(128).reshape(16, 8), INT4_ASYM group size 8, input arange(32) (+ i for sample
    i); "sampled": two calibration samples of which scale estimation uses one (the sampling flow of process_stats)."""
    import nncf_amd
    ref = ([0.473328, 0.929023, 1.446527, 1.920595, 2.517054, 3.030102, 3.584279, 4.043509, 4.620008, 5.165322, 5.710637,
            6.122581, 6.655914, 7.237174, 7.722580, 8.255914],
           [0.473445, 0.928777, 1.446328, 1.920052, 2.516778, 3.029870, 3.584271, 4.042929, 4.619769, 5.165224, 5.710679,
            6.121212, 6.654546, 7.236652, 7.721212, 8.254545])[sampled]
    lin = torch.nn.Linear(8, 16, bias=False)
    lin.weight = torch.nn.Parameter(torch.arange(0, 8 * 16, dtype=torch.float32).reshape(16, 8))
    model = torch.nn.Sequential(lin).cuda()
    x = torch.arange(0, 4 * 8, dtype=torch.float32).reshape(1, 4, 8)
    subset = 2 if sampled else 1
    se_subset = subset // 2 if sampled else subset
    nncf_amd.compress_weights(
        model, mode="int4_asym", ratio=1.0, group_size=8, scale_estimation=True, all_layers=True,
        dataset=nncf_amd.Dataset([x + i for i in range(subset)]), subset_size=subset,
        advanced_parameters=nncf_amd.AdvancedCompressionParameters(
            scale_estimation_params=nncf_amd.AdvancedScaleEstimationParameters(subset_size=se_subset)))
    (scale,) = model._nncf_amd_se_scales.values()
    assert tuple(scale.shape) == (16, 1, 1)
    np.testing.assert_allclose(scale.cpu().numpy().reshape(-1), np.array(ref, np.float32), rtol=1e-5, atol=1e-8)
