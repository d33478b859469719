"""Host logic of the layer-wise GPTQ feeder (nncf_amd/layerwise.py) without a GPU: the Hessian accumulator and the
token-mean reducer are HIP-only in the product, so they are replaced here by torch stand-ins (test-only) and the
thread hand-over of the suspended forwards is what gets tested: group order, the activations each group sees
when the consumer rewrites the weights between groups (the GPTQ contract), equality with the re-run scheme, and that
no thread outlives the iteration."""
import threading

import pytest
import torch

from nncf_amd import layerwise as LW
from nncf_amd import quantize_model as qm
from nncf_amd.parameters import Dataset
from tests.tiny_models import make_tiny_lm, tiny_lm_dataset


class _TorchAccumulator:
    def __init__(self, k, device, chunk_tokens=0):
        self.k, self.n, self.H = k, 0, torch.zeros(k, k, dtype=torch.float64)

    def add(self, x2d, batch_size):
        x = x2d.double()
        self.H += x.T @ x
        self.n += batch_size

    def result(self):
        return self.H * (2.0 / max(self.n, 1))


@pytest.fixture(autouse=True)
def stand_ins(monkeypatch):
    monkeypatch.setattr(LW, "HessianAccumulator", _TorchAccumulator)
    monkeypatch.setattr(LW, "activation_stats", lambda x, want_stats=False: (x[0].float().mean(dim=0), None))


def _consume(model, samples, names, tr, **kw):
    """Plays GPTQ's part: after every group, the group's weights are overwritten (here: coarsely rounded)."""
    mods = dict(model.named_modules())
    out = []
    for group, H, stats in LW.LayerwiseFeeder(model, Dataset(samples), len(samples), tr, names, **kw):
        out.append((group, H.clone(), [m.clone() for m in stats.mean_values], list(stats.shape_values)))
        for n in group:
            w = mods[n].weight.data
            mods[n].weight.data = torch.round(w * 16) / 16
    return out


@pytest.mark.timeout(120)
def test_suspended_feeder_equals_rerun_feeder_under_weight_updates():
    samples = tiny_lm_dataset()
    n0 = threading.active_count()
    runs = {}
    for key, kw in (("suspend", {}), ("rerun_short", dict(suspend=False, short_circuit=True)),
                    ("rerun_plain", dict(suspend=False, short_circuit=False))):
        model = make_tiny_lm()
        tr = qm.trace_model(model, samples[0])
        names = [n for n in tr.order if n != "embed"]
        runs[key] = _consume(model, samples, names, tr, **kw)
        assert threading.active_count() == n0
        assert "forward" not in model.blocks[0].__dict__
    a, b, c = runs["suspend"], runs["rerun_short"], runs["rerun_plain"]
    assert [g for g, *_ in a] == [g for g, *_ in b] == [g for g, *_ in c] and len(a) == 13
    assert a[0][0] == ["blocks.0.attn.q_proj", "blocks.0.attn.k_proj", "blocks.0.attn.v_proj"]
    for (g, h1, m1, s1), (_, h2, m2, s2), (_, h3, m3, s3) in zip(a, b, c):
        assert s1 == s2 == s3 and len(s1) == len(samples)
        torch.testing.assert_close(h1, h3, rtol=1e-9, atol=1e-12, msg=str(g))
        torch.testing.assert_close(h2, h3, rtol=1e-9, atol=1e-12, msg=str(g))
        for x, y in zip(m1, m3):
            torch.testing.assert_close(x, y, rtol=1e-6, atol=1e-8)
    # the rewritten weights matter: without the updates the later groups see other activations
    model = make_tiny_lm()
    tr = qm.trace_model(model, samples[0])
    names = [n for n in tr.order if n != "embed"]
    untouched = [(g, H.clone()) for g, H, _ in LW.LayerwiseFeeder(model, Dataset(samples), len(samples), tr, names)]
    torch.testing.assert_close(untouched[0][1], a[0][1])
    assert not torch.allclose(untouched[-1][1], a[-1][1], rtol=1e-6, atol=1e-9)


@pytest.mark.timeout(120)
def test_suspended_feeder_early_stop_error_and_repeated_module():
    samples = tiny_lm_dataset()
    model = make_tiny_lm()
    tr = qm.trace_model(model, samples[0])
    names = [n for n in tr.order if n != "embed"]
    n0 = threading.active_count()
    it = iter(LW.LayerwiseFeeder(model, Dataset(samples), len(samples), tr, names))
    next(it), next(it)
    assert threading.active_count() == n0 + len(samples)  # every forward waits in front of the third group
    it.close()
    assert threading.active_count() == n0

    calls = {"n": 0}

    def boom(mod, args):
        calls["n"] += 1
        if calls["n"] == 3:
            raise RuntimeError("boom")

    h = model.blocks[1].mlp.register_forward_pre_hook(boom)
    try:
        with pytest.raises(RuntimeError, match="boom"):
            list(LW.LayerwiseFeeder(model, Dataset(samples), len(samples), tr, names))
    finally:
        h.remove()
    assert threading.active_count() == n0

    # a Linear that runs twice per forward is seen at its first call only (the re-run scheme stops there too)
    lin = model.blocks[0].attn.o_proj
    orig = lin.forward
    extra = []
    hk = model.blocks[0].norm2.register_forward_hook(lambda m, a, out: extra.append(lin(out.detach())) and None)
    try:
        got = list(LW.LayerwiseFeeder(model, Dataset(samples), len(samples), tr, names))
    finally:
        hk.remove()
    assert len(got) == 13 and len(extra) == len(samples)
    assert lin.forward == orig and threading.active_count() == n0
