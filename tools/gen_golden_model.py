#!/usr/bin/env python3
"""
Model-level golden fixtures (tests/golden/model_*.npz): runs the REFERENCE nncf.compress_weights (imported from
/root/reference/src, build container only) on the synthetic models of tests/tiny_models.py and on a random-init
OPT-125m-shaped model (BASELINE config[0]), and stores what the compressed state_dict holds per weight.

    PYTHONPATH=/root/reference/src PYTHONDONTWRITEBYTECODE=1 python3 tools/gen_golden_model.py

Fixtures are data only: packed weights, fp16 scales, zero points, AWQ activation scales, a logits slice; for the
125 M-parameter OPT model sha256 digests of those tensors instead of the tensors.
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.environ.get("NNCF_REFERENCE_SRC", "/root/reference/src"))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

import nncf  # noqa: E402
from nncf import CompressWeightsMode as M  # noqa: E402
from nncf import SensitivityMetric as SM  # noqa: E402

from tests.tiny_models import TINY_CONFIGS, make_tiny_lm, tiny_lm_dataset  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")



def _ref_kwargs(kw):
    """tests/tiny_models.py TINY_CONFIGS values (plain strings) -> the reference's enums."""
    kw = dict(kw)
    kw["mode"] = M(kw["mode"])
    if "sensitivity_metric" in kw:
        kw["sensitivity_metric"] = SM(kw["sensitivity_metric"])
    if "backup_mode" in kw:
        kw["backup_mode"] = nncf.BackupMode(kw["backup_mode"])
    return kw


def _module_of_hook_key(key):
    # "__nncf_hooks.post_hooks.blocks:0:attn:q_proj:weight__0.0._scale" -> "blocks.0.attn.q_proj"
    mid = key.split(".")[2]
    name = mid.rsplit("__", 1)[0]
    assert name.endswith(":weight"), key
    return name[: -len(":weight")].replace(":", ".")


def harvest(model, first_sample):
    """{module_name: {"w": packed, "s": scale, "z": zp}}, {hook key: awq scale}, logits."""
    sd = model.state_dict()
    per = {}
    awq = {}
    for k, v in sd.items():
        if k.startswith("__nncf_hooks.post_hooks.") and (k.endswith("._scale") or k.endswith("._zero_point")):
            mod = _module_of_hook_key(k)
            per.setdefault(mod, {})["s" if k.endswith("._scale") else "z"] = v.numpy()
        elif k.startswith("__nncf_hooks.pre_hooks.") and k.endswith("._scale_value"):
            awq[k.split(".")[2]] = v.numpy()
    for mod in per:
        per[mod]["w"] = sd[mod + ".weight"].numpy()
    with torch.no_grad():
        logits = model(first_sample)
    if not isinstance(logits, torch.Tensor):
        logits = logits.logits
    return per, awq, logits.float().numpy()


def run_tiny(name, model_kw, kw):
    kw = _ref_kwargs(kw)
    data = kw.pop("data", False)
    fallback = kw.pop("fallback", None)
    ignored = kw.pop("ignored", None)
    model = make_tiny_lm(**model_kw)
    samples = tiny_lm_dataset()
    ds = nncf.Dataset(samples if data else samples[:1])
    if fallback is not None:
        kw["advanced_parameters"] = nncf.AdvancedCompressionParameters(
            group_size_fallback_mode=nncf.GroupSizeFallbackMode(fallback), min_adjusted_group_size=16)
    if ignored is not None:
        # the reference names graph nodes, not modules, and its tracer does not see nn.ModuleList indexing:
        # module "blocks.1.attn.q_proj" is the node "attn/q_proj/linear/1" (second call of that scope)
        names = []
        for m in ignored:
            _, idx, rest = m.split(".", 2)
            names.append(rest.replace(".", "/") + "/linear/" + idx)
        kw["ignored_scope"] = nncf.IgnoredScope(names=names)
    cm = nncf.compress_weights(model, dataset=ds, **kw)
    return harvest(cm, samples[0])


def gen_tiny():
    out = {}
    meta = {}
    for name, (mkw, kw) in TINY_CONFIGS.items():
        per, awq, logits = run_tiny(name, mkw, kw)
        meta[name] = {"modules": sorted(per), "awq_keys": sorted(awq)}
        for mod, d in per.items():
            for k, v in d.items():
                out[f"{name}|{mod}|{k}"] = v
        for k, v in awq.items():
            out[f"{name}|awq|{k}"] = v
        out[f"{name}|logits"] = logits.astype(np.float32)
        print(name, len(per), "compressed weights,", len(awq), "awq hooks")
    out["__meta__"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, "model_tiny.npz"), **out)


def gen_opt():
    """BASELINE config[0]: OPTForCausalLM(OPTConfig()) random init (seed 0), INT8_ASYM per-channel, data-free."""
    from transformers import OPTConfig, OPTForCausalLM

    torch.manual_seed(0)
    model = OPTForCausalLM(OPTConfig()).eval()
    ids = torch.randint(0, 50272, (1, 16), generator=torch.Generator().manual_seed(0))
    cm = nncf.compress_weights(model, dataset=nncf.Dataset([{"input_ids": ids}]))
    sd = cm.state_dict()
    digests = {}
    for k, v in sd.items():
        if k.startswith("__nncf_hooks.post_hooks.") and (k.endswith("._scale") or k.endswith("._zero_point")):
            mod = _module_of_hook_key(k)
            digests.setdefault(mod, {})["s" if k.endswith("._scale") else "z"] = hashlib.sha256(
                v.contiguous().numpy().tobytes()).hexdigest()
    for mod in digests:
        w = sd[mod + ".weight"]
        digests[mod]["w"] = hashlib.sha256(w.contiguous().numpy().tobytes()).hexdigest()
        digests[mod]["shape"] = list(w.shape)
        digests[mod]["dtype"] = str(w.dtype)
    with torch.no_grad():
        logits = cm(input_ids=ids).logits[0, :, :64].float().numpy()
    with open(os.path.join(OUT, "model_opt125m_int8_asym.json"), "w") as fh:
        json.dump({"n_compressed": len(digests), "digests": digests, "logits_0_64": logits.tolist(),
                   "input_ids": ids.tolist()}, fh)
    print("opt-125m:", len(digests), "compressed weights")


if __name__ == "__main__":
    which = sys.argv[1:] or ["tiny", "opt"]
    if "tiny" in which:
        gen_tiny()
    if "opt" in which:
        gen_opt()
