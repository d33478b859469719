"""Layer-wise input feeder for GPTQ on torch models: what the reference's OpenVINO-only
`LayerwiseEngine` does for GPTQ.apply (src/nncf/quantization/algorithms/layerwise/engine.py:145-170,
openvino_iterator.py:166-229; consumed at gptq.py:116-144) -- every target Linear sees the activations produced by
the model whose EARLIER Linears are already quantized.

MI355X-first shape of it:
* the raw activations X are never stored: each calibration sample streams through `nncf_amd_hessian_accumulate`
  (MFMA SYRK, running average of gptq.py:237-240) and the fused token-mean reducer while the hooked forward runs,
  then the forward is cut short (the rest of the model is not needed);
* Linears that read one activation (q/k/v, gate/up) form one group: ONE Hessian serves all of them;
* default (`suspend=True`): every calibration sample's forward runs exactly ONCE.  Each sample's forward lives in
  its own Python thread and is suspended in the pre-hook of the first Linear of every group; the feeder resumes
  the samples one after the other (strict hand-over: never two threads at once, so the order of every launch is
  the sequential one), and once all of them wait in front of a group its Hessian is complete.  The caller
  quantizes the group (in-place `weight.data` swap), the suspended forwards go on through the quantized Linears to
  the next group.  No prefix is ever recomputed (the re-run scheme below costs about two extra block forwards per
  block and a model call per sample and group), no assumption about the model's structure is needed, and what a
  suspended forward holds is its live activations (tens of MB per sample against 288 GB of HBM).
* `suspend=False`: the re-run scheme.  Every group is a pass over the samples that is cut short at the group;
  finished decoder blocks (children of an nn.ModuleList) are short-circuited: their forward returns the cached
  output of the latest finished block, so a pass costs the unfinished part of one block, not the model prefix.
  This assumes the blocks of a ModuleList run as a chain (decoder-only transformers); `short_circuit=False` runs the
  real prefix instead.
"""
from __future__ import annotations

import threading
from typing import Dict, Iterator, List, Optional, Tuple

import torch
import torch.nn as nn

from nncf_amd.gptq import HESSIAN_CHUNK_TOKENS, HessianAccumulator
from nncf_amd.statistics import WCTensorStatistic, activation_stats


DEFAULT_SUSPEND = True  # see the module docstring; tests flip it to compare the two schemes
# SYRK launches of 32k tokens while the calibration forwards are still being issued: they fill the gaps of the
# host-bound stretches of a forward, which one long launch per group at the end cannot (Llama-3-8B-shaped model, GPTQ:
# 10.3 s against 10.5 s with gptq._list_chunk_tokens' one launch per K = 4096 group)
FEEDER_CHUNK_TOKENS = HESSIAN_CHUNK_TOKENS


class _EarlyExit(Exception):
    pass


def _detach(out):
    if isinstance(out, torch.Tensor):
        return out.detach()
    if isinstance(out, (tuple, list)):
        return type(out)(_detach(o) for o in out)
    if isinstance(out, dict):
        return {k: _detach(v) for k, v in out.items()}
    return out


class _SuspendedForward(threading.Thread):
    """One calibration sample's forward, run in a thread of its own so that it can wait inside a hook."""

    def __init__(self, feeder: "LayerwiseFeeder", item, device, stream):
        super().__init__(daemon=True)
        self.feeder, self.item, self.device, self.stream = feeder, item, device, stream
        self.go = threading.Semaphore(0)
        self.at: Optional[int] = None  # index of the group this forward waits in front of
        self.x: Optional[torch.Tensor] = None  # the activation it arrived with
        self.seen = set()
        self.started = False
        self.finished = False
        self.cancel = False
        self.error: Optional[BaseException] = None

    def run(self) -> None:
        self.go.acquire()
        try:
            if not self.cancel:
                self.feeder._tls.worker = self
                if self.device.type == "cuda":  # current device, stream and autocast state are per thread
                    torch.cuda.set_device(self.device)
                    enabled, dtype = self.feeder._autocast
                    with torch.cuda.stream(self.stream), torch.no_grad(), \
                            torch.autocast("cuda", dtype=dtype, enabled=enabled):
                        self.feeder._forward_fn(self.feeder._model, self.item)
                else:
                    with torch.no_grad():
                        self.feeder._forward_fn(self.feeder._model, self.item)
        except _EarlyExit:
            pass
        except BaseException as e:  # handed to the feeder's thread
            self.error = e
        finally:
            self.at, self.x, self.finished = None, None, True
            self.feeder._back.release()

    def wait_at(self, gi: int, x: torch.Tensor) -> None:
        """Called from the hook, in this thread: hand control back and sleep until resumed."""
        self.at, self.x = gi, x
        self.feeder._back.release()
        self.go.acquire()
        self.at, self.x = None, None
        if self.cancel:
            raise _EarlyExit


class LayerwiseFeeder:
    def __init__(self, model: nn.Module, dataset, subset_size: int, trace, target_names: List[str],
                 short_circuit: bool = True, suspend: Optional[bool] = None):
        from nncf_amd.quantize_model import _forward, _model_device, _to_device

        self._model, self._forward_fn = model, _forward
        device = _model_device(model)
        self._samples = []
        for i, item in enumerate(dataset.get_inference_data()):
            if i >= subset_size:
                break
            self._samples.append(_to_device(item, device))
        if not self._samples:
            raise ValueError("GPTQ needs a non-empty calibration dataset")
        mods = dict(model.named_modules())
        order = {n: i for i, n in enumerate(trace.order)}
        targets = sorted((n for n in target_names if n in trace.act_key), key=lambda n: order[n])
        # groups of Linears reading the same activation, in execution order of their first member
        # (the trace's activation key predates AWQ: members whose inputs AWQ rescales differently, or only some of
        # them, do NOT see the same tensor any more -- the Hessian is taken behind the member's own input multiply, so
        # such members get groups of their own; the reference builds one Hessian per node, gptq.py:118-140)
        def input_scale_key(name):
            sq = getattr(mods[name], "awq_input_scale", None)
            val = getattr(sq, "_scale_value", None)
            return None if val is None else hash(val.detach().float().cpu().numpy().tobytes())

        groups: Dict[tuple, List[str]] = {}
        for n in targets:
            groups.setdefault((trace.act_key[n], input_scale_key(n)), []).append(n)
        self._groups = sorted(groups.values(), key=lambda g: order[g[0]])
        self._mods = mods
        # decoder blocks: children of ModuleLists that hold targets
        self._block_of: Dict[str, Optional[str]] = {}
        block_names = []
        if short_circuit:
            for lname, lmod in mods.items():
                if isinstance(lmod, nn.ModuleList):
                    for idx in range(len(lmod)):
                        block_names.append(f"{lname}.{idx}" if lname else str(idx))
        for n in targets:
            owner = [b for b in block_names if n.startswith(b + ".")]
            self._block_of[n] = min(owner, key=len) if owner else None  # outermost stack
        self._patched: List[nn.Module] = []
        self._cache: Optional[List] = None
        self._cur = 0
        self._suspend = DEFAULT_SUSPEND if suspend is None else suspend
        self._device = device
        self._tls = threading.local()
        self._back = threading.Semaphore(0)

    # -------------------------------------------------------------------------------------------------
    def _run(self, hook_module: nn.Module, pre_hook=None, post_hook=None) -> None:
        h = (hook_module.register_forward_pre_hook(pre_hook, with_kwargs=True) if pre_hook is not None
             else hook_module.register_forward_hook(post_hook))
        try:
            with torch.no_grad():
                for i, item in enumerate(self._samples):
                    self._cur = i
                    try:
                        self._forward_fn(self._model, item)
                    except _EarlyExit:
                        pass
        finally:
            h.remove()

    def _finish_block(self, block_name: str) -> None:
        """Outputs of the finished block for every sample; from now on the block (and every block finished before
        it) returns them instead of running."""
        block = self._mods[block_name]
        outs: List = [None] * len(self._samples)

        def post(mod, args, output):
            outs[self._cur] = _detach(output)
            raise _EarlyExit

        self._run(block, post_hook=post)
        if any(o is None for o in outs):  # the block did not run for some sample: leave it unpatched
            return
        self._cache = outs
        block.forward = lambda *a, **k: self._cache[self._cur]
        self._patched.append(block)

    def _restore(self) -> None:
        for b in self._patched:
            if "forward" in b.__dict__:
                del b.__dict__["forward"]
        self._patched = []
        self._cache = None

    # ------------------------------------------------------------------------------------------------- suspend
    def _resume(self, w: _SuspendedForward) -> None:
        """Runs `w` until it waits in front of the next group or ends; only one thread is ever runnable."""
        if not w.started:
            w.started = True
            w.start()
        w.go.release()
        self._back.acquire()
        if w.error is not None:
            raise w.error

    def _iter_suspended(self) -> Iterator[Tuple[List[str], torch.Tensor, WCTensorStatistic]]:
        stream = torch.cuda.current_stream(self._device) if self._device.type == "cuda" else None
        self._autocast = (torch.is_autocast_enabled("cuda"), torch.get_autocast_dtype("cuda"))
        workers = [_SuspendedForward(self, item, self._device, stream) for item in self._samples]
        handles = []

        def make_pre(gi):
            def pre(mod, args, kwargs):
                w = getattr(self._tls, "worker", None)
                if w is None or gi in w.seen:  # somebody else's forward / the module runs again in this forward
                    return
                w.seen.add(gi)
                w.wait_at(gi, args[0] if args else next(iter(kwargs.values())))
            return pre

        try:
            for gi, group in enumerate(self._groups):
                handles.append(self._mods[group[0]].register_forward_pre_hook(make_pre(gi), with_kwargs=True))
            for gi, group in enumerate(self._groups):
                first = self._mods[group[0]]
                k = first.in_features
                acc = HessianAccumulator(k, first.weight.device, chunk_tokens=FEEDER_CHUNK_TOKENS)
                stats = WCTensorStatistic()
                for w in workers:
                    # a forward that reaches the groups in another order than the trace only skips what it missed
                    while not w.finished and (w.at is None or w.at < gi):
                        self._resume(w)
                    if w.finished or w.at != gi:
                        continue
                    x = w.x
                    if x.dtype == torch.float64:
                        x = x.float()
                    bs = x.shape[0] if x.ndim == 3 else 1  # gptq.py:226-235
                    x2 = x.reshape(-1, k)
                    acc.add(x2, bs)  # buffered: ~32k tokens per MFMA SYRK launch
                    mean, _ = activation_stats(x2.reshape(1, -1, k), want_stats=False)
                    stats.mean_values.append(mean.to(x.dtype))
                    stats.shape_values.append(tuple(x.shape))
                if not stats.mean_values:
                    # no calibration forward reached these Linears: an all-zero Hessian would mark every column
                    # dead and GPTQ would zero the whole weight
                    msg = f"GPTQ: no calibration sample reached {group}; exclude them (ignored_scope) or fix the dataset"
                    raise RuntimeError(msg)
                yield group, acc.result(), stats
        finally:
            for w in workers:  # the rest of the model is not needed (or the consumer stopped early)
                if w.started and not w.finished:
                    w.cancel = True
                    w.go.release()
                    self._back.acquire()
            for w in workers:
                if w.started:
                    w.join()
            for h in handles:
                h.remove()

    def __iter__(self) -> Iterator[Tuple[List[str], torch.Tensor, WCTensorStatistic]]:
        if self._suspend:
            yield from self._iter_suspended()
            return
        try:
            for gi, group in enumerate(self._groups):
                first = self._mods[group[0]]
                k = first.in_features
                acc = HessianAccumulator(k, first.weight.device, chunk_tokens=FEEDER_CHUNK_TOKENS)
                stats = WCTensorStatistic()

                def pre(mod, args, kwargs, acc=acc, stats=stats, k=k):
                    x = args[0] if args else next(iter(kwargs.values()))
                    if x.dtype == torch.float64:
                        x = x.float()
                    bs = x.shape[0] if x.ndim == 3 else 1  # gptq.py:226-235
                    x2 = x.reshape(-1, k)
                    acc.add(x2, bs)  # buffered: ~32k tokens per MFMA SYRK launch
                    mean, _ = activation_stats(x2.reshape(1, -1, k), want_stats=False)
                    stats.mean_values.append(mean.to(x.dtype))
                    stats.shape_values.append(tuple(x.shape))
                    raise _EarlyExit

                self._run(first, pre_hook=pre)
                if not stats.mean_values:
                    msg = f"GPTQ: no calibration sample reached {group}; exclude them (ignored_scope) or fix the dataset"
                    raise RuntimeError(msg)
                yield group, acc.result(), stats
                # the caller has replaced the weights of `group` by now; close the block if this was its last group
                blk = self._block_of.get(group[0])
                nxt = self._groups[gi + 1] if gi + 1 < len(self._groups) else None
                if blk is not None and nxt is not None and self._block_of.get(nxt[0]) != blk:
                    self._finish_block(blk)
        finally:
            self._restore()
