"""Multi-GPU sharding of the lowering path: one process per GPU (torchrun), weight tensors are independent work
items (the reference walks them sequentially, largest first: algorithm.py:1198-1211), so they are assigned to
ranks by longest-processing-time-first and compressed without any data-path collective.  The only exchange is the
optional assembly of the compressed payload on every rank: ONE all_gather of a flat uint8 buffer per rank
(RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests)."""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import torch
import torch.distributed as dist


def partition_lpt(sizes: Sequence[int], world_size: int) -> List[List[int]]:
    """Greedy LPT: items sorted by size descending, each to the currently lightest rank.  Deterministic
    (ties broken by index / rank), identical on every rank."""
    order = sorted(range(len(sizes)), key=lambda i: (-int(sizes[i]), i))
    loads = [0] * world_size
    parts: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        parts[r].append(i)
        loads[r] += int(sizes[i])
    return parts


def partition_by_expert(expert_of: Sequence[int], world_size: int) -> List[List[int]]:
    """BASELINE configs[4] ("C5 uses expert index -> rank"): item i belongs to expert expert_of[i] and goes
    to rank expert % world_size, so the w1 / w2 / w3 of one expert (and the same expert of every layer) stay on one
    rank -- the placement an expert-parallel model already has.  Within a rank the items keep their list order."""
    parts: List[List[int]] = [[] for _ in range(world_size)]
    for i, e in enumerate(expert_of):
        parts[int(e) % world_size].append(i)
    return parts


def owner_of(parts: List[List[int]]) -> Dict[int, int]:
    return {i: r for r, p in enumerate(parts) for i in p}


SLOT_ALIGN = 16  # every tensor of a flat payload starts on a 16-byte boundary (view(dtype) needs element alignment)


def _slot(nbytes: int) -> int:
    return (nbytes + SLOT_ALIGN - 1) // SLOT_ALIGN * SLOT_ALIGN


def _spec_nbytes(shape, dtype) -> int:
    numel = 1
    for s in shape:
        numel *= s
    return numel * torch.empty(0, dtype=dtype).element_size()


def _flatten(tensors: Sequence[torch.Tensor]) -> torch.Tensor:
    """Concatenation of the tensors' bytes, each padded to its 16-byte slot."""
    if not tensors:
        return torch.empty(0, dtype=torch.uint8)
    pieces = []
    for t in tensors:
        b = t.contiguous().view(torch.uint8).reshape(-1) if t.numel() else torch.empty(0, dtype=torch.uint8, device=t.device)
        pieces.append(b)
        gap = _slot(b.numel()) - b.numel()
        if gap:
            pieces.append(torch.zeros(gap, dtype=torch.uint8, device=t.device))
    return torch.cat(pieces)


def all_gather_payloads(local: Sequence[torch.Tensor], specs: Sequence[Sequence[Tuple[Tuple[int, ...], torch.dtype]]],
                        group=None) -> List[List[torch.Tensor]]:
    """Every rank contributes its list of tensors; returns, on every rank, the lists of all ranks.
    specs[r] = [(shape, dtype), ...] of rank r's tensors (known everywhere: it follows from the partition).
    One collective: payloads are padded to the longest rank and gathered with all_gather_into_tensor."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)

    def nbytes(spec):
        return sum(_slot(_spec_nbytes(shape, dtype)) for shape, dtype in spec)

    sizes = [nbytes(s) for s in specs]
    pad = max(sizes) if sizes else 0  # a multiple of SLOT_ALIGN, so every rank's block starts aligned too
    dev = local[0].device if local else torch.device("cpu")
    send = torch.zeros(pad, dtype=torch.uint8, device=dev)
    flat = _flatten(local).to(dev)
    assert flat.numel() == sizes[rank], (flat.numel(), sizes[rank])
    send[:flat.numel()] = flat
    recv = torch.empty(world * pad, dtype=torch.uint8, device=dev)
    if pad:
        dist.all_gather_into_tensor(recv, send, group=group)
    out: List[List[torch.Tensor]] = []
    for r in range(world):
        off = r * pad
        items = []
        for shape, dtype in specs[r]:
            nb = _spec_nbytes(shape, dtype)
            items.append(recv[off:off + nb].view(dtype).reshape(shape))
            off += _slot(nb)
        out.append(items)
    return out


def compress_sharded(weights: Sequence[torch.Tensor], config, compress_fn, gather: bool = True, group=None,
                     expert_of: Sequence[int] = None):
    """weights: the same ordered list (shapes at least) on every rank; each rank compresses only the tensors it owns
    with compress_fn(list_of_tensors, config) -> list of CompressedWeight-like objects (tensor, scale, zero_point).
    Placement: LPT on the element count, or -- when expert_of[i] names the expert of weight i (MoE expert weights,
    BASELINE configs[4]) -- expert e on rank e % world.
    Returns {index: (tensor, scale, zero_point)} for the owned tensors, or for ALL tensors when gather=True."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if expert_of is not None:
        if len(expert_of) != len(weights):
            raise ValueError("compress_sharded: expert_of must name one expert per weight")
        parts = partition_by_expert(expert_of, world)
    else:
        parts = partition_lpt([w.numel() for w in weights], world)
    mine = parts[rank]
    results = compress_fn([weights[i] for i in mine], config) if mine else []
    local = {i: (cw.tensor, cw.scale, cw.zero_point) for i, cw in zip(mine, results)}
    if not gather or world == 1:
        return local
    # every rank derives every other rank's tensor specs by exchanging a tiny metadata object once
    meta = [[(tuple(t.shape), t.dtype) for trip in (local[i] for i in mine) for t in trip if t is not None]]
    has_zp = [[local[i][2] is not None for i in mine]]
    all_meta: List = [None] * world
    all_zp: List = [None] * world
    dist.all_gather_object(all_meta, meta[0], group=group)
    dist.all_gather_object(all_zp, has_zp[0], group=group)
    flat_local = [t for i in mine for t in local[i] if t is not None]
    gathered = all_gather_payloads(flat_local, all_meta, group=group)
    out = {}
    for r in range(world):
        it = iter(gathered[r])
        for i, zp_flag in zip(parts[r], all_zp[r]):
            tensor, scale = next(it), next(it)
            zp = next(it) if zp_flag else None
            out[i] = (tensor, scale, zp)
    return out


def shard_samples(n_samples: int, group=None) -> List[int]:
    """Calibration samples of this rank when statistics collection is data-parallel: sample i goes to rank
    i mod world (round-robin keeps the token counts balanced when the set is sorted by length)."""
    return list(range(dist.get_rank(group), n_samples, dist.get_world_size(group)))


def all_gather_sample_statistics(local: Sequence[torch.Tensor], n_samples: int, group=None) -> List[torch.Tensor]:
    """Data-parallel statistics collection for AWQ / scale estimation: every rank reduced the samples
    `shard_samples(n_samples)` of ONE activation to a [K] token mean each (collectors.py:512-516 keeps them as a
    list, one entry per sample); returns the full list in the original sample order on every rank, which is what
    `process_stats` (activation_stats.py:20-57) expects.  One all_gather_into_tensor of a padded [per_rank, K]
    block in the statistics' own dtype: bit-identical to a single-rank collection, nothing is summed."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    mine = shard_samples(n_samples, group)
    assert len(local) == len(mine), (len(local), len(mine))
    per_rank = -(-n_samples // world)
    # K and dtype are the same everywhere; a rank without samples learns them from the gathered metadata
    meta: List = [None] * world
    dist.all_gather_object(meta, (tuple(local[0].shape), local[0].dtype, str(local[0].device)) if local else None,
                           group=group)
    shape, dtype, _ = next(m for m in meta if m is not None)
    dev = local[0].device if local else torch.device("cuda", torch.cuda.current_device()) \
        if dist.get_backend(group) == "nccl" else torch.device("cpu")
    send = torch.zeros((per_rank,) + shape, dtype=dtype, device=dev)
    if local:
        send[:len(local)] = torch.stack(list(local))
    recv = torch.empty((world * per_rank,) + shape, dtype=dtype, device=dev)
    dist.all_gather_into_tensor(recv, send, group=group)
    return [recv[(i % world) * per_rank + i // world] for i in range(n_samples)]


# ---------------------------------------------------------------------------------------------------------------------
# Layer-sharded data-aware compression (BASELINE configs[3]: GPTQ with layers sharded across the GPUs of a node,
# configs[2]: AWQ / scale estimation).  The unit of work is what shares one activation: the Linears that read it have
# ONE Hessian / one set of statistics (gptq.py:118-171 walks nodes one by one; q/k/v and gate/up see the same input).
# Units are independent given their activations, so they are spread by LPT on an operation count and every rank
# loads / streams only the activations of the units it owns.  Nothing is reduced across ranks (Hessians stay on their
# owner); the only exchange is the final all-gather of packed codes + scales (+ zero points).
def run_sharded(costs: Sequence[float], work_fn, gather: bool = True, group=None, chunk: int = 1) -> Dict:
    """costs[i] = relative cost of unit i (same list on every rank); work_fn(i) -> {key: tuple of tensors / None}
    is called on the owner of unit i only (chunk > 1: work_fn(list of up to `chunk` owned units), for work that runs
    several units side by side).  Returns the union of all ranks' dictionaries (gather=True) or this rank's own.
    Keys must be unique across units and picklable; one all_gather_object of metadata (keys, shapes, dtypes) + ONE
    all_gather_into_tensor of the padded flat payload."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    scale = max([float(c) for c in costs] + [1.0])
    parts = partition_lpt([int(float(c) / scale * (1 << 40)) + 1 for c in costs], world)
    local: Dict = {}
    mine = sorted(parts[rank])  # unit order: neighbouring units (a layer's) share a chunk
    calls = [mine[c:c + chunk] for c in range(0, len(mine), chunk)] if chunk > 1 else mine
    for i in calls:
        for key, val in work_fn(i).items():
            if key in local:
                raise ValueError(f"run_sharded: key {key!r} returned twice")
            local[key] = tuple(val)
    if not gather or world == 1:
        return local
    keys = list(local)
    meta = [(k, [None if t is None else (tuple(t.shape), t.dtype) for t in local[k]]) for k in keys]
    all_meta: List = [None] * world
    dist.all_gather_object(all_meta, meta, group=group)
    flat = [t for k in keys for t in local[k] if t is not None]
    specs = [[sd for _, slots in m for sd in slots if sd is not None] for m in all_meta]
    if not flat:  # this rank owns nothing: it still takes part in the collective, on the device the others use
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        flat = [torch.empty(0, dtype=torch.uint8, device=dev)]
        specs[rank] = [((0,), torch.uint8)]
    gathered = all_gather_payloads(flat, specs, group=group)
    out: Dict = {}
    for r in range(world):
        it = iter(gathered[r])
        for k, slots in all_meta[r]:
            out[k] = tuple(None if sd is None else next(it) for sd in slots)
    return out


def gptq_unit_cost(k: int, tokens: int, out_features: Sequence[int]) -> float:
    """Operation count of one GPTQ unit: Hessian K^2 T (symmetric half, MFMA), Cholesky-with-inverse (2/3) K^3, and
    per weight the column sweep + trailing updates N K^2."""
    return float(k) * k * tokens + (2.0 / 3.0) * float(k) ** 3 + sum(float(n) * k * k for n in out_features)


def gptq_sharded(units, config, load_unit, gather: bool = True, group=None, damp_percent: float = 0.1,
                 block_size: int = 128, backend=None, concurrent_units: int = 4) -> Dict:
    """GPTQ over a list of units, layer-sharded (gptq.py:82-171 without the cross-layer replay: every unit comes with
    its own activations, as in BASELINE configs[3]'s synthetic calibration set).
      units[i] = {"k": K, "tokens": T, "weights": [(name, out_features), ...]}   -- the same list on every rank
      load_unit(i) -> (activations: list of [.., T_j, K] tensors, weights: {name: [N, K] tensor}) on the owner only
    Returns {name: (packed codes uint8, scale fp32, zero point int32 | None)}.
    `backend` (test hook) supplies calculate_hessian / prepare_hinv / quantize_weights / pack.
    concurrent_units: a rank takes its units in chunks of this many (a decoder layer has four) through
    gptq.quantize_units(overlap="units"): all Hessians of the chunk, then every unit's Hinv + column sweeps on a HIP
    stream of its own (a chunk's activations, Hessians and Hinvs are resident together)."""
    from nncf_amd.parameters import CompressedWeight, as_config

    config = as_config(config)
    costs = [gptq_unit_cost(u["k"], u["tokens"], [n for _, n in u["weights"]]) for u in units]
    if backend is None and concurrent_units > 1:
        from nncf_amd import gptq as _gptq, weight_lowering as _wl

        def work_chunk(indices):
            def loaded():
                for i in indices:
                    xs, ws = load_unit(i)
                    yield xs, {name: ws[name] for name, _ in units[i]["weights"]}

            def pack(name, w, q, scale, zp):
                cw = _wl.compress_weight_packed(q.to(w.dtype), (1,), config, CompressedWeight(None, scale, zp))
                return cw.tensor, scale, zp

            return _gptq.quantize_units(loaded(), config, damp_percent, block_size, overlap="units", on_weight=pack)

        return run_sharded(costs, work_chunk, gather=gather, group=group, chunk=concurrent_units)
    if backend is None:
        from nncf_amd import gptq as _gptq, weight_lowering as _wl

        class backend:  # noqa: N801
            calculate_hessian = staticmethod(lambda xs: _gptq.calculate_hessian(xs)[0])
            prepare_hinv = staticmethod(_gptq.prepare_hinv)

            @staticmethod
            def quantize_weights(w32, hinv, dead):
                return _gptq.quantize_weights(None, w32, config, hinv=hinv, dead=dead, damp_percent=damp_percent,
                                              block_size=block_size)

            @staticmethod
            def pack(q, scale, zp, dtype):
                cw = _wl.compress_weight_packed(q.to(dtype), (1,), config, CompressedWeight(None, scale, zp))
                return cw.tensor

    def work(i):
        xs, ws = load_unit(i)
        hessian = backend.calculate_hessian(xs)
        del xs
        hinv, dead = backend.prepare_hinv(hessian, damp_percent)
        del hessian
        out = {}
        for name, _ in units[i]["weights"]:
            w = ws[name]
            q, scale, zp = backend.quantize_weights(w.to(torch.float32).contiguous(), hinv, dead)
            out[name] = (backend.pack(q, scale, zp, w.dtype), scale, zp)
        return out

    return run_sharded(costs, work, gather=gather, group=group)


def scale_estimation_sharded(items, config, load_item, gather: bool = True, group=None, subset_size: int = 64,
                             initial_steps: int = 5, scale_steps: int = 5, backend=None) -> Dict:
    """Scale estimation (+ lowering with the estimated parameters) over independent weights, sharded by LPT on the
    element count (scale_estimation.py:97-158 walks the weights one by one; each needs only its own statistics).
      items[i] = (name, (out_features, in_features));  load_item(i) -> (weight [N, K], WCTensorStatistic) on the owner
    Returns {name: (packed codes, scale fp32, zero point | None)}."""
    from nncf_amd.parameters import CompressedWeight, as_config

    config = as_config(config)
    if backend is None:
        from nncf_amd import scale_estimation as _se, weight_lowering as _wl

        class backend:  # noqa: N801
            @staticmethod
            def estimate(w, stats):
                return _se.calculate_quantization_params(stats, w, (1,), config, -1, subset_size, initial_steps, scale_steps)

            @staticmethod
            def pack(w, scale, zp):
                cw = _wl.compress_weight_packed(w, (1,), config, CompressedWeight(None, scale, zp))
                return cw.tensor, cw.scale, cw.zero_point

    def work(i):
        name = items[i][0]
        w, stats = load_item(i)
        scale, zp = backend.estimate(w, stats)
        return {name: backend.pack(w, scale, zp)}

    return run_sharded([a * b for _, (a, b) in items], work, gather=gather, group=group)
