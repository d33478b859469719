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


def owner_of(parts: List[List[int]]) -> Dict[int, int]:
    return {i: r for r, p in enumerate(parts) for i in p}


def _flatten(tensors: Sequence[torch.Tensor]) -> torch.Tensor:
    if not tensors:
        return torch.empty(0, dtype=torch.uint8)
    return torch.cat([t.contiguous().view(torch.uint8).reshape(-1) for t in tensors])


def all_gather_payloads(local: Sequence[torch.Tensor], specs: Sequence[Sequence[Tuple[Tuple[int, ...], torch.dtype]]],
                        group=None) -> List[List[torch.Tensor]]:
    """Every rank contributes its list of tensors; returns, on every rank, the lists of all ranks.
    specs[r] = [(shape, dtype), ...] of rank r's tensors (known everywhere: it follows from the partition).
    One collective: payloads are padded to the longest rank and gathered with all_gather_into_tensor."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)

    def nbytes(spec):
        n = 0
        for shape, dtype in spec:
            numel = 1
            for s in shape:
                numel *= s
            n += numel * torch.empty(0, dtype=dtype).element_size()
        return n

    sizes = [nbytes(s) for s in specs]
    pad = max(sizes) if sizes else 0
    pad = (pad + 15) // 16 * 16
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
            numel = 1
            for s in shape:
                numel *= s
            nb = numel * torch.empty(0, dtype=dtype).element_size()
            items.append(recv[off:off + nb].view(dtype).reshape(shape))
            off += nb
        out.append(items)
    return out


def compress_sharded(weights: Sequence[torch.Tensor], config, compress_fn, gather: bool = True, group=None):
    """weights: the same ordered list (shapes at least) on every rank; each rank compresses only the tensors it owns
    with compress_fn(list_of_tensors, config) -> list of CompressedWeight-like objects (tensor, scale, zero_point).
    Returns {index: (tensor, scale, zero_point)} for the owned tensors, or for ALL tensors when gather=True."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
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
