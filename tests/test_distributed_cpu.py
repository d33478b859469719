"""N > 1 path on CPU: world_size-2 gloo processes exercise the LPT partition and the single all-gather that
assembles compressed payloads (the compression itself is replaced by the oracle here: no GPU in this test)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_partition_lpt_is_balanced_and_deterministic():
    from nncf_amd.distributed import owner_of, partition_lpt
    shapes = [(4096, 4096), (1024, 4096), (1024, 4096), (4096, 4096), (14336, 4096), (14336, 4096), (4096, 14336)] * 32
    sizes = [a * b for a, b in shapes]
    for world in (1, 2, 4, 8):
        parts = partition_lpt(sizes, world)
        assert sorted(i for p in parts for i in p) == list(range(len(sizes)))
        loads = [sum(sizes[i] for i in p) for p in parts]
        assert max(loads) / (sum(loads) / world) < 1.02
        assert parts == partition_lpt(sizes, world)
        assert len(owner_of(parts)) == len(sizes)
    assert partition_lpt([], 4) == [[], [], [], []]
    assert partition_lpt([5], 3) == [[0], [], []]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from types import SimpleNamespace

        from nncf_amd.distributed import compress_sharded
        from oracle import wc_oracle as O

        rng = np.random.default_rng(0)
        shapes = [(8, 64), (16, 128), (4, 32), (32, 64), (2, 256)]
        weights = [torch.from_numpy((rng.standard_normal(s) * 0.05).astype(np.float32)) for s in shapes]

        def fake_compress(ws, mode):
            out = []
            for w in ws:
                c, s, z = O.compress_weight(w.numpy(), (1,), mode, 32)
                out.append(SimpleNamespace(tensor=torch.from_numpy(O.pack_uint4(c)), scale=torch.from_numpy(s),
                                           zero_point=torch.from_numpy(z)))
            return out

        res = compress_sharded(weights, "int4_asym", fake_compress, gather=True)
        ok = sorted(res) == list(range(len(shapes)))
        for i, w in enumerate(weights):
            c, s, z = O.compress_weight(w.numpy(), (1,), "int4_asym", 32)
            t, sc, zp = res[i]
            ok = ok and np.array_equal(t.numpy(), O.pack_uint4(c)) and np.array_equal(sc.numpy(), s) \
                and np.array_equal(zp.numpy(), z)
        local_only = compress_sharded(weights, "int4_asym", fake_compress, gather=False)
        ok = ok and 0 < len(local_only) < len(shapes)
        # data-parallel statistics: 5 samples over 2 ranks (ragged), bf16 means keep their bits and their order
        from nncf_amd.distributed import all_gather_sample_statistics, shard_samples

        means = [torch.from_numpy(rng.standard_normal(48).astype(np.float32)).to(torch.bfloat16) for _ in range(5)]
        mine = shard_samples(5)
        ok = ok and mine == list(range(rank, 5, world))
        full = all_gather_sample_statistics([means[i] for i in mine], 5)
        ok = ok and len(full) == 5 and all(torch.equal(a, b) for a, b in zip(full, means))
        one = all_gather_sample_statistics([means[0]] if rank == 0 else [], 1)  # a rank without samples
        ok = ok and len(one) == 1 and torch.equal(one[0], means[0])
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gloo_all_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=100) for _ in procs)
    for p in procs:
        p.join(30)
    assert results == {0: True, 1: True}
