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


def test_partition_by_expert_keeps_an_expert_on_one_rank():
    """configs[4] places expert e on rank e % N, so w1 / w2 / w3 of an expert (every layer) share a rank."""
    from nncf_amd.distributed import partition_by_expert
    n_layers, n_experts = 4, 8
    expert_of = [e for _ in range(n_layers) for e in range(n_experts) for _ in range(3)]
    for world in (1, 2, 4, 8):
        parts = partition_by_expert(expert_of, world)
        assert sorted(i for p in parts for i in p) == list(range(len(expert_of)))
        for r, p in enumerate(parts):
            assert {expert_of[i] % world for i in p} <= {r}
            assert len(p) == len(expert_of) // world
    assert partition_by_expert([], 2) == [[], []]
    assert partition_by_expert([5, 5, 6], 4) == [[], [0, 1], [2], []]


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
        # expert placement: weight i of expert i // 2 -> rank (i // 2) % world, same gathered result
        by_expert = compress_sharded(weights, "int4_asym", fake_compress, gather=False, expert_of=[0, 0, 1, 1, 2])
        ok = ok and sorted(by_expert) == ([0, 1, 4] if rank == 0 else [2, 3])
        res_e = compress_sharded(weights, "int4_asym", fake_compress, gather=True, expert_of=[0, 0, 1, 1, 2])
        ok = ok and all(torch.equal(res_e[i][0], res[i][0]) and torch.equal(res_e[i][1], res[i][1]) for i in res)
        # payloads whose byte counts are not multiples of 4 (packed int4 of 3 x 6 values = 9 bytes, odd int8 rows)
        # in front of fp32 / int32 tensors: every tensor has its own 16-byte slot in the flat buffer
        from nncf_amd.distributed import all_gather_payloads

        mine_odd = [torch.arange(9 + rank, dtype=torch.uint8), torch.full((3,), 1.5 + rank), torch.arange(5, dtype=torch.int8),
                    torch.arange(3, dtype=torch.int32) + rank]
        specs = [[((9 + r,), torch.uint8), ((3,), torch.float32), ((5,), torch.int8), ((3,), torch.int32)] for r in range(world)]
        got = all_gather_payloads(mine_odd, specs)
        for r in range(world):
            ok = ok and torch.equal(got[r][0], torch.arange(9 + r, dtype=torch.uint8)) \
                and torch.equal(got[r][1], torch.full((3,), 1.5 + r)) and torch.equal(got[r][2], torch.arange(5, dtype=torch.int8)) \
                and torch.equal(got[r][3], torch.arange(3, dtype=torch.int32) + r)
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


# ----------------------------------------------------------------------------------------------------- data-aware
def _oracle_gptq_backend(mode, gs, loaded):
    """torch / numpy stand-ins for the HIP-only GPTQ steps (test-only): the oracle does the arithmetic."""
    from oracle import wc_oracle as O

    class backend:  # noqa: N801
        @staticmethod
        def calculate_hessian(xs):
            return torch.from_numpy(O.gptq_hessian([x.numpy() for x in xs])[0])

        @staticmethod
        def prepare_hinv(hessian, damp):
            hinv, dead = O.gptq_hinv(hessian.numpy().copy(), damp)
            return torch.from_numpy(hinv), torch.from_numpy(dead)

        @staticmethod
        def quantize_weights(w32, hinv, dead):
            q, s, z = O.gptq_quantize_weights(None, w32.numpy().copy(), mode, gs, hinv=hinv.numpy(), dead=dead.numpy())
            return torch.from_numpy(q), torch.from_numpy(s), None if z is None else torch.from_numpy(z)

        @staticmethod
        def pack(q, scale, zp, dtype):
            codes, _, _ = O.do_integer_quantization(q.numpy(), mode, gs, (1,), scale.numpy(), None if zp is None else zp.numpy())
            return torch.from_numpy(O.pack_uint4(codes) if zp is not None else O.pack_int4(codes))

    return backend


def _gptq_case():
    rng = np.random.default_rng(7)
    units = [{"k": 64, "tokens": 40, "weights": [("l0.q", 16), ("l0.k", 8)]},
             {"k": 32, "tokens": 24, "weights": [("l0.o", 16)]},
             {"k": 64, "tokens": 40, "weights": [("l1.gate", 24), ("l1.up", 24)]},
             {"k": 96, "tokens": 30, "weights": [("l1.down", 8)]}]
    data = []
    for u in units:
        xs = [torch.from_numpy(rng.standard_normal((1, u["tokens"] // 2, u["k"])).astype(np.float32)) for _ in range(2)]
        ws = {n: torch.from_numpy((rng.standard_normal((o, u["k"])) * 0.05).astype(np.float32)) for n, o in u["weights"]}
        data.append((xs, ws))
    return units, data


def _worker_data_aware(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nncf_amd.distributed import gptq_sharded, gptq_unit_cost, partition_lpt, run_sharded, scale_estimation_sharded
        from nncf_amd.parameters import WeightCompressionConfig
        from oracle import wc_oracle as O

        mode, gs = "int4_asym", 32
        units, data = _gptq_case()
        loaded = []

        def load_unit(i):
            loaded.append(i)
            return data[i]

        res = gptq_sharded(units, WeightCompressionConfig(mode, gs), load_unit, backend=_oracle_gptq_backend(mode, gs, loaded))
        costs = [gptq_unit_cost(u["k"], u["tokens"], [n for _, n in u["weights"]]) for u in units]
        ok = sorted(loaded) == sorted(partition_lpt([int(c) for c in costs], world)[rank]) and 0 < len(loaded) < len(units)
        names = [n for u in units for n, _ in u["weights"]]
        ok = ok and sorted(res) == sorted(names)
        # every rank now holds every weight's payload, equal to a single-process run of the same chain
        bk = _oracle_gptq_backend(mode, gs, [])
        for u, (xs, ws) in zip(units, data):
            hinv, dead = bk.prepare_hinv(bk.calculate_hessian(xs), 0.1)
            for n, _ in u["weights"]:
                qw, sc, zp = bk.quantize_weights(ws[n].clone(), hinv, dead)
                packed = bk.pack(qw, sc, zp, torch.float32)
                ok = ok and torch.equal(res[n][0], packed) and torch.equal(res[n][1], sc) and torch.equal(res[n][2], zp)
        # scale estimation, sharded per weight; one item only: rank 1 owns nothing and still joins the collective
        from nncf_amd.statistics import WCTensorStatistic

        rng = np.random.default_rng(3)
        items = [("a", (16, 64)), ("b", (8, 64)), ("c", (32, 32))]
        ws = [torch.from_numpy((rng.standard_normal(s) * 0.05).astype(np.float32)) for _, s in items]
        stats = [WCTensorStatistic([torch.from_numpy(rng.standard_normal(s[1]).astype(np.float32)) for _ in range(6)],
                                   [(1, 9, s[1])] * 6) for _, s in items]

        class se_backend:  # noqa: N801
            @staticmethod
            def estimate(w, st):
                sc, zp = O.scale_estimation_params([m.numpy() for m in st.mean_values], st.shape_values, w.numpy(), (1,),
                                                   "int4_sym", 32, -1, 64, 5, 5)
                return torch.from_numpy(sc), None

            @staticmethod
            def pack(w, scale, zp):
                codes, _, _ = O.do_integer_quantization(w.numpy(), "int4_sym", 32, (1,), scale.numpy(), None)
                return torch.from_numpy(O.pack_int4(codes)), scale, None

        for sub in (items, items[:1]):
            got = scale_estimation_sharded(sub, WeightCompressionConfig("int4_sym", 32), lambda i: (ws[i], stats[i]),
                                           backend=se_backend)
            ok = ok and sorted(got) == sorted(n for n, _ in sub)
            for i, (n, _) in enumerate(sub):
                sc, _ = se_backend.estimate(ws[i], stats[i])
                ref = se_backend.pack(ws[i], sc, None)
                ok = ok and torch.equal(got[n][0], ref[0]) and torch.equal(got[n][1], sc) and got[n][2] is None
        own = run_sharded([1.0, 1.0, 1.0], lambda i: {("k", i): (torch.full((2,), float(i)), None)}, gather=False)
        ok = ok and 0 < len(own) < 3
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_layer_sharded_gptq_and_scale_estimation():
    """gptq_sharded / scale_estimation_sharded on two gloo ranks: units go to ranks by LPT on the operation count,
    each rank loads only its units' activations, nothing but the final payload is exchanged, and every rank ends
    with the same compressed weights as a single-process run (arithmetic by the oracle: no GPU here)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_data_aware, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=170) for _ in procs)
    for p in procs:
        p.join(30)
    assert results == {0: True, 1: True}
