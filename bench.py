#!/usr/bin/env python3
"""
bench.py -- throughput of the weight-compression lowering hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload llama3-8b-int4sym-g128]

A "step" is one pass of the fused lowering (group min/max -> scale -> round -> nibble pack) over every linear
weight of a Llama-3-8B-shaped model (7 shapes x 32 layers, bf16, random init, resident in HBM), INT4_SYM
group_size=128, through the C ABI (nncf_amd_int_quantize_batched).  Metric: GB/s of bf16 weights compressed
(input bytes of the WHOLE model / wall time of the slowest rank).  N > 1: one process per GPU (torchrun); the
model's tensor list is cut by longest-processing-time-first (nncf_amd.distributed.partition_lpt, the partition the
product uses), every rank creates and compresses ONLY its shard, no data-path collective: strong scaling of one
model ("wall time of Llama-3-8B / 70B at 1/2/4/8 GPUs" = ms_per_step).  Without a launcher, `--gpus N` starts the
N rank processes itself (launch_ranks); under torchrun the ranks are already there.  For N > 1 the RCCL all-gather
that assembles the compressed payload on every rank is always timed as well: `value` is without it,
`value_gathered` with it, `ranks_seen` is what an all-reduce of 1 returns.  Mixtral workloads place expert e on rank
e % N; --scaling weak gives every rank a whole model of its own instead (round-1 behaviour).  After the
timed region the same step runs back to back for >= 2 s (`sustained_ms_per_step`).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

# [out, in] linear shapes per decoder layer: q, k, v, o, gate, up, down
SHAPES = {
    "llama3-8b": ([(4096, 4096), (1024, 4096), (1024, 4096), (4096, 4096), (14336, 4096), (14336, 4096),
                   (4096, 14336)], 32),
    "llama3-70b": ([(8192, 8192), (1024, 8192), (1024, 8192), (8192, 8192), (28672, 8192), (28672, 8192),
                    (8192, 28672)], 80),
    "mixtral-8x7b-experts": ([(14336, 4096), (14336, 4096), (4096, 14336)] * 8, 32),
    "tiny": ([(512, 1024), (256, 1024), (1024, 512)], 2),
}
WORKLOADS = {
    # name: (geometry, mode, group_size)
    "llama3-8b-int4sym-g128": ("llama3-8b", "int4_sym", 128),
    "llama3-8b-int4asym-g128": ("llama3-8b", "int4_asym", 128),
    "llama3-8b-int8asym": ("llama3-8b", "int8_asym", -1),
    "llama3-70b-int4sym-g128": ("llama3-70b", "int4_sym", 128),
    "mixtral-nf4-g32": ("mixtral-8x7b-experts", "nf4", 32),
    "mixtral-mxfp4-g32": ("mixtral-8x7b-experts", "mxfp4", 32),
    "tiny-int4sym-g128": ("tiny", "int4_sym", 128),
}


def algorithmic_bytes_per_elem(mode, gs, cols):
    """bf16 in (2 B) + packed 4-bit (0.5 B) or 8-bit (1 B) out + fp32 scale (+ int32 zp) per group."""
    g = gs if gs != -1 else cols
    out = 1.0 if mode.startswith("int8") else 0.5
    params = 4.0 / g * (2 if mode.endswith("asym") else 1)
    return 2.0 + out + params


class Job:
    """One rank's resident weights + preallocated outputs + C-ABI descriptor arrays."""

    def __init__(self, tensors, mode, gs, device, seed=0):
        """tensors: [(global index, (out, in)), ...] = this rank's shard of the model's tensor list."""
        import ctypes

        from nncf_amd import _lib

        self._lib, self._ct = _lib, ctypes
        self.mode, self.gs, self.device = mode, gs, device
        self.is_int = mode.startswith("int")
        four = not mode.startswith("int8")
        self.weights, self.outs, self.scales, self.zps = [], [], [], []
        gen = torch.Generator(device=device)
        for index, (o, i) in tensors:
            gen.manual_seed(seed * 100003 + index)  # a tensor's values depend on its index, not on who owns it
            w = (torch.randn(o, i, device=device, generator=gen, dtype=torch.float32) * 0.02).to(torch.bfloat16)
            g = gs if gs != -1 else i
            self.weights.append(w)
            self.outs.append(torch.empty(o * i // 2 if four else o * i, dtype=torch.uint8, device=device))
            self.scales.append(torch.empty(o * (i // g), dtype=torch.float32, device=device))
            self.zps.append(torch.empty(o * (i // g), dtype=torch.int32, device=device)
                            if mode.endswith("asym") else None)
        n = len(self.weights)
        self.n = n
        self.elems = sum(w.numel() for w in self.weights)
        self.in_bytes = 2 * self.elems
        self.alg_bytes = sum(w.numel() * algorithmic_bytes_per_elem(mode, gs, w.shape[1]) for w in self.weights)
        VP = ctypes.c_void_p * max(n, 1)
        I64 = ctypes.c_int64 * max(n, 1)
        self.c_w = VP(*[w.data_ptr() for w in self.weights])
        self.c_rows = I64(*[w.shape[0] for w in self.weights])
        self.c_cols = I64(*[w.shape[1] for w in self.weights])
        self.c_out = VP(*[t.data_ptr() for t in self.outs])
        self.c_scale = VP(*[t.data_ptr() for t in self.scales])
        self.c_zp = VP(*[t.data_ptr() if t is not None else None for t in self.zps]) if mode.endswith("asym") else None
        self.stream = torch.cuda.current_stream(device)

    def step(self):
        if self.n == 0:
            return
        L, ct = self._lib.lib(), self._ct
        s = ct.c_void_p(self.stream.cuda_stream)
        if self.is_int:
            four = not self.mode.startswith("int8")
            st = L.nncf_amd_int_quantize_batched(self.n, self.c_w, self._lib.BF16, self.c_rows, self.c_cols, self.gs,
                                                 self._lib.MODE_ID[self.mode], None if four else self.c_out,
                                                 self.c_out if four else None, self.c_scale, self.c_zp, s)
            self._lib.check(st)
        else:
            st = L.nncf_amd_float_quantize_batched(self.n, self.c_w, self._lib.BF16, self.c_rows, self.c_cols, self.gs,
                                                   self._lib.MODE_ID[self.mode], self.c_out, self.c_scale, None, s)
            self._lib.check(st)


def cpu_baseline(geometry, mode, gs, budget_s=12.0):
    """The reference's torch-CPU op chain (oracle/torch_chain.py, a port) on the host cores, bounded sample.
    Threads: the box's CPU share for one GPU (OMP_NUM_THREADS, 16 on the pool's boxes; os.cpu_count() there shows the
    whole machine and is reported next to it)."""
    from oracle.torch_chain import compress_float_torch, compress_int_torch

    shapes, _ = SHAPES[geometry]
    shapes = shapes[:7]
    visible = os.cpu_count() or 1
    threads = max(1, min(visible, int(os.environ.get("OMP_NUM_THREADS", "16"))))
    torch.set_num_threads(threads)
    if mode.startswith("int"):
        fn, what = (lambda w: compress_int_torch(w, mode, gs, pack=True)), "reference torch-CPU op chain + pack_int4"
    else:
        fn, what = (lambda w: compress_float_torch(w, mode, gs)), "reference torch-CPU op chain (fp32 quantile values, no packing: the reference's torch path has none)"
    gen = torch.Generator().manual_seed(0)
    ws = [(torch.randn(o, i, generator=gen) * 0.02).to(torch.bfloat16) for (o, i) in shapes]
    for w in ws[:2]:
        fn(w)  # warm-up (thread pool, allocator)
    done_bytes, t_total, passes = 0, 0.0, 0
    while t_total < budget_s and passes < 200:
        for w in ws:
            t0 = time.perf_counter()
            fn(w)
            t_total += time.perf_counter() - t0
            done_bytes += 2 * w.numel()
            if t_total > 2.5 * budget_s:
                break
        passes += 1
    return {"value": round(done_bytes / t_total / 1e9, 4), "unit": "GB/s", "cores": threads, "host_cpus_visible": visible,
            "kind": "port",
            "sample": f"{what} (oracle/torch_chain.py) on the {len(ws)} tensors of one {geometry} layer, bf16, {mode} "
                      f"g{gs}, {passes} passes, {t_total:.1f}s of CPU work on {threads} threads"}


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one GPU each, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* set as torchrun would), pass rank 0's JSON line through, fail if any rank fails.  Runs
    before anything in this process touches the GPU (no HIP call, no torch.cuda query that initialises it)."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), NNCF_AMD_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out)
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        sys.stderr.write(f"bench.py: ranks failed (rank, exit code): {bad}\n")
        sys.exit(1)
    sys.exit(0)


def shard_of(model, n_shapes, world, placement):
    """Indices into `model` per rank; within a rank the reference's order, largest weight first
    (algorithm.py:1198-1200), ties by position."""
    from nncf_amd.distributed import partition_by_expert, partition_lpt

    if placement == "expert":  # the three matrices of expert e (every layer) on rank e % N
        parts = partition_by_expert([(k % n_shapes) // 3 for k, _ in model], world)
    else:
        parts = partition_lpt([o * i for _, (o, i) in model], world)
    return [sorted(p, key=lambda k: (-model[k][1][0] * model[k][1][1], k)) for p in parts]


def dry_run(args):
    import torch.distributed as dist

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    geometry, mode, gs = WORKLOADS[args.workload]
    shapes, n_layers = SHAPES[geometry]
    n_layers = args.layers or n_layers
    model = [(layer * len(shapes) + j, sh) for layer in range(n_layers) for j, sh in enumerate(shapes)]
    placement = ("expert" if geometry.startswith("mixtral") else "lpt") if args.placement == "auto" else args.placement
    parts = shard_of(model, len(shapes), world, placement)
    seen = 1
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        one = torch.ones(1, dtype=torch.int64)
        dist.all_reduce(one)
        seen = int(one.item())
        mine = torch.tensor([sum(model[k][1][0] * model[k][1][1] for k in parts[rank])], dtype=torch.int64)
        dist.all_reduce(mine)
        assert int(mine.item()) == sum(o * i for _, (o, i) in model)
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "ranks_seen": seen, "placement": placement,
                          "tensors_per_rank": [len(p) for p in parts],
                          "elements_per_rank": [sum(model[k][1][0] * model[k][1][1] for k in p) for p in parts],
                          "launcher": "bench.py --gpus N (own rank processes)"
                          if os.environ.get("NNCF_AMD_BENCH_SPAWNED") else "external (torchrun)"}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="llama3-8b-int4sym-g128", choices=sorted(WORKLOADS))
    ap.add_argument("--layers", type=int, default=0, help="override the layer count (debug)")
    ap.add_argument("--gather", action="store_true", help="(kept for old command lines: N > 1 always times the gather)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="strong: ONE model sharded over the ranks (default); weak: a whole model per rank")
    ap.add_argument("--placement", choices=["auto", "lpt", "expert"], default="auto",
                    help="tensor -> rank: LPT on the element count, or expert e -> rank e %% N (auto: expert for the "
                         "mixtral workloads, SURVEY 8e)")
    ap.add_argument("--sustained-s", type=float, default=2.0, help="seconds of back-to-back steps after the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dry-run", action="store_true",
                    help="rendezvous only (no GPU): the ranks connect over gloo, count themselves and rank 0 prints "
                         "the partition it would run -- checks the launcher and the placement on a CPU-only host")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        launch_ranks(args.gpus, sys.argv[1:])  # does not return
    if args.dry_run:
        return dry_run(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # one GPU per rank is the real thing (RCCL).  With fewer GPUs than ranks (rehearsal of the N > 1 code path on a
    # one-GPU box) the ranks share the cards and synchronise over gloo; such a run is marked in the JSON line.
    n_dev = torch.cuda.device_count()
    rehearsal = world > 1 and n_dev < world
    device = torch.device("cuda", local_rank % max(n_dev, 1))
    torch.cuda.set_device(device)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    coll_dev = torch.device("cpu") if rehearsal else device

    geometry, mode, gs = WORKLOADS[args.workload]
    shapes, n_layers = SHAPES[geometry]
    n_layers = args.layers or n_layers
    model = [(layer * len(shapes) + j, sh) for layer in range(n_layers) for j, sh in enumerate(shapes)]
    model_elems = sum(o * i for _, (o, i) in model)
    placement = args.placement
    if placement == "auto":
        placement = "expert" if geometry.startswith("mixtral") else "lpt"
    if args.scaling == "strong" and world > 1:
        mine = shard_of(model, len(shapes), world, placement)[rank]
        job = Job([model[k] for k in mine], mode, gs, device, seed=0)
    else:
        order = sorted(range(len(model)), key=lambda k: (-model[k][1][0] * model[k][1][1], k))
        job = Job([model[k] for k in order], mode, gs, device, seed=rank)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    ranks_seen = 1
    if dist is not None:  # how many ranks the collective backend really connects (the driver checks it against N)
        one = torch.ones(1, device=coll_dev, dtype=torch.int64)
        dist.all_reduce(one)
        ranks_seen = int(one.item())

    for _ in range(args.warmup):
        job.step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        job.step()
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)

    # kernel-side timing with HIP events on the launch stream (roofline leg)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev_steps = max(3, min(args.steps, 10))
    with torch.cuda.stream(job.stream):
        ev0.record(job.stream)
        for _ in range(ev_steps):
            job.step()
        ev1.record(job.stream)
    torch.cuda.synchronize()
    dev_ms_per_step = ev0.elapsed_time(ev1) / ev_steps

    # the exchange step of the path: every rank ends with the whole compressed model (packed codes + scales + zero
    # points of all tensors).  ONE all_gather_into_tensor of the padded flat payload, RCCL over xGMI; always timed
    # for N > 1 (the metric's end point is "gathered if multi-GPU", 8e: "with and without the gather").
    gather_ms, gather_bytes = None, None
    if dist is not None:
        payload = torch.cat([t.view(torch.uint8) for t in job.outs] + [t.view(torch.uint8) for t in job.scales]
                            + [t.view(torch.uint8) for t in job.zps if t is not None]
                            or [torch.empty(0, dtype=torch.uint8, device=device)])
        size = torch.tensor([payload.numel()], device=coll_dev, dtype=torch.int64)
        dist.all_reduce(size, op=dist.ReduceOp.MAX)  # shards differ by a few tensors: pad to the longest
        pad = (int(size.item()) + 15) // 16 * 16
        send = torch.zeros(pad, dtype=torch.uint8, device=device)
        send[:payload.numel()] = payload
        del payload
        if rehearsal:  # gloo: host buffers; exercises the code path, says nothing about xGMI
            send = send.cpu()
        out = torch.empty(world * pad, dtype=torch.uint8, device=send.device)
        gather_bytes = world * pad
        reps = 1 if rehearsal else 3
        dist.all_gather_into_tensor(out, send)  # untimed first call: communicator / channel set-up
        barrier()
        t1 = time.perf_counter()
        for _ in range(reps):
            dist.all_gather_into_tensor(out, send)
        barrier()
        gather_ms = max_over_ranks((time.perf_counter() - t1) / reps) * 1e3
        del out, send

    # sustained leg: the same step back to back for >= sustained_s seconds (DVFS settled), wall clock + HIP events
    sustained = None
    if args.sustained_s > 0 and job.n:
        n_sus = max(args.steps, int(args.sustained_s / max(dev_ms_per_step * 1e-3, 1e-6)) + 1)
        n_sus = min(n_sus, 200000)
        sev0, sev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        ts = time.perf_counter()
        sev0.record(job.stream)
        for _ in range(n_sus):
            job.step()
        sev1.record(job.stream)
        barrier()
        sus_wall = max_over_ranks(time.perf_counter() - ts)
        sustained = {"steps": n_sus, "seconds": round(sus_wall, 3), "ms_per_step": sus_wall / n_sus * 1e3,
                     "device_ms_per_step": sev0.elapsed_time(sev1) / n_sus}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        strong = args.scaling == "strong"
        total_in_bytes = 2 * model_elems if strong else job.in_bytes * world
        value = total_in_bytes / (elapsed / args.steps) / 1e9
        achieved = job.alg_bytes / (dev_ms_per_step * 1e-3) / 1e9  # rank 0's shard on rank 0's GPU
        if gs != -1:  # grouped modes: multi-tensor tile kernel, launches of <= MAX_BATCH tensors
            launches = -(-job.n // job._lib.MAX_BATCH)
            kernel = "wc::k_grouped_batch<%s, bf16, VPG=%d>" % (
                {"int4_sym": "IntPolicyT<INT4_SYM>", "int4_asym": "IntPolicyT<INT4_ASYM>", "nf4": "FloatPolicy<NF4>",
                 "mxfp4": "FloatPolicy<E2M1, MXFP4>"}.get(mode, mode), gs * 2 // 16)
        else:  # per-channel: multi-tensor row kernel, one series of launches per row-length class
            per_class = {}
            for w in job.weights:
                per_class[w.shape[1]] = per_class.get(w.shape[1], 0) + 1
            launches = sum(-(-c // job._lib.MAX_BATCH) for c in per_class.values())
            kernel = "wc::k_rows_batch<IntPolicyT<%s>, bf16> (%d row-length classes)" % (mode.upper(), len(per_class))
        # HBM bytes per launch: NOT measured by this run.  The figure is read from the committed rocprofv3 --pmc passes
        # of this same command (profiles/r*_traffic.json, made by tools/traffic_from_pmc.py) and labelled as such;
        # null for any other shape of run.
        traffic, traffic_source = None, None
        for name in ("r3_traffic.json", "r2_traffic.json"):
            try:
                with open(os.path.join(ROOT, "profiles", name)) as fh:
                    entry = json.load(fh).get(args.workload, {})
            except OSError:
                continue
            if not args.layers and world == 1 and entry.get("traffic_bytes_per_launch") is not None:
                traffic = entry["traffic_bytes_per_launch"]
                traffic_source = f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, " \
                                 "committed; not re-measured by this run)"
                break
        res = {
            "metric": "GB/s bf16 weights compressed (node)",
            "value": round(value, 2),
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{geometry}-shaped linears x{n_layers} layers, "
                                   f"bf16 random-init, {mode} group_size={gs}, fused quantize+pack, data-free RTN",
                       "tensors_in_model": len(model), "elements_in_model": model_elems,
                       "tensors_on_rank0": job.n, "elements_on_rank0": job.elems,
                       "partition": (("ONE model: expert e -> rank e % N (the three matrices of an expert stay together), "
                                      if placement == "expert" else "ONE model: tensor list cut by LPT over the ranks, ")
                                     + "each rank holds and compresses only its shard, largest weight first; no "
                                       "data-path collective inside the timed region") if strong else
                                    "each rank compresses its own whole model; no collective"},
            "model_wall_ms": round(ms_per_step, 4),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": kernel, "launches_per_step": launches,
                         "avg_launch_us": round(dev_ms_per_step * 1e3 / max(launches, 1), 3),
                         "algorithmic_bytes_per_launch": round(job.alg_bytes / max(launches, 1), 1),
                         "device_ms_per_step": round(dev_ms_per_step, 4)},
        }
        if sustained is not None:
            res["sustained_ms_per_step"] = round(sustained["ms_per_step"], 4)
            res["sustained"] = {"steps": sustained["steps"], "seconds": sustained["seconds"],
                                "device_ms_per_step": round(sustained["device_ms_per_step"], 4),
                                "value": round(total_in_bytes / (sustained["ms_per_step"] * 1e-3) / 1e9, 2),
                                "roofline_frac": round(job.alg_bytes / (sustained["device_ms_per_step"] * 1e-3) / 1e9
                                                       / HBM_PEAK_GBS, 4)}
        if world > 1:
            res["ranks_seen"] = ranks_seen
            res["launcher"] = "bench.py --gpus N (own rank processes)" if os.environ.get("NNCF_AMD_BENCH_SPAWNED") \
                else "external (torchrun)"
            res["collective_backend"] = "gloo (rehearsal)" if rehearsal else "nccl (RCCL)"
        if rehearsal:
            res["rehearsal"] = f"{world} ranks share {n_dev} GPU(s), gloo barrier: exercises the sharded code path, not a scaling number"
        if gather_ms is not None:
            res["allgather_ms"] = round(gather_ms, 3)
            res["allgather_bytes_per_rank_out"] = gather_bytes
            # the 8(d) metric with its multi-GPU end point: compressed model resident on EVERY rank
            res["value_gathered"] = round(total_in_bytes / ((ms_per_step + gather_ms) * 1e-3) / 1e9, 2)
            res["model_wall_ms_gathered"] = round(ms_per_step + gather_ms, 4)
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(geometry, mode, gs)
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
