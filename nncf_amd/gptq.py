"""Host-side mirror of GPTQ's tensor math (src/nncf/quantization/algorithms/weight_compression/gptq.py:200-396) for
torch tensors on an AMD GPU.

  _calculate_hessian : H = (2/n) sum X^T X through the fp32-MFMA SYRK kernel (nncf_amd_hessian_accumulate).
  _quantize_weights  : dead columns, damping, Cholesky trio (torch.linalg on the device: dense LAPACK, not on the
                       custom-kernel path), then per 128-column block one fused HIP column sweep
                       (nncf_amd_gptq_block) + one trailing GEMM (torch.addmm), instead of K Python iterations.
"""
from __future__ import annotations

from functools import reduce
from operator import mul
from typing import List, Optional

import torch

from nncf_amd import _lib
from nncf_amd import scale_estimation as se
from nncf_amd.errors import UnsupportedModelError
from nncf_amd.parameters import WeightCompressionConfig, as_config
from nncf_amd.statistics import WCTensorStatistic, activations_to_wc_statistics


HESSIAN_CHUNK_TOKENS = 32768  # tokens per SYRK launch: the MFMA kernel reaches its rate from ~16k tokens on
HESSIAN_CHUNK_BYTES = 1 << 30
HESSIAN_LIST_MAX_TOKENS = 262144  # the item-list kernel copies nothing: the buffered items are only kept referenced
HESSIAN_LIST_MAX_BYTES = 4 << 30


def _list_chunk_tokens(k: int) -> int:
    """Tokens per launch of the item-list SYRK kernel.  With few 256 x 256 tiles of H (small K) the kernel splits the
    token axis over ~2048 blocks (csrc/wc_hessian.hip: launch_syrk16b_list) and every block ends with 65536 atomic
    adds, so a block should see ~16k tokens: 128 x [2048, 4096] bf16 items took 5.5 ms in 32k-token launches and
    3.8 ms in one launch (800 -> 1150 TFLOP/s); from K = 8192 on the default already gives a block that many."""
    nt = (k + 255) // 256
    pairs = nt * (nt + 1) // 2
    splits = max(1, -(-2048 // pairs))
    return min(HESSIAN_LIST_MAX_TOKENS, max(HESSIAN_CHUNK_TOKENS, 16384 * splits))


class HessianAccumulator:
    """Running average of gptq.py:237-240, `H <- H n/(n+b) + (2/(n+b)) X^T X`, for a stream of calibration items.
    The recursion telescopes to H = (2/n) sum X^T X (as real arithmetic), so the raw sums are accumulated and
    scaled once.  Items are buffered and concatenated into chunks of ~32k tokens per SYRK launch: one long launch
    runs the MFMA kernel at 3x the rate of sixteen 2048-token ones."""

    def __init__(self, k: int, device, chunk_tokens: Optional[int] = None):
        self.k, self.nsamples = k, 0
        self._chunk_tokens = chunk_tokens or HESSIAN_CHUNK_TOKENS
        self._list_tokens = chunk_tokens or _list_chunk_tokens(k)
        self.H = torch.zeros((k, k), dtype=torch.float32, device=device)
        self._buf: List[torch.Tensor] = []
        self._buf_tokens = self._buf_bs = 0
        self._final = False

    def add(self, x2d: torch.Tensor, batch_size: int) -> None:
        """x2d [tokens, K] (any float dtype the kernel takes); batch_size = the item's contribution to n."""
        if self._final:
            raise RuntimeError("HessianAccumulator: result() was already taken")
        if self._buf and self._buf[0].dtype != x2d.dtype:
            self.flush()
        self._buf.append(x2d)
        self._buf_tokens += x2d.shape[0]
        self._buf_bs += batch_size
        if x2d.dtype != torch.float32 and self.k >= 2048:  # the item-list kernel (flush): no concatenation copy
            # nothing is copied, but the buffered items stay referenced: bound them in bytes as well (K = 28672 bf16:
            # 4 GiB = 75k tokens per launch, far beyond where the kernel's rate has levelled off)
            if (self._buf_tokens >= self._list_tokens
                    or self._buf_tokens * self.k * x2d.element_size() >= HESSIAN_LIST_MAX_BYTES):
                self.flush()
        elif (self._buf_tokens >= self._chunk_tokens
                or self._buf_tokens * self.k * x2d.element_size() >= HESSIAN_CHUNK_BYTES):
            self.flush()

    def flush(self) -> None:
        """Adds the buffered items' raw X^T X to the upper triangle of H (alpha = beta = 1: no scaling pass and no
        symmetrising pass per chunk; `result` scales by 2 / n and mirrors once)."""
        if not self._buf:
            return
        import ctypes

        self.nsamples += self._buf_bs
        first = self._buf[0]
        if any(b.dtype != first.dtype for b in self._buf):  # add() flushes on a dtype change; the kernel reads ONE type
            raise RuntimeError("HessianAccumulator: buffered items of different dtypes")
        if len(self._buf) > 1 and first.dtype != torch.float32 and self.k >= 2048:
            # the MFMA kernel walks the list of items itself: no concatenation copy (20 ms per Llama-3-70B layer)
            items = [b.contiguous() for b in self._buf]
            n = len(items)
            xs = (ctypes.c_void_p * n)(*[b.data_ptr() for b in items])
            tk = (ctypes.c_int64 * n)(*[b.shape[0] for b in items])
            with torch.cuda.device(first.device):
                st = _lib.lib().nncf_amd_hessian_accumulate_list(_lib.ptr(self.H), n, xs, tk, _lib.dtype_id(first), self.k,
                                                                 1.0, _lib.stream_of(first))
            _lib.check(st)
        else:
            x = first if len(self._buf) == 1 else torch.cat(self._buf, dim=0)
            x = x.contiguous()
            with torch.cuda.device(x.device):
                st = _lib.lib().nncf_amd_hessian_accumulate_upper(_lib.ptr(self.H), _lib.ptr(x), _lib.dtype_id(x),
                                                                  x.shape[0], self.k, 1.0, 1.0, _lib.stream_of(x))
            _lib.check(st)
        self._buf, self._buf_tokens, self._buf_bs = [], 0, 0
        self._final = False

    def result(self) -> torch.Tensor:
        """H = (2 / n) sum X^T X, symmetric.  The accumulator is finished by this call."""
        self.flush()
        if not self._final and self.nsamples > 0:
            with torch.cuda.device(self.H.device):
                st = _lib.lib().nncf_amd_hessian_finalize(_lib.ptr(self.H), self.k, 2.0 / self.nsamples,
                                                          _lib.stream_of(self.H))
            _lib.check(st)
            self._final = True
        return self.H


def calculate_hessian(inputs: List[torch.Tensor], is_3d_weight: bool = False) -> torch.Tensor:
    """gptq.py:200-242 for a matmul node with non-transposed activations.  Returns fp32 [batch, K, K]."""
    first = inputs[0]
    _lib.require_gpu(first, "calculate_hessian")
    k = first.shape[-1]
    hb = 1 if not is_3d_weight else reduce(mul, first.shape[:-2], 1)
    accs = [HessianAccumulator(k, first.device) for _ in range(hb)]
    for inp in inputs:
        is_3d_act = inp.ndim == 3
        batch_size = 1 if is_3d_weight or not is_3d_act else inp.shape[0]
        if inp.dtype == torch.float64:
            inp = inp.float()
        xs = list(inp.reshape(hb, -1, k)) if is_3d_weight else [inp.reshape(-1, k)]
        for acc, x in zip(accs, xs):
            acc.add(x, batch_size)
    return torch.stack([acc.result() for acc in accs])


def quantize_weights(hessian: torch.Tensor, weight: torch.Tensor, config: WeightCompressionConfig,
                     inputs: Optional[List[torch.Tensor]] = None, damp_percent: float = 0.1, block_size: int = 128,
                     scale_estimation: bool = False, reduction_axes=(1,),
                     input_stats: Optional[WCTensorStatistic] = None, hinv: Optional[torch.Tensor] = None,
                     dead: Optional[torch.Tensor] = None):
    """gptq.py:244-379.  hessian [K, K] and weight [N, K] fp32 are modified in place like the reference does.
    Returns (quantized fp32 [N, K], scales [N, K/gs, 1] (or [N, 1] for group_size -1), zero_points | None).
    `input_stats` (token means over the FULL hidden axis, one per sample) may stand in for `inputs` when scale
    estimation is on: the mean of a column slice is the slice of the mean (gptq.py:320-321, 381-396).
    `hinv`, `dead` = a `prepare_hinv(hessian)` result to share between weights that read the same activation
    (`hessian` may then be None)."""
    config = as_config(config)
    _lib.require_gpu(weight, "quantize_weights")
    if tuple(reduction_axes) != (1,) or weight.ndim != 2:
        raise UnsupportedModelError("nncf_amd GPTQ expects a [out, in] weight reduced along axis 1")
    if block_size > 128:
        raise UnsupportedModelError("nncf_amd GPTQ block_size must be <= 128")
    W = weight
    assert W.dtype == torch.float32 and W.is_contiguous()
    N, K = W.shape
    if hinv is None:
        hinv, dead = prepare_hinv(hessian, damp_percent)
    W.masked_fill_(dead.unsqueeze(0), 0.0)  # W[:, dead] = 0 (gptq.py:271) without the host round trip of a mask index
    gs = config.group_size if config.group_size != -1 else K
    ngroups = K // gs

    asym = config.is_integer and config.is_asym_mode
    Q = torch.zeros_like(W)
    scales = torch.empty((N, ngroups), dtype=torch.float32, device=W.device)
    zps = torch.empty((N, ngroups), dtype=torch.int32, device=W.device) if asym else None
    use_se = scale_estimation and config.is_integer and config.num_bits == 4
    mode_id = _lib.MODE_ID[config.mode.value]
    L = _lib.lib()
    block_cfg = WeightCompressionConfig(mode=config.mode, group_size=-1)
    # Trailing update (gptq.py:363), two levels: after each 128-column block the error goes into the rest of its
    # SUPER-block only (k = 128 GEMM over <= 896 columns); the columns beyond get the errors of all eight blocks in
    # ONE k = 1024 GEMM when the super-block is done.  Same sums, regrouped: the k = 128 update of the whole
    # remainder streams W once per block (4 N K^2 / 128 bytes: 210 GB for 8192 x 28672), this form once per 1024.
    #
    # Group parameters are read from W[:, c:c+gs] with every EARLIER block's error applied (gptq.py:312-333, 363).
    # A group that starts inside a super-block (after its first column) and runs past its end s2 would see columns
    # [s2, c+gs) without this super-block's pending errors, so the near range of such a super-block is extended to
    # the end of that group (`near_end`): those columns take the per-block k = 128 updates like the super-block's
    # own columns, and the far GEMM starts behind them.  Group sizes that divide the super-block (32/64/128/256/...)
    # or are multiples of it never straddle: near_end == s2.
    super_cols = SUPER_BLOCKS * block_size
    with torch.cuda.device(W.device):
        stream = _lib.stream_of(W)
        for s1 in range(0, K, super_cols):
            s2 = min(s1 + super_cols, K)
            near_end = near_range_end(s1, s2, gs, K)
            errs = torch.empty((N, s2 - s1), dtype=torch.float32, device=W.device) if near_end < K else None
            for i1 in range(s1, s2, block_size):
                i2 = min(i1 + block_size, s2)
                count = i2 - i1
                if use_se:
                    # gptq.py:319-327: parameters of every group that STARTS in this block, from the current W
                    for c in range(i1, i2):
                        if c % gs:
                            continue
                        if input_stats is not None:
                            stats = WCTensorStatistic([m[..., c:c + gs] for m in input_stats.mean_values],
                                                      [tuple(sh[:-1]) + (gs,) for sh in input_stats.shape_values])
                        else:
                            acts = [inp[..., c:c + gs] for inp in inputs]
                            stats = activations_to_wc_statistics(acts)
                        sc, zp = se.calculate_quantization_params(stats, W[:, c:c + gs].contiguous(), (1,), block_cfg)
                        scales[:, c // gs] = sc.reshape(N)
                        if asym:
                            zps[:, c // gs] = zp.reshape(N).to(torch.int32)
                err = torch.empty((N, count), dtype=torch.float32, device=W.device)
                st = L.nncf_amd_gptq_block(_lib.ptr(W), _lib.ptr(Q), _lib.ptr(err), _lib.ptr(hinv), N, K, i1, count, gs,
                                           mode_id, _lib.ptr(scales), _lib.ptr(zps), int(use_se), stream)
                _lib.check(st)
                if i2 < near_end:
                    W[:, i2:near_end].addmm_(err, hinv[i1:i2, i2:near_end], alpha=-1.0)
                if errs is not None:
                    errs[:, i1 - s1:i2 - s1] = err
            if errs is not None:
                _mm(W[:, near_end:], errs, hinv[s1:s2, near_end:], acc=W[:, near_end:], negate=True)
    out_scales = scales.reshape(N, ngroups, 1)
    out_zps = zps.reshape(N, ngroups, 1) if asym else None
    if config.group_size == -1:
        out_scales = out_scales.squeeze(-1)
        if out_zps is not None:
            out_zps = out_zps.squeeze(-1)
    return Q, out_scales, out_zps


def quantize_node_weight(weight: torch.Tensor, inputs: List[torch.Tensor], config: WeightCompressionConfig,
                         reduction_axes=(1,), damp_percent: float = 0.1, block_size: int = 128,
                         scale_estimation: bool = False):
    """The per-node body of GPTQ.apply (gptq.py:136-169) for a 2-D [out, in] or 3-D / MoE [experts, out, in] weight:
    Hessian(s) of the node's inputs, then `quantize_weights` per expert with that expert's Hessian and inputs
    (`inp[batch_idx]`, :157), results stacked on a leading axis.  Returns (quantized fp32 weight, scale,
    zero_point | None) shaped like the reference's CompressedWeight parameters."""
    config = as_config(config)
    _lib.require_gpu(weight, "quantize_node_weight")
    is_3d = weight.ndim == 3
    if weight.ndim not in (2, 3) or tuple(reduction_axes) != ((2,) if is_3d else (1,)):
        raise UnsupportedModelError("nncf_amd GPTQ expects an [(experts,) out, in] weight reduced along its last axis")
    w32 = weight.to(torch.float32)
    hessian = calculate_hessian(inputs, is_3d)
    wb = w32 if is_3d else w32.unsqueeze(0)
    qs, scs, zps = [], [], []
    for b in range(hessian.shape[0]):
        xs = [inp[b] for inp in inputs] if is_3d else inputs
        q, sc, zp = quantize_weights(hessian[b], wb[b].contiguous().clone(), config, xs, damp_percent, block_size,
                                     scale_estimation, (1,))
        qs.append(q)
        scs.append(sc)
        zps.append(zp)
    if not is_3d:
        return qs[0], scs[0], zps[0]
    return torch.stack(qs), torch.stack(scs), (torch.stack(zps) if None not in zps else None)


def quantize_units(units, config: WeightCompressionConfig, damp_percent: float = 0.1, block_size: int = 128,
                   overlap: str = "units", on_weight=None, max_streams: int = 4):
    """GPTQ over a sequence of independent units (gptq.py:118-171 for nodes that come with their own activations: the
    Linears reading one activation share ONE Hessian and ONE Hinv).  `units` yields (inputs, {name: weight [N, K]});
    returns {name: (quantized fp32 [N, K], scale, zero_point | None)} -- or whatever `on_weight(name, weight, q, scale,
    zp)` returns for each weight (e.g. the packed codes, so that the fp32 result does not stay alive).

    overlap (same kernels and the same order within a unit in every mode; only what may run side by side changes):
      "none"     one stream, unit after unit.
      "units"    the Hessians on the caller's stream, largest first (MFMA-bound, each fills the chip); every unit's
                 Hinv + column sweeps on a HIP stream of its own as soon as its Hessian is there: those are long chains of small dependent kernels (128-column base blocks, small
                 GEMMs of the recursion's lower levels, in-block sweeps) that leave most of the chip idle, so the
                 chains of different units interleave.  Nothing in a chain waits for the host (the Cholesky failure
                 flags are read once per batch).  Units are taken in batches of `max_streams`, so a batch's Hessians,
                 Hinvs and fp32 weight copies are what is resident at once.
      "hessian"  the Hessian of unit i + 1 on a second stream under unit i's chains.  Measured on a Llama-3-70B layer
                 (tools/bench_configs.py c4): 612 -> 606 ms, the SYRK's one-wave-per-SIMD blocks leave a second queue
                 no room on a CU until they retire; kept for memory-tight callers."""
    config = as_config(config)
    if overlap not in ("none", "units", "hessian"):
        raise ValueError(f"quantize_units: overlap must be 'none', 'units' or 'hessian', got {overlap!r}")
    out = {}

    def sweeps(weights, hinv, dead):
        for name, w in weights.items():
            w32 = w.to(torch.float32).contiguous()
            if w32 is w or w32.data_ptr() == w.data_ptr():
                w32 = w32.clone()  # quantize_weights works in place
            q, scale, zp = quantize_weights(None, w32, config, hinv=hinv, dead=dead, damp_percent=damp_percent,
                                            block_size=block_size)
            out[name] = on_weight(name, w, q, scale, zp) if on_weight is not None else (q, scale, zp)

    def run_unit_batch(batch):
        dev = batch[0][0][0].device
        main = torch.cuda.current_stream(dev)
        infos = torch.zeros((len(batch), 1), dtype=torch.int32, device=dev)
        # Hessians on the caller's stream, the largest first, an event behind each: a unit's chain starts when ITS
        # Hessian is done, under the Hessians of the smaller units (the start of a chain is its most latency-bound part:
        # the lowest levels of the recursion).  Llama-3-70B layer: 485-500 -> 472 ms; the smallest first instead: 499
        by_size = sorted(range(len(batch)), key=lambda i: -next(iter(batch[i][1].values())).shape[-1])
        pending, ready = [None] * len(batch), [None] * len(batch)
        for i in by_size:
            pending[i] = (calculate_hessian(batch[i][0])[0], batch[i][1])
            ready[i] = torch.cuda.Event()
            ready[i].record(main)
        # the longest chain first: its tail is what the others have to fit under.  Units are dealt onto at most
        # `max_streams` streams (longest-processing-time first on K^3): a process has few hardware queues
        # (GPU_MAX_HW_QUEUES, 4 by default), and streams that share one serialise on it.  Measured on a Llama-3-70B
        # layer (four units): 602 ms on one stream, 568 / 562 / 544 ms on 2 / 3 / 4; the Linears of ONE unit on streams
        # of their own on top of that (up to 7 streams) gave the gain back (588-598 ms)
        order = sorted(range(len(pending)), key=lambda i: -pending[i][0].shape[0])
        n_streams = max(1, min(max_streams, len(pending)))
        # the same stream objects for every batch of a process: the caching allocator keeps freed blocks per stream, so
        # fresh streams per batch left ~18 GB more reserved behind per Llama-3-70B layer (fresh 547 / 536 / 538 ms, kept
        # 544 / 548 / 555 in one process).  (A high-priority stream for the longest chain was measured and dropped:
        # 535 -> 555 ms.)
        streams = _side_streams(dev, n_streams)
        load = [0.0] * n_streams
        for i in order:
            hessian, weights = pending[i]
            k = min(range(n_streams), key=lambda j: load[j])
            load[k] += float(hessian.shape[0]) ** 3
            streams[k].wait_event(ready[i])
            with torch.cuda.stream(streams[k]):
                hinv, dead = prepare_hinv(hessian, damp_percent, info=infos[i])
                sweeps(weights, hinv, dead)
                del hinv
            hessian.record_stream(streams[k])
        for st in streams:
            main.wait_stream(st)
        check_hinv_info(infos)
        for v in out.values():  # results were allocated on the side streams: hand them to the caller's stream
            for t in (v if isinstance(v, (tuple, list)) else (v,)):
                if isinstance(t, torch.Tensor):
                    t.record_stream(main)
        return

    if overlap == "units":
        # batches of max_streams units: what is resident at once (Hessians, Hinvs, fp32 copies of the weights) stays
        # bounded however long the sequence is
        it = iter(units)
        while True:
            batch = [u for _, u in zip(range(max(1, max_streams)), it)]
            if not batch:
                return out
            run_unit_batch(batch)

    it = iter(units)
    dev = side = main = None

    def start_hessian(unit):
        nonlocal dev, side, main
        inputs, weights = unit
        if dev is None:
            dev = inputs[0].device
            if overlap == "hessian":
                main, side = torch.cuda.current_stream(dev), torch.cuda.Stream(dev)
        if overlap != "hessian":
            return calculate_hessian(inputs)[0], None, weights
        side.wait_stream(main)  # the inputs were produced on the caller's stream
        with torch.cuda.stream(side):
            hessian = calculate_hessian(inputs)[0]
            done = torch.cuda.Event()
            done.record(side)
        return hessian, done, weights

    nxt = next(it, None)
    pending = start_hessian(nxt) if nxt is not None else None
    while pending is not None:
        hessian, done, weights = pending
        nxt = next(it, None)
        pending = start_hessian(nxt) if nxt is not None else None  # queued before this unit's host-synchronising Hinv
        if done is not None:
            main.wait_event(done)
            hessian.record_stream(main)
        hinv, dead = prepare_hinv(hessian, damp_percent)
        del hessian
        sweeps(weights, hinv, dead)
        del hinv
    if side is not None:
        main.wait_stream(side)
    return out


_SIDE_STREAMS = {}


def _side_streams(dev: torch.device, n: int):
    """`n` side streams of `dev`, created once per process and device."""
    have = _SIDE_STREAMS.setdefault(torch.device(dev).index if torch.device(dev).index is not None
                                    else torch.cuda.current_device(), [])
    while len(have) < n:
        have.append(torch.cuda.Stream(dev))
    return have[:n]


SUPER_BLOCKS = 8  # 128-column blocks per far trailing update (see quantize_weights)


def near_range_end(s1: int, s2: int, gs: int, K: int) -> int:
    """End of the column range that must receive every block's error right after the block (not deferred to the
    super-block's far GEMM) for the super-block [s1, s2): s2 itself, unless a group that starts strictly inside
    the super-block straddles s2 -- then the end of that group (its parameters are computed from columns up to
    there while this super-block's errors are still pending, gptq.py:312-333)."""
    if s2 >= K:
        return K
    g0 = (s2 // gs) * gs  # start of the group that holds column s2
    if s1 < g0 < s2:
        return min(g0 + gs, K)
    return s2
CHOLINV_BASE = 128  # blocks up to this size go to nncf_amd_cholinv_block (one workgroup)


TRI_SPLIT_MIN = 1024  # triangular-aware products split down to this size (measured: 306 -> 231 ms at K = 28672)

BF16X3_MIN_FLOPS = 3e10  # products from this size on run on the bf16 matrix cores (six-fold three-way split operands)


def _split3(x: torch.Tensor, contract_cols: bool, order: int, negate: bool = False) -> torch.Tensor:
    """nncf_amd_split3_bf16: the exact three-way bf16 split of an fp32 matrix view, six-fold along the contraction
    index (order 0: left operand, order 1: right operand), rows of the result = contraction index."""
    r, c = x.shape
    k, w = (c, r) if contract_cols else (r, c)
    dst = torch.empty((6 * k, w), dtype=torch.bfloat16, device=x.device)
    _lib.check(_lib.lib().nncf_amd_split3_bf16(_lib.ptr(x), r, c, x.stride(0), int(contract_cols), order, int(negate),
                                               _lib.ptr(dst), w, _lib.stream_of(x)))
    return dst


def _bf16x3_route(m: int, k: int, n: int, A: torch.Tensor, B: torch.Tensor, out: torch.Tensor, tri: int = 0) -> bool:
    """Whether a product goes to the bf16 split kernels: large enough (a triangular one has half the flops), aligned,
    and addressable with the 32-bit offsets of the kernel's buffer reads."""
    flops = 2.0 * m * k * n * (0.5 if tri else 1.0)
    return (A.is_cuda and flops >= BF16X3_MIN_FLOPS and m % 8 == 0 and n % 8 == 0 and 12 * k * max(m, n) < 2 ** 32
            and A.dtype == B.dtype == out.dtype == torch.float32 and A.stride(1) == 1 and B.stride(1) == 1
            and out.stride(1) == 1)


def _mm(out: torch.Tensor, A: torch.Tensor, B: torch.Tensor, b_t: bool = False, acc: Optional[torch.Tensor] = None,
        negate: bool = False, tri: int = 0) -> torch.Tensor:
    """out = (acc | 0) +- A @ (B^T if b_t else B), fp32 in and out (`acc` may be `out`).  Large products run on the
    bf16 MFMA: both operands as their exact three-way bf16 split, six-fold along the contraction (k_split3_*), then
    ONE hand-written GEMM over the 6x longer contraction with fp32 accumulation (k_gemm16t, the Hessian kernel's tile
    engine; csrc/wc_hessian.hip) -- the six leading products of the fp32 product, as accurate as the fp32 GEMM
    (measured 5e-7..9e-6 against 4e-7..5e-6 of max|C|), at 1.17-1.32 PFLOP/s = 170-205 TFLOP/s of fp32 work where the
    library's fp32 GEMM does 119-153: far trailing update 8192 x 1024 x 27648 3.51 -> 2.44 ms incl. the splits, X T^T
    at n = 14336 38.6 -> 29.6 ms, n = 3584 0.77 -> 0.65 ms.  Smaller or unaligned products keep the fp32 GEMM.
    `tri` (bf16 route only; the fp32 GEMM computes everything): the triangular structure the kernel may exploit, see
    nncf_amd_gemm_bf16_tn_acc -- 1: B^T with B lower triangular, 2: B lower triangular, 3: A lower triangular, 4: only
    the lower triangle of `out` is wanted."""
    m, k = A.shape
    n = B.shape[0] if b_t else B.shape[1]
    if _bf16x3_route(m, k, n, A, B, out, tri):
        with torch.cuda.device(A.device):
            p6 = _split3(A, True, 0, negate)
            q6 = _split3(B, b_t, 1)
            if acc is None:
                out.zero_()
            elif acc is not out and (acc.data_ptr() != out.data_ptr() or acc.stride() != out.stride()):
                out.copy_(acc)
            _lib.check(_lib.lib().nncf_amd_gemm_bf16_tn_acc(_lib.ptr(out), out.stride(0), _lib.ptr(p6), m, _lib.ptr(q6), n,
                                                            m, n, 6 * k, tri, _lib.stream_of(out)))
        return out
    Bm = B.T if b_t else B
    if acc is None:
        if negate:
            return torch.addmm(out, A, Bm, beta=0.0, alpha=-1.0, out=out)  # (beta = 0: `out` is not read)
        return torch.matmul(A, Bm, out=out)
    return torch.addmm(acc, A, Bm, beta=1.0, alpha=-1.0 if negate else 1.0, out=out)


def _mm_dense_lowerT(X: torch.Tensor, T: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """X @ T^T for lower-triangular T, skipping T's zero block: with T = [[T11, 0], [T21, T22]],
    X T^T = [X1 T11^T, X1 T21^T + X2 T22^T] (3 of the 4 quarter products; recursively towards half the flops)."""
    n = T.shape[0]
    if out is None:
        out = torch.empty((X.shape[0], n), dtype=X.dtype, device=X.device)
    if _bf16x3_route(X.shape[0], n, n, X, T, out, tri=1):
        return _mm(out, X, T, b_t=True, tri=1)  # one launch, the zero tiles of T skipped in the kernel
    if n < TRI_SPLIT_MIN:
        return _mm(out, X, T, b_t=True)
    h = ((n // 2 + 127) // 128) * 128
    _mm_dense_lowerT(X[:, :h], T[:h, :h], out[:, :h])
    _mm_dense_lowerT(X[:, h:], T[h:, h:], out[:, h:])
    _mm(out[:, h:], X[:, :h], T[h:, :h], b_t=True, acc=out[:, h:])
    return out


def _mm_dense_lower(X: torch.Tensor, T: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """X @ T for lower-triangular T: [X1 T11 + X2 T21, X2 T22]."""
    n = T.shape[0]
    if out is None:
        out = torch.empty((X.shape[0], n), dtype=X.dtype, device=X.device)
    if _bf16x3_route(X.shape[0], n, n, X, T, out, tri=2):
        return _mm(out, X, T, tri=2)
    if n < TRI_SPLIT_MIN:
        return _mm(out, X, T)
    h = ((n // 2 + 127) // 128) * 128
    _mm_dense_lower(X[:, :h], T[:h, :h], out[:, :h])
    _mm_dense_lower(X[:, h:], T[h:, h:], out[:, h:])
    _mm(out[:, :h], X[:, h:], T[h:, :h], acc=out[:, :h])
    return out


def _mm_lower_dense(T: torch.Tensor, X: torch.Tensor, out: torch.Tensor, negate: bool = False) -> torch.Tensor:
    """+-T @ X for lower-triangular T: [[T11 X1], [T21 X1 + T22 X2]]."""
    n = T.shape[0]
    if _bf16x3_route(n, n, X.shape[1], T, X, out, tri=3):
        return _mm(out, T, X, negate=negate, tri=3)
    if n < TRI_SPLIT_MIN:
        return _mm(out, T, X, negate=negate)
    h = ((n // 2 + 127) // 128) * 128
    _mm_lower_dense(T[:h, :h], X[:h], out[:h], negate)
    _mm_lower_dense(T[h:, h:], X[h:], out[h:], negate)
    _mm(out[h:], T[h:, :h], X[:h], acc=out[h:], negate=negate)
    return out


def _syrk_lower_sub(C: torch.Tensor, A: torch.Tensor) -> torch.Tensor:
    """Lower triangle of C - A A^T, IN PLACE in C (the strictly upper part is left unspecified: only the lower triangle
    of a Schur complement is ever read, and C is the recursion's own copy of the matrix)."""

    def rec(C, A):
        m = C.shape[0]
        if _bf16x3_route(m, A.shape[1], m, A, A, C, tri=4):
            _mm(C, A, A, b_t=True, acc=C, negate=True, tri=4)
            return
        if m < TRI_SPLIT_MIN:
            _mm(C, A, A, b_t=True, acc=C, negate=True)
            return
        h = ((m // 2 + 127) // 128) * 128
        rec(C[:h, :h], A[:h])
        rec(C[h:, h:], A[h:])
        _mm(C[h:, :h], A[h:], A[:h], b_t=True, acc=C[h:, :h], negate=True)

    rec(C, A)
    return C


def _cholinv_base(A: torch.Tensor, Linv: torch.Tensor, info: torch.Tensor) -> None:
    """L^-1 of one block of at most CHOLINV_BASE rows: one workgroup, block in registers (csrc/wc_cholinv.hip),
    ~0.2 us per column instead of the ~2 us of the library potrf plus a triangular solve against the identity.
    `info` (int32 on the device) is raised when a pivot is not positive."""
    if A.dtype != torch.float32 or A.stride(1) != 1 or Linv.stride(1) != 1:
        raise ValueError("cholinv block: fp32 blocks with unit column stride expected")
    _lib.check(_lib.lib().nncf_amd_cholinv_block(_lib.ptr(A), A.stride(0), _lib.ptr(Linv), Linv.stride(0),
                                                 A.shape[0], _lib.ptr(info), _lib.stream_of(A)))


def _cholinv_lower(A: torch.Tensor, Linv: torch.Tensor, info: torch.Tensor) -> None:
    """Writes L^-1 (lower triangle; the caller zero-fills the rest) of the SPD matrix A = L L^T into `Linv`; only
    the lower triangle of A is read, and A is OVERWRITTEN (its trailing blocks become the Schur complements).  With
    A = [[A11, .], [A21, A22]]:
        I11 = rec(A11);  L21 = A21 I11^T;  S = A22 - L21 L21^T;  I22 = rec(S);  I21 = -I22 (L21 I11)
    i.e. everything above the base size is a plain fp32 GEMM (hipBLASLt runs them at ~150 TFLOP/s on MI355X, the
    library potrf / trsm chain at 35-60), and the inverse factor comes out of ONE sweep instead of a factorization
    plus a triangular solve against the identity.  The products with a triangular operand and the symmetric Schur
    complement skip their zero / redundant blocks."""
    n = A.shape[0]
    if n <= CHOLINV_BASE:
        _cholinv_base(A, Linv, info)
        return
    h = ((n // 2 + 127) // 128) * 128
    if h >= n:
        h = n // 2
    I11 = Linv[:h, :h]
    _cholinv_lower(A[:h, :h], I11, info)
    L21 = _mm_dense_lowerT(A[h:, :h], I11)
    S = _syrk_lower_sub(A[h:, h:], L21)
    _cholinv_lower(S, Linv[h:, h:], info)
    tmp = _mm_dense_lower(L21, I11)
    del L21
    _mm_lower_dense(Linv[h:, h:], tmp, Linv[h:, :h], negate=True)


_NOT_SPD = "linalg.cholesky: The factorization could not be completed because the input is not positive-definite"


def _upper_cholesky_of_inverse(hessian: torch.Tensor, info: Optional[torch.Tensor] = None) -> torch.Tensor:
    """U upper triangular with H^-1 = U^T U: what the reference gets from cholesky -> cholesky_inverse ->
    cholesky(upper=True) (gptq.py:283-285), from ONE factorization instead of two plus an inversion.  With J the
    index reversal, J H J = L L^T (plain Cholesky) gives H = R R^T for the upper triangular R = J L J, hence
    H^-1 = R^-T R^-1 = U^T U with U = R^-1 = J L^-1 J, and a Cholesky factor with positive diagonal is unique.  In
    fp32 this is at least as close to the exact U as the three-step chain (one rounding cascade less).  L^-1 comes
    from the GEMM-based recursion `_cholinv_lower`.
    `info` (int32 [1] on the device, zeroed by the caller): the failure flag is left there and NOT read, so that the
    call stays asynchronous (several factorizations in flight on different streams); the caller checks it once with
    `check_hinv_info`.  Without it the call synchronises and raises like the reference's cholesky."""
    A = torch.flip(hessian, (0, 1)).contiguous()
    Linv = torch.zeros_like(A)
    own = info is None
    if own:
        info = torch.zeros(1, dtype=torch.int32, device=A.device)
    _cholinv_lower(A, Linv, info)
    U = torch.flip(Linv, (0, 1)).contiguous()
    if own:
        check_hinv_info(info)  # one synchronisation per factorization; same failure the reference's cholesky raises
    return U


def check_hinv_info(info: torch.Tensor) -> None:
    if int(info.max().item()) != 0:
        raise torch.linalg.LinAlgError(_NOT_SPD)


def prepare_hinv(hessian: torch.Tensor, damp_percent: float = 0.1, info: Optional[torch.Tensor] = None):
    """gptq.py:269-295 up to `hessian_inv`: dead columns, damping, U = chol_upper(H^-1).  `hessian` [K, K] fp32 is
    modified in place like the reference does.  Returns (U, dead-column mask); Linears that read one activation
    share both.  Nothing here waits for the device when `info` is given (see `_upper_cholesky_of_inverse`)."""
    assert hessian.ndim == 2 and hessian.dtype == torch.float32, "Hessian should be 2D fp32"
    diag = hessian.diagonal()
    dead = diag == 0
    fixed = torch.where(dead, torch.ones((), dtype=diag.dtype, device=diag.device), diag)  # H[dead, dead] = 1  (:270)
    damp = damp_percent * torch.mean(fixed)
    diag.copy_(fixed + damp)
    return _upper_cholesky_of_inverse(hessian, info), dead


class GPTQ:
    """Parameter holder with the reference's constructor defaults (gptq.py:49-65) and the two tensor-level methods."""

    def __init__(self, damp_percent: float = 0.1, block_size: int = 128, subset_size: int = 128,
                 scale_estimation: bool = False):
        self._damp_percent, self._block_size = damp_percent, block_size
        self._subset_size, self._scale_estimation = subset_size, scale_estimation

    def _calculate_hessian(self, node, inputs: List[torch.Tensor], is_3d_weight: bool = False) -> torch.Tensor:
        return calculate_hessian(inputs, is_3d_weight)

    def _quantize_weights(self, wc_params, hessian: torch.Tensor, weight_tensor: torch.Tensor,
                          inputs: List[torch.Tensor]):
        return quantize_weights(hessian, weight_tensor, wc_params.compression_config, inputs, self._damp_percent,
                                self._block_size, self._scale_estimation, wc_params.reduction_axes)

    activations_to_wc_statistics = staticmethod(activations_to_wc_statistics)
