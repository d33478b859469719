"""The RCCL (`nccl` backend) code path of nncf_amd.distributed on the one GPU a test box has: a world of size 1 still
initialises RCCL, picks the device for the collectives and runs all_gather_into_tensor on device buffers, with the
real HIP lowering as the per-rank work (the 2-rank logic is covered on CPU by tests/test_distributed_cpu.py)."""
import os
import socket

import numpy as np
import pytest
import torch

from oracle import wc_oracle as O
from tests.gpu_utils import host

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nccl_world():
    import torch.distributed as dist
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_nccl_backend_paths(nccl_world):
    from nncf_amd import distributed as D, weight_lowering as wl
    from nncf_amd.parameters import WeightCompressionConfig
    dist = nccl_world
    assert dist.get_backend() == "nccl"
    torch.manual_seed(0)
    cfg = WeightCompressionConfig("int4_asym", 64)
    ws = [(torch.randn(r, c, device="cuda") * 0.05).to(torch.bfloat16) for r, c in [(64, 256), (8, 1024), (130, 128)]]
    res = D.compress_sharded(ws, cfg, wl.compress_weights_packed_batched, gather=True)
    assert sorted(res) == [0, 1, 2]
    for i, w in enumerate(ws):
        c, s, z = O.compress_weight(host(w), (1,), "int4_asym", 64)
        t, sc, zp = res[i]
        assert t.is_cuda and np.array_equal(host(t).ravel(), O.pack_uint4(c).ravel())
        assert np.array_equal(host(sc), s) and np.array_equal(host(zp), z)
    # the generic engine with HIP work and the payload gather on device buffers
    out = D.run_sharded([3.0, 1.0], lambda i: {f"w{i}": (wl.compress_weight_packed(ws[i], (1,), cfg).tensor, None)})
    assert sorted(out) == ["w0", "w1"] and out["w0"][1] is None and out["w0"][0].is_cuda
    # data-parallel statistics: device selection for the nccl backend, bf16 payload keeps its bits
    means = [torch.randn(96, device="cuda").to(torch.bfloat16) for _ in range(5)]
    full = D.all_gather_sample_statistics(means, 5)
    assert len(full) == 5 and all(torch.equal(a, b) for a, b in zip(full, means)) and full[0].is_cuda
    # a GPTQ unit through the sharded driver with the real kernels
    g = torch.Generator(device="cuda").manual_seed(1)
    units = [{"k": 256, "tokens": 512, "weights": [("a", 48), ("b", 16)]}]
    xs = [torch.randn(1, 256, 256, device="cuda", generator=g).to(torch.bfloat16) for _ in range(2)]
    wts = {"a": torch.randn(48, 256, device="cuda", generator=g) * 0.05, "b": torch.randn(16, 256, device="cuda", generator=g) * 0.05}
    got = D.gptq_sharded(units, WeightCompressionConfig("int4_sym", 64), lambda i: (xs, wts))
    assert sorted(got) == ["a", "b"] and got["a"][2] is None
    assert tuple(got["a"][1].shape) == (48, 4, 1) and got["a"][0].numel() == 48 * 256 // 2


@pytest.mark.timeout(400)
def test_bench_gpus_2_from_plain_python():
    """The driver's command `python bench.py --gpus 2 ...` (no launcher): two rank processes of bench.py's own,
    n_gpus 2, both ranks seen by the collective, the gathered metric present.  On a one-GPU box the ranks share the
    card (marked `rehearsal`, gloo); on a multi-GPU box the same command runs RCCL."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "tiny-int4sym-g128",
                        "--steps", "3", "--warmup", "1", "--sustained-s", "0.05"], env=env, capture_output=True,
                       text=True, timeout=360)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2
    assert line["value"] > 0 and line["value_gathered"] > 0 and line["allgather_ms"] > 0
    assert line["config"]["tensors_on_rank0"] == 3 and line["sustained_ms_per_step"] > 0
    if torch.cuda.device_count() < 2:
        assert "rehearsal" in line
