"""Small synthetic torch models shared by tools/gen_golden_model.py (which runs the REFERENCE on them in the build
container) and the model-level tests.  Own code; shapes chosen so that every weight-compression branch is hit:
embedding + last layer in backup precision, q/k/v sharing one activation, a gated MLP (MatMul -> Mul -> MatMul
AWQ pattern) and a plain Act -> MatMul MLP."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class RMSNorm(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))

    def forward(self, x):
        return self.weight * x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6)


class Attention(nn.Module):
    def __init__(self, hidden, heads):
        super().__init__()
        self.heads = heads
        self.q_proj = nn.Linear(hidden, hidden, bias=False)
        self.k_proj = nn.Linear(hidden, hidden, bias=False)
        self.v_proj = nn.Linear(hidden, hidden, bias=False)
        self.o_proj = nn.Linear(hidden, hidden, bias=False)

    def forward(self, x):
        b, t, h = x.shape
        d = h // self.heads
        q = self.q_proj(x).reshape(b, t, self.heads, d).transpose(1, 2)
        k = self.k_proj(x).reshape(b, t, self.heads, d).transpose(1, 2)
        v = self.v_proj(x).reshape(b, t, self.heads, d).transpose(1, 2)
        a = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) / (d ** 0.5), dim=-1)
        o = torch.matmul(a, v).transpose(1, 2).reshape(b, t, h)
        return self.o_proj(o)


class GatedMLP(nn.Module):
    def __init__(self, hidden, inter):
        super().__init__()
        self.gate_proj = nn.Linear(hidden, inter, bias=False)
        self.up_proj = nn.Linear(hidden, inter, bias=False)
        self.down_proj = nn.Linear(inter, hidden, bias=False)

    def forward(self, x):
        return self.down_proj(F.silu(self.gate_proj(x)) * self.up_proj(x))


class PlainMLP(nn.Module):
    def __init__(self, hidden, inter):
        super().__init__()
        self.fc1 = nn.Linear(hidden, inter, bias=True)
        self.fc2 = nn.Linear(inter, hidden, bias=True)

    def forward(self, x):
        return self.fc2(F.relu(self.fc1(x)))


class Block(nn.Module):
    def __init__(self, hidden, heads, inter, gated):
        super().__init__()
        self.norm1 = RMSNorm(hidden)
        self.attn = Attention(hidden, heads)
        self.norm2 = RMSNorm(hidden)
        self.mlp = GatedMLP(hidden, inter) if gated else PlainMLP(hidden, inter)

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


class TinyLM(nn.Module):
    """Decoder-style language model: embedding, `layers` blocks (gated MLPs first, the last block a plain MLP),
    final norm, untied lm_head."""

    def __init__(self, vocab=160, hidden=64, heads=4, inter=128, layers=3, tie=False):
        super().__init__()
        self.embed = nn.Embedding(vocab, hidden)
        self.blocks = nn.ModuleList([Block(hidden, heads, inter, gated=(i < layers - 1)) for i in range(layers)])
        self.norm = RMSNorm(hidden)
        self.lm_head = nn.Linear(hidden, vocab, bias=False)
        if tie:
            self.lm_head.weight = self.embed.weight

    def forward(self, input_ids):
        x = self.embed(input_ids)
        for blk in self.blocks:
            x = blk(x)
        return self.lm_head(self.norm(x))


def make_tiny_lm(seed=0, tie=False, **kw):
    """Deterministic init (CPU generator): per-layer weight magnitudes differ so that mixed-precision scores are
    well separated, a few input channels of every block are outliers so that AWQ has something to find."""
    torch.manual_seed(seed)
    m = TinyLM(tie=tie, **kw)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for i, (name, p) in enumerate(m.named_parameters()):
            if p.ndim == 2:
                p.copy_(torch.randn(p.shape, generator=g) * (0.02 + 0.01 * (i % 7)))
        m.embed.weight[:, 5] *= 8.0
        m.embed.weight[:, 41] *= 5.0
    m.eval()
    return m


def tiny_lm_dataset(n=6, tokens=12, vocab=160, seed=123):
    g = torch.Generator().manual_seed(seed)
    return [torch.randint(0, vocab, (1, tokens + (i % 3)), generator=g) for i in range(n)]


# name -> (model kwargs, compress_weights kwargs as plain strings / numbers).  Extra keys: data (calibration set of
# 6 samples instead of the single tracing sample), fallback (group_size_fallback_mode, min_adjusted_group_size 16),
# ignored (module names).  Shared by the golden generator (reference enums) and the GPU tests (nncf_amd enums).
TINY_CONFIGS = {
    "int8_asym": ({}, dict(mode="int8_asym")),
    "int8_sym": ({}, dict(mode="int8_sym")),
    "int8_asym_tied": (dict(tie=True), dict(mode="int8_asym")),
    "int4_sym_g32": ({}, dict(mode="int4_sym", group_size=32)),
    "int4_asym_g32_all": ({}, dict(mode="int4_asym", group_size=32, all_layers=True)),
    "int4_sym_gm1": ({}, dict(mode="int4_sym", group_size=-1)),
    "int4_asym_g16_r07_datafree": ({}, dict(mode="int4_asym", group_size=16, ratio=0.7,
                                           sensitivity_metric="weight_quantization_error")),
    "int4_sym_g32_r06_hawq": ({}, dict(mode="int4_sym", group_size=32, ratio=0.6, data=True,
                                      sensitivity_metric="hessian_input_activation")),
    "int4_sym_g32_r06_meanvar": ({}, dict(mode="int4_sym", group_size=32, ratio=0.6, data=True,
                                         sensitivity_metric="mean_activation_variance")),
    "int4_sym_g32_r06_maxvar": ({}, dict(mode="int4_sym", group_size=32, ratio=0.6, data=True,
                                        sensitivity_metric="max_activation_variance")),
    "int4_sym_g32_r06_meanmax": ({}, dict(mode="int4_sym", group_size=32, ratio=0.6, data=True,
                                         sensitivity_metric="mean_activation_magnitude")),
    "int4_asym_g32_se": ({}, dict(mode="int4_asym", group_size=32, data=True, scale_estimation=True)),
    "int4_sym_g32_se": ({}, dict(mode="int4_sym", group_size=32, data=True, scale_estimation=True)),
    "int4_asym_g32_awq": ({}, dict(mode="int4_asym", group_size=32, data=True, awq=True)),
    "int4_sym_g32_awq_datafree": ({}, dict(mode="int4_sym", group_size=32, awq=True)),
    "int4_asym_g32_r08_awq_se": ({}, dict(mode="int4_asym", group_size=32, ratio=0.8, data=True, awq=True,
                                         scale_estimation=True)),
    "int4_sym_g48_ignore": ({}, dict(mode="int4_sym", group_size=48, fallback="ignore")),
    "int4_sym_g48_adjust": ({}, dict(mode="int4_sym", group_size=48, fallback="adjust")),
    "int4_sym_g32_backup_none": ({}, dict(mode="int4_sym", group_size=32, backup_mode="none")),
    "int4_sym_g32_backup_int8sym": ({}, dict(mode="int4_sym", group_size=32, ratio=0.5, backup_mode="int8_sym")),
    "int4_sym_g32_ignored": ({}, dict(mode="int4_sym", group_size=32, ignored=["blocks.1.attn.q_proj",
                                                                               "blocks.0.mlp.down_proj"])),
}
