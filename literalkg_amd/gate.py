"""Literal gates (LiteralE-style) on the HIP path.

Parameter names and shapes follow the reference's ``GateMul`` / ``Gate`` (gate.py:5-51) so that
checkpoints interchange.  The arithmetic differs in structure: the reference concatenates
``[x | num | txt]`` (an N x (D+302) copy) before ``g`` and runs four Linears; here the whole gate is ONE
launch from 16 384 rows on (``ops.fused_gate``): the ``g`` and ``z`` projections as one stacked tall GEMM
whose K-panels are x and the literal tables as they lie in memory, tanh / sigmoid / blend in its epilogue,
the result written straight into its column slot of the concatenated table; the backward is one blend
kernel + one two-panel data-gradient product + one long-k product per wide panel -- or, when the incoming
gradient is zero outside a few rows (one-layer models), the same products over those rows alone.  Below
16 384 rows: one accumulating panel GEMM per input panel (``ops.multi_linear``) and the blend kernel
(``ops.gate_blend``).
"""
import torch
import torch.nn as nn

from . import ops


class GateMul(nn.Module):
    """Both literal kinds: g over [x | num | txt], z over three bias-free projections (gate.py:5-28)."""

    def __init__(self, emb_size, num_lit_size, txt_lit_size):
        super().__init__()
        self.emb_size, self.num_lit_size, self.txt_lit_size = emb_size, num_lit_size, txt_lit_size
        self.g = nn.Linear(emb_size + num_lit_size + txt_lit_size, emb_size)
        self.gate_ent = nn.Linear(emb_size, emb_size, bias=False)
        self.gate_num_lit = nn.Linear(num_lit_size, emb_size, bias=False)
        self.gate_txt_lit = nn.Linear(txt_lit_size, emb_size, bias=False)
        self.gate_bias = nn.Parameter(torch.zeros(emb_size))

    def forward(self, x_ent, x_lit_num, x_lit_txt, out=None):
        d, n = self.emb_size, self.num_lit_size
        wg = self.g.weight
        if ops.gate_fusable(x_ent, (x_lit_num, x_lit_txt), d):     # one launch: stacked projections + blend epilogue
            return ops.fused_gate(x_ent, (x_lit_num, x_lit_txt), (wg[:, :d], wg[:, d:d + n], wg[:, d + n:]),
                                  (self.gate_ent.weight, self.gate_num_lit.weight, self.gate_txt_lit.weight),
                                  self.g.bias, self.gate_bias, out)
        gpre = ops.multi_linear((x_ent, x_lit_num, x_lit_txt), (wg[:, :d], wg[:, d:d + n], wg[:, d + n:]),
                                self.g.bias)
        zpre = ops.multi_linear((x_ent, x_lit_num, x_lit_txt),
                                (self.gate_ent.weight, self.gate_num_lit.weight, self.gate_txt_lit.weight),
                                self.gate_bias)
        return ops.gate_blend(x_ent, gpre, zpre, out)


class Gate(nn.Module):
    """One literal kind (gate.py:30-51)."""

    def __init__(self, emb_size, lit_size):
        super().__init__()
        self.emb_size, self.lit_size = emb_size, lit_size
        self.g = nn.Linear(emb_size + lit_size, emb_size)
        self.gate_ent = nn.Linear(emb_size, emb_size, bias=False)
        self.gate_lit = nn.Linear(lit_size, emb_size, bias=False)
        self.gate_bias = nn.Parameter(torch.zeros(emb_size))

    def forward(self, x_ent, x_lit, out=None):
        d = self.emb_size
        wg = self.g.weight
        if ops.gate_fusable(x_ent, (x_lit,), d):
            return ops.fused_gate(x_ent, (x_lit,), (wg[:, :d], wg[:, d:]), (self.gate_ent.weight, self.gate_lit.weight),
                                  self.g.bias, self.gate_bias, out)
        gpre = ops.multi_linear((x_ent, x_lit), (wg[:, :d], wg[:, d:]), self.g.bias)
        zpre = ops.multi_linear((x_ent, x_lit), (self.gate_ent.weight, self.gate_lit.weight), self.gate_bias)
        return ops.gate_blend(x_ent, gpre, zpre, out)
