"""literalkg_amd -- MI355X (gfx950) native hot path of LiteralKG.

Host code is Python on PyTorch-ROCm and mirrors the reference's nn.Module
surface (``LiteralKG(args, n_entities, n_relations, A_in, num_lit, txt_lit)``,
``model(*input, device=, mode=)``, same ``state_dict`` keys); all arithmetic on
the path runs in hand-written HIP kernels behind the C ABI of
``include/literalkg_hip.h`` (``literalkg_amd/lib/liblkg_hip.so``).  There is no
CPU fallback: importing works anywhere, computing needs the library and a GPU.
"""
from .gate import Gate, GateMul          # noqa: F401
from .model import Aggregator, LiteralKG  # noqa: F401
from .graph import KGStructure           # noqa: F401

__all__ = ["LiteralKG", "Aggregator", "Gate", "GateMul", "KGStructure"]
