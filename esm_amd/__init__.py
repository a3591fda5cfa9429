"""esm_amd: MI355X-native forward path for the ESM C / ESM3 transformer trunk.

Same Python surface as the reference (ESMC.forward / ESM3.forward / TransformerStack.forward, the reference's
state_dict keys); the arithmetic runs in hand-written gfx950 HIP kernels behind the C-ABI in include/esmk.h.
"""
__all__ = ["ESMC", "ESM3", "TransformerStack"]


def __getattr__(name):
    if name == "ESMC":
        from .esmc import ESMC
        return ESMC
    if name == "ESM3":
        from .esm3 import ESM3
        return ESM3
    if name == "TransformerStack":
        from .transformer_stack import TransformerStack
        return TransformerStack
    raise AttributeError(name)
