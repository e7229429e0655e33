from .irreps import Irrep, Irreps, MulIr, as_irreps
from .wigner import wigner_3j, cg_nonzeros

__all__ = ["Irrep", "Irreps", "MulIr", "as_irreps", "wigner_3j", "cg_nonzeros"]
