"""Irreps algebra used by the host side of the MI355X tensor-product path.

This is the small subset of ``e3nn.o3.Irreps`` / ``Irrep`` that the reference's
hot path touches (call sites e.g.
``e3_layers/nn/message_passing.py:75,157-187``, ``e3_layers/nn/pointwise.py:61-92``,
``e3_layers/utils/utils.py:87-96``).  It is pure Python: the objects only drive
path-table construction for the HIP kernels, they never touch tensors.

Layout reminder (``README.md:108-110`` of the reference): a tensor annotated
with irreps ``"4x1o+2x0e"`` has last dimension ``4*3 + 2*1`` and every
``mul x l`` block is stored ``[mul][2l+1]`` (channel slow, m fast).
"""
from __future__ import annotations

import re
from typing import Iterable, Iterator, List, NamedTuple, Sequence, Tuple, Union

_IRREP_RE = re.compile(r"^\s*(\d+)\s*([eoy])\s*$")


class Irrep(tuple):
    """One irreducible representation of O(3): degree ``l`` and parity ``p`` (+1 even, -1 odd)."""

    __slots__ = ()

    def __new__(cls, l: Union[int, str, "Irrep", Tuple[int, int]], p: int = None):
        if p is None:
            if isinstance(l, Irrep):
                return l
            if isinstance(l, str):
                m = _IRREP_RE.match(l)
                if m is None:
                    raise ValueError(f"cannot parse irrep {l!r}")
                deg = int(m.group(1))
                tag = m.group(2)
                par = {"e": 1, "o": -1, "y": (-1) ** deg}[tag]
                l, p = deg, par
            elif isinstance(l, (tuple, list)) and len(l) == 2:
                l, p = l
            else:
                raise ValueError(f"cannot build an Irrep from {l!r}")
        l = int(l)
        p = int(p)
        if l < 0 or p not in (1, -1):
            raise ValueError(f"invalid irrep l={l} p={p}")
        return tuple.__new__(cls, (l, p))

    @property
    def l(self) -> int:  # noqa: E743
        return self[0]

    @property
    def p(self) -> int:
        return self[1]

    @property
    def dim(self) -> int:
        return 2 * self[0] + 1

    def __repr__(self) -> str:
        return f"{self[0]}{'e' if self[1] == 1 else 'o'}"

    __str__ = __repr__

    def __mul__(self, other) -> Iterator["Irrep"]:
        """Selection rule: all ``Irrep(l, p1*p2)`` with ``|l1-l2| <= l <= l1+l2``, ascending."""
        other = Irrep(other)
        p = self.p * other.p
        for l in range(abs(self.l - other.l), self.l + other.l + 1):
            yield Irrep(l, p)

    def __rmul__(self, mul: int) -> "Irreps":
        return Irreps([(int(mul), self)])

    def is_scalar(self) -> bool:
        return self.l == 0 and self.p == 1


class MulIr(NamedTuple):
    mul: int
    ir: Irrep

    @property
    def dim(self) -> int:
        return self.mul * self.ir.dim

    def __repr__(self) -> str:
        return f"{self.mul}x{self.ir}"


class Irreps(tuple):
    """Direct sum of irreps with multiplicities, e.g. ``Irreps("64x0e+64x1o")``."""

    __slots__ = ()

    def __new__(cls, spec: Union[str, "Irreps", Irrep, Iterable] = ()):
        if isinstance(spec, Irreps):
            return spec
        items: List[MulIr] = []
        if isinstance(spec, Irrep):
            items.append(MulIr(1, spec))
        elif isinstance(spec, str):
            text = spec.strip()
            if text:
                for chunk in text.split("+"):
                    chunk = chunk.strip()
                    if "x" in chunk:
                        mul_s, ir_s = chunk.split("x", 1)
                        mul = int(mul_s)
                    else:
                        mul, ir_s = 1, chunk
                    if mul < 0:
                        raise ValueError(f"negative multiplicity in {spec!r}")
                    items.append(MulIr(mul, Irrep(ir_s)))
        else:
            for entry in spec:
                if isinstance(entry, MulIr):
                    items.append(entry)
                elif isinstance(entry, Irrep):
                    items.append(MulIr(1, entry))
                elif isinstance(entry, str):
                    items.extend(Irreps(entry))
                else:
                    mul, ir = entry
                    items.append(MulIr(int(mul), Irrep(ir)))
        return tuple.__new__(cls, items)

    # ---- sizes -------------------------------------------------------------
    @property
    def dim(self) -> int:
        return sum(mi.dim for mi in self)

    @property
    def num_irreps(self) -> int:
        return sum(mi.mul for mi in self)

    @property
    def lmax(self) -> int:
        if len(self) == 0:
            raise ValueError("lmax of empty Irreps")
        return max(mi.ir.l for mi in self)

    @property
    def ls(self) -> List[int]:
        return [mi.ir.l for mi in self for _ in range(mi.mul)]

    def slices(self) -> List[slice]:
        out, start = [], 0
        for mi in self:
            out.append(slice(start, start + mi.dim))
            start += mi.dim
        return out

    def offsets(self) -> List[int]:
        return [s.start for s in self.slices()]

    # ---- algebra -----------------------------------------------------------
    def simplify(self) -> "Irreps":
        """Merge *adjacent* entries that carry the same irrep; drop empty ones."""
        out: List[MulIr] = []
        for mi in self:
            if mi.mul == 0:
                continue
            if out and out[-1].ir == mi.ir:
                out[-1] = MulIr(out[-1].mul + mi.mul, mi.ir)
            else:
                out.append(mi)
        return Irreps(out)

    def remove_zero_multiplicities(self) -> "Irreps":
        return Irreps([mi for mi in self if mi.mul > 0])

    def sort(self):
        """Stable sort by ``(l, p)`` (odd before even at equal l, as tuple order gives).

        Returns ``(irreps, p, inv)`` with ``p[i]`` the new position of old entry ``i``
        and ``inv`` its inverse, the convention ``e3nn`` uses and
        ``e3_layers/nn/pointwise.py:69-75`` relies on.
        """
        order = sorted(range(len(self)), key=lambda i: (self[i].ir, i))
        inv = tuple(order)
        p = [0] * len(self)
        for new, old in enumerate(order):
            p[old] = new
        return Irreps([self[i] for i in order]), tuple(p), inv

    def __add__(self, other) -> "Irreps":
        return Irreps(tuple.__add__(self, Irreps(other)))

    def __radd__(self, other) -> "Irreps":
        return Irreps(other) + self

    def __mul__(self, n: int) -> "Irreps":
        if not isinstance(n, int):
            raise TypeError("Irreps can only be repeated by an int")
        return Irreps(tuple.__mul__(self, n))

    __rmul__ = __mul__

    def __eq__(self, other) -> bool:
        try:
            other = Irreps(other)
        except (ValueError, TypeError):
            return False
        return tuple.__eq__(self, other)

    def __ne__(self, other) -> bool:
        return not self.__eq__(other)

    def __hash__(self) -> int:
        return tuple.__hash__(self)

    def __contains__(self, ir) -> bool:
        try:
            ir = Irrep(ir)
        except (ValueError, TypeError):
            return False
        return any(mi.ir == ir for mi in self)

    def count(self, ir) -> int:
        ir = Irrep(ir)
        return sum(mi.mul for mi in self if mi.ir == ir)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return Irreps(tuple.__getitem__(self, i))
        return tuple.__getitem__(self, i)

    def __repr__(self) -> str:
        return "+".join(repr(mi) for mi in self)

    __str__ = __repr__

    @staticmethod
    def spherical_harmonics(lmax: int, p: int = -1) -> "Irreps":
        return Irreps([(1, Irrep(l, p ** l)) for l in range(lmax + 1)])


def as_irreps(x) -> Irreps:
    return x if isinstance(x, Irreps) else Irreps(x)
