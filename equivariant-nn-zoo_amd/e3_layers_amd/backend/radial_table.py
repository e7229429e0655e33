"""Per-edge radial weights through a knot table (``csrc/e3k_rtable.hip``).

``weight = fc(edge_radial)`` (``e3_layers/nn/message_passing.py:74-79,93``) with ``edge_radial =
RadialBasisEncoding(edge_length)`` (``e3_layers/nn/embedding.py:210-219``) is a smooth function of one scalar per
edge.  Instead of pushing every edge through the MLP (the largest block of matrix work of a training step: forward,
dgrad and wgrad GEMMs of ``[E, 64] x [64, weight_numel]``), the MLP is evaluated on ``KNOTS + 1`` equidistant radii and
every edge interpolates quadratically between the three knots around it; the backward transposes the interpolation
(an ordered sum per knot over edges sorted by knot, no atomics) and then differentiates the MLP on the knots only.
Interpolation error (h^3): ~4e-8 relative at the default 2048 knots for the shipped models (bound and measurements per knot
count: tools/knot_error.py), below the ~1e-7 rounding error of evaluating the MLP in fp32 per edge.  The knot
count also sets how much of the table the 4 MB L2 of an XCD holds when the tensor-product kernels gather its rows
(e3k_tp_fwd_table): 2048 knots measured 0.23 ms / step faster than 4096 at 256 molecules.

Applies when ``edge_radial`` still carries the tag ``RadialBasisEncoding.forward`` puts on its output (so nothing was
concatenated to it: the diffusion configs mix bond types / residue offsets into the edge embedding and take the
per-edge path), the envelope is the polynomial cutoff (f is constant beyond ``r_max``), the radii need no gradient
(no forces / double backward: the composed per-edge ops serve those), and the batch has several times more edges than
the table has knots.  ``E3K_RADIAL_TABLE=0`` disables it.
"""
from __future__ import annotations

import os
import weakref
from typing import Optional

import torch

from . import lib as L
from . import ops
from .graph import build_topology      # (conv_block imports this module: keep it free of conv_block)

ENABLED = int(os.environ.get("E3K_RADIAL_TABLE", "1"))
KNOTS = int(os.environ.get("E3K_RADIAL_KNOTS", "2048"))          # intervals; KNOTS + 1 table rows
MIN_EDGES_PER_KNOT = float(os.environ.get("E3K_RADIAL_MIN_EDGES_PER_KNOT", "4"))      # below this the per-edge MLP is the cheaper one


class RadialSource:
    """What ``RadialBasisEncoding`` knows about an edge embedding it produced: the module, the radii, and the version
    counter of the embedding tensor when it was tagged (an in-place op on the embedding bumps it: the tag then no longer
    describes the tensor's contents and the table is not used)."""

    __slots__ = ("module", "r", "version", "_bins", "_knot_basis", "_stack", "__weakref__")

    def __init__(self, module, r: torch.Tensor, version: int = 0):
        self.module, self.r, self.version = weakref.ref(module), r, int(version)
        self._bins = None
        self._knot_basis = None
        self._stack = {}      # id(MessagePassing) -> (its radial MLP's rows on the knots, mode): nn/message_passing.py:_stack_rows

    def bins(self):
        """(centre knot int32 [E], offset t [E], CSR by knot) -- once per batch, shared by the layers."""
        if self._bins is None:
            r = L.f32c(self.r.detach().reshape(-1))
            e = r.numel()
            mod = self.module()
            bin2 = torch.empty(2, e, dtype=torch.int64, device=r.device)
            t = torch.empty(e, dtype=torch.float32, device=r.device)
            L.check(L.load().e3k_rtable_bin(L.ptr(r), e, float(mod.basis.r_max), KNOTS, L.ptr(bin2), L.ptr(t), L.stream_ptr()),
                    "e3k_rtable_bin")
            topo = build_topology(bin2, KNOTS + 1)
            self._bins = (topo.dst, t, topo.dst_ptr, topo.dst_perm)
        return self._bins

    def knot_basis(self):
        """RadialBasisEncoding on the knots (differentiable w.r.t. the Bessel frequencies) -- once per forward."""
        kb = self._knot_basis
        if kb is None or kb[1] != torch.is_grad_enabled():
            mod = self.module()
            b, c = mod.basis, mod.cutoff
            knots = _knots(float(b.r_max), self.r.device)
            basis = ops.radial_basis(knots, b.bessel_weights, b.r_max, b.r_min, c.p, b.one_over_r, c.cutoff.kind)
            kb = self._knot_basis = (basis, torch.is_grad_enabled())
        return kb[0]


_KNOT_CACHE = {}


def _knots(r_max: float, device) -> torch.Tensor:
    key = (r_max, KNOTS, str(device))
    k = _KNOT_CACHE.get(key)
    if k is None:
        k = torch.arange(KNOTS + 1, dtype=torch.float32) * (r_max / KNOTS)
        k[0] = 1e-6 * r_max / KNOTS            # sin(w r) / r at r = 0: evaluate next to it (f is smooth there)
        k = _KNOT_CACHE[key] = k.to(device)
    return k


def source_of(edge_radial) -> Optional[RadialSource]:
    return getattr(edge_radial, "_e3k_radial_src", None)


def applicable(edge_radial, w_last=None) -> bool:
    """``w_last``: the last-layer weight of the radial MLP that would run on the table -- its a-posteriori error guard
    (``guard`` below) can veto the table for that MLP."""
    if not ENABLED or not edge_radial.is_cuda:
        return False
    src = source_of(edge_radial)
    if src is None or src.module() is None or src.r.requires_grad:
        return False
    if edge_radial._version != src.version:
        return False                       # modified in place since RadialBasisEncoding produced it
    if w_last is not None and not guard_ok(w_last):
        return False
    return edge_radial.shape[0] >= MIN_EDGES_PER_KNOT * (KNOTS + 1)


# ---- a-posteriori guard of the interpolation error ---------------------------------------------------------------------
# Quadratic Lagrange interpolation on knots h apart is off by at most h^3 max|f(3)| / (9 sqrt 3) (f(3): third derivative);
# on the table itself h^3 f(3) is the third finite difference, so
#     err <= max|T[i+3] - 3 T[i+2] + 3 T[i+1] - T[i]| / (9 sqrt 3)
# -- read off the table rows the forward has just computed, relative to max|T|.  4e-8 for the shipped models at random
# init (8 Bessel functions through a smooth MLP); it grows like (frequency x weight scale)^3, so a 32-function basis, grown
# Bessel frequencies or large trained weights can push it towards the 1e-5 parity budget.  The bound is evaluated on the
# device the first time an MLP's table is built and every GUARD_EVERY-th time after (a few elementwise passes over 31 MB
# on the radial stream), copied to pinned memory without a sync and looked at on a later call: above GUARD_TOL the table
# is switched off for that MLP (per-edge evaluation from then on) with a warning.
GUARD_TOL = float(os.environ.get("E3K_RADIAL_TABLE_TOL", "1e-6"))
GUARD_EVERY = int(os.environ.get("E3K_RADIAL_TABLE_CHECK_EVERY", "64"))
_C3 = 1.0 / (9.0 * 3.0 ** 0.5)


class _Guard:
    __slots__ = ("calls", "pending", "ok", "last", "__weakref__")

    def __init__(self):
        self.calls, self.pending, self.ok, self.last = 0, [], True, None


_GUARDS: dict = {}      # id(weight) -> (weak reference to the weight, its guard): tensors compare elementwise, so they cannot
                        # key a WeakKeyDictionary; the entry is dropped when the weight dies


def _guard_of(w_last, create: bool = False):
    hit = _GUARDS.get(id(w_last))
    if hit is not None and hit[0]() is w_last:
        return hit[1]
    if not create:
        return None
    key = id(w_last)
    g = _Guard()
    _GUARDS[key] = (weakref.ref(w_last, lambda _r, key=key: _GUARDS.pop(key, None)), g)
    return g


def _poll(g: _Guard) -> None:
    if torch.cuda.is_current_stream_capturing():
        return                             # Event.query() is not allowed while a stream captures (it would invalidate the capture);
                                           # ``drain_guards()`` empties the lists before CapturedStep starts recording
    while g.pending and g.pending[0][0].query():
        _, host = g.pending.pop(0)
        g.last = float(host[0])
        if not (g.last <= GUARD_TOL):       # (also catches NaN)
            if g.ok:
                import warnings

                warnings.warn(f"radial knot table: interpolation error bound {g.last:.2e} exceeds {GUARD_TOL:.0e} "
                              f"({KNOTS} knots): this radial MLP is evaluated per edge from now on "
                              "(E3K_RADIAL_KNOTS raises the resolution)")
            g.ok = False


def drain_guards() -> None:
    """Read back every pending guard estimate (call after a device synchronisation, before a graph capture: no event may be
    queried while the stream records -- ``run/graph_step.CapturedStep`` does)."""
    for ref, g in list(_GUARDS.values()):
        if ref() is None:
            continue
        for ev, _ in g.pending:
            ev.synchronize()
        _poll(g)


def guard_ok(w_last) -> bool:
    g = _guard_of(w_last)
    if g is None:
        return True
    _poll(g)
    return g.ok


def guard_error(w_last):
    """Last error bound read back for this MLP (None before the first one arrived)."""
    g = _guard_of(w_last)
    if g is None:
        return None
    _poll(g)
    return g.last


def guard(w_last, table: torch.Tensor) -> None:
    """Call with the table just computed (on the stream that computed it)."""
    g = _guard_of(w_last, create=True)
    g.calls += 1
    _poll(g)
    if (g.calls - 1) % max(GUARD_EVERY, 1) != 0 or torch.cuda.is_current_stream_capturing() or table.shape[0] < 4:
        return
    with torch.no_grad():
        t = table.detach()
        d3 = t[3:] - 3.0 * t[2:-1] + 3.0 * t[1:-2] - t[:-3]
        est = (d3.abs().amax() * _C3 / t.abs().amax().clamp_min(1e-30)).reshape(1)
        host = torch.empty(1, dtype=torch.float32).pin_memory()
        host.copy_(est, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
    g.pending.append((ev, host))


def interp_fwd_raw(table: torch.Tensor, bins) -> torch.Tensor:
    bin32, t, ptr, perm = bins
    e, width = bin32.numel(), table.shape[1]
    w = torch.empty(e, width, device=table.device, dtype=torch.float32)
    with ops.timed_launch("rtable_fwd", (e, KNOTS, width)):
        L.check(L.load().e3k_rtable_interp_fwd(L.ptr(table), L.ptr(perm), L.ptr(bin32), L.ptr(t), e, KNOTS, width, L.ptr(w),
                                               L.stream_ptr()), "e3k_rtable_interp_fwd")
    return w


def interp_bwd_raw(g_w: torch.Tensor, bins) -> torch.Tensor:
    bin32, t, ptr, perm = bins
    width = g_w.shape[1]
    g_t = torch.empty(KNOTS + 1, width, device=g_w.device, dtype=torch.float32)
    work = torch.empty(L.load().e3k_rtable_bwd_workspace_floats(KNOTS, width), device=g_w.device, dtype=torch.float32)
    with ops.timed_launch("rtable_bwd", (bin32.numel(), KNOTS, width)):
        L.check(L.load().e3k_rtable_interp_bwd(L.ptr(g_w), L.ptr(ptr), L.ptr(perm), L.ptr(t), bin32.numel(), KNOTS, width,
                                               L.ptr(work), L.ptr(g_t), L.stream_ptr()), "e3k_rtable_interp_bwd")
    return g_t


class RadialTableFn(torch.autograd.Function):
    """w [E, W] = interpolation of the table T [KNOTS + 1, W] at the edges' radii; backward: g_T (the radii are data)."""

    @staticmethod
    def forward(ctx, table, src: RadialSource):
        ctx.src = src
        return interp_fwd_raw(L.f32c(table), src.bins())

    @staticmethod
    def backward(ctx, g_w):
        if torch.is_grad_enabled():
            raise RuntimeError("double backward through the radial table is not built: set E3K_RADIAL_TABLE=0")
        return interp_bwd_raw(L.f32c(g_w), ctx.src.bins()), None


def last_weight(fc):
    """The last-layer weight Parameter of a FullyConnectedNet: the key of its guard."""
    return list(fc.children())[-1].weight


def table_weights(fc, edge_radial) -> torch.Tensor:
    """``fc(edge_radial)`` through the knot table (call only when ``applicable(edge_radial, last_weight(fc))``)."""
    src = source_of(edge_radial)
    table = fc(src.knot_basis())            # the MLP on KNOTS + 1 rows: its forward AND backward shrink with it
    guard(last_weight(fc), table)
    return RadialTableFn.apply(table, src)
