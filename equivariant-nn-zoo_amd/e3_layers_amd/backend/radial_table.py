"""Per-edge radial weights through a knot table (``csrc/e3k_rtable.hip``, ``csrc/e3k_slope.hip``).

``weight = fc(edge_radial)`` (``e3_layers/nn/message_passing.py:74-79,93``) with ``edge_radial =
RadialBasisEncoding(edge_length)`` (``e3_layers/nn/embedding.py:210-219``) is a smooth function of one scalar per
edge.  Instead of pushing every edge through the MLP (the largest block of matrix work of a training step: forward,
dgrad and wgrad GEMMs of ``[E, 64] x [64, weight_numel]``), the MLP is evaluated on ``knots + 1`` equidistant radii and
every edge interpolates between the FOUR knots around it with cubic Lagrange weights (round 4; rounds 2-3: quadratic on
2 048 knots); the backward transposes the interpolation (an ordered sum per knot over edges sorted by knot, no atomics)
and then differentiates the MLP on the knots only.

Knot counts (``tools/knot_error.py``; float64 study in DESIGN.md section 4): cubic, 512 knots: 1.0e-7 relative -- the fp32
rounding of the table itself -- where the quadratic rule needed 2 048; the table the tensor-product kernels gather from
(``e3k_tp_fwd_table``: four rows per edge) is 3.9 MB per layer instead of 14 and fits an XCD's 4 MB L2.

Force training (``GradientOutput``: the radii require grad) needs ``dw/dr`` as well.  Differentiating the interpolation
weights amplifies the table's fp32 rounding by 1 / knot spacing (7e-6 .. 4e-5 relative at any knot count: beyond the
force tolerance), so the slope is a table of its own, ``D = dT/dr`` on the knots (``conv_native.RadialStackFn`` with
``slope`` -> ``e3k_radial_slope_fwd``: the hidden chain's forward-mode derivative per knot in float64, the last layer in fp32),
interpolated with the same weights (2e-7 at 512 knots).

Applies when ``edge_radial`` still carries the tag ``RadialBasisEncoding.forward`` puts on its output (so nothing was
concatenated to it: the diffusion configs mix bond types / residue offsets into the edge embedding and take the
per-edge path), the envelope is the polynomial cutoff (f is constant beyond ``r_max``) and the batch has several times
more edges than the table has knots.  ``E3K_RADIAL_TABLE=0`` disables it.
"""
from __future__ import annotations

import os
import weakref
from typing import Optional

import torch

from .tuning import knob as _knob

from . import lib as L
from . import ops

ENABLED = _knob("E3K_RADIAL_TABLE")
# Target knot counts over [0, r_max]; the spacing actually used is the power of two at or below r_max / target (``layout``):
# r / h, the offset inside a knot interval and the knot radii k h are then EXACT in fp32 -- with an arbitrary spacing their
# rounding (6e-8 relative, i.e. 3e-5 of an interval at knot 500) shows up as 1e-6 in the interpolated weights.
KNOTS = _knob("E3K_RADIAL_KNOTS")                  # r_max 4: 512 intervals of 2^-7 A; r_max 5: 640
KNOTS_SLOPE = _knob("E3K_RADIAL_KNOTS_SLOPE")       # ... when the radii require grad (value + slope tables)
MIN_EDGES_PER_KNOT = _knob("E3K_RADIAL_MIN_EDGES_PER_KNOT")      # below this the per-edge MLP is the cheaper one
# Refinement (round 6).  When a guard's bound passes the tolerance (the weights sharpen under the optimizer: the bound grows like
# scale^4) the first answer is a FINER table, not the per-edge MLP: both target counts double (the cubic's error drops 16 x; the
# bound is re-measured on the new table at once), up to KNOTS_MAX; only a bound that 2048 knots cannot hold switches that MLP's
# table off.  Measured on the headline step: 3.94 ms at 512 knots, 4.09 ms at 1024, 4.46 ms with ONE layer's MLP per edge
# (``tools/soak.sh``: 5 000 Adam steps at lr 1e-2 on four batches trip layer 1's per-column bound).  One resolution for every
# table in the process: the layers share the batch's bins and edge records.  A keyed table's guard (rows x keys against the bins
# kernel's 4 000-row histogram) does not refine, and a keyed source follows a refinement only as far as its stack still fits.
KNOTS_MAX = _knob("E3K_RADIAL_KNOTS_MAX")
_KNOTS_AT_START = (int(KNOTS), int(KNOTS_SLOPE))
REFINEMENTS = 0      # how many times the counts doubled in this process (CapturedStep compares it with its own: a graph recorded
                     # before a refinement holds the coarse tables and records itself again)
# Keyed tables (``KeyedRadialSource``: an edge embedding that is a function of the radius and a small categorical key -- config_diffusion's
# bond type).  Built and pinned to the oracle in round 5 (score 2.4e-6, parameter gradients 3.1e-6), and OFF by default: the only shipped
# model it serves has 32 channels, whose tensor-product plans have no in-kernel table form, so the weights are still materialised
# [E, W] per layer, its radial MLP is 32 wide (2.6 GFLOP per layer per edge pass: not what the step waits for), and 42 k edges spread
# over 4 x 733 knot rows are 14 edges per row -- the transposed interpolation's segments are nearly empty.  Measured, 128 molecules,
# graph-replayed step: 3.52 ms with the tables against 3.32 ms per edge (profiles/r05_bench_lines.json).  1 switches it on.
KEYED = _knob("E3K_RADIAL_TABLE_KEYED")


def layout(r_max: float, target: int):
    """(intervals K, spacing h): h = 2^-m <= r_max / target, K h >= r_max (the table's last rows sit at or beyond r_max, where the
    function is constant)."""
    import math

    m = math.ceil(math.log2(max(int(target), 4) / float(r_max)) - 1e-9)
    h = 2.0 ** (-m)
    return max(int(math.ceil(float(r_max) / h - 1e-9)), 4), h


class KnotBins:
    """What a batch's edges look like from the table's side (``e3k_rtable_bins``): ``bin`` int32 [E] (knot i: stencil rows
    i - 1 .. i + 2), ``coef`` [E, 4] (the four weights), the edges grouped by knot -- ``ptr`` [K + 2], ``perm`` [E] (ascending
    edge id inside a knot), ``seg`` [K + 2] (first <= 64-edge segment of every knot) -- and the knot count."""

    __slots__ = ("bin", "coef", "ptr", "perm", "seg", "knots", "spacing", "blocks", "_buf", "_rec", "__weakref__")

    def __init__(self, bin, coef, ptr, perm, seg, knots: int, spacing: float, buf=None):
        self.bin, self.coef, self.ptr, self.perm, self.seg, self.knots, self.spacing = bin, coef, ptr, perm, seg, int(knots), float(spacing)
        self._buf = buf
        self._rec = {}
        self.blocks = 1      # > 1: ``blocks`` tables of (knots + 1) / blocks rows stacked (keyed tables, ``build_bins(key=...)``)

    def records(self, topo, sh: torch.Tensor, walk: str) -> torch.Tensor:
        """The batch's EDGE RECORDS [E, 16] int32 (``e3k_edge_records``) for the walk over the destination CSR (``walk="dst"``: the
        forward, neighbour = source) or over the source CSR (``"src"``: the input gradient, neighbour = destination): per edge of
        the walk ONE 64-byte block {neighbour, knot, four interpolation weights, nine spherical harmonics, edge id} -- what the
        packed-table tensor-product kernels read instead of chasing perm -> edge -> {src, bin, coef, sh}.  Built once per batch and
        walk (on the current stream), shared by all the layers."""
        perm, nbr = (topo.dst_perm, topo.src) if walk == "dst" else (topo.src_perm, topo.dst)
        key = (walk, sh.data_ptr(), sh._version, perm.data_ptr())
        hit = self._rec.get(key)
        if hit is None:
            sh = L.f32c(sh.detach())
            e = perm.numel()
            if sh.dim() != 2 or sh.shape[0] != e or sh.shape[1] > 9:
                raise ValueError(f"edge records take [E, <= 9] spherical harmonics, got {tuple(sh.shape)} for {e} edges")
            rec = torch.empty(max(e, 1), 16, dtype=torch.int32, device=sh.device)
            L.check(L.load().e3k_edge_records(L.ptr(perm), L.ptr(nbr), L.ptr(self.bin), L.ptr(self.coef), L.ptr(sh), sh.shape[1], e,
                                              L.ptr(rec), L.stream_ptr()), "e3k_edge_records")
            hit = self._rec[key] = (rec, sh)      # (sh kept: its address is part of the key)
        return hit[0]

    def tensors(self):
        """The distinct allocations behind the views (what ``record_stream`` has to see)."""
        return (self._buf, self.coef) if self._buf is not None else (self.bin, self.coef, self.ptr, self.perm, self.seg)

    def __iter__(self):
        return iter(self.tensors())


def build_bins(r: torch.Tensor, r_max: float, target: int, key: Optional[torch.Tensor] = None, n_keys: int = 1) -> KnotBins:
    """``target``: the knot count asked for over [0, r_max] (``layout`` turns it into a power-of-two spacing).
    ``key`` [E] int64 in [0, n_keys): KEYED tables -- ``n_keys`` tables of ``knots + 1`` rows stacked into one, an edge's knot is
    ``key[e] (knots + 1) + i``; the returned bins describe the stacked table (``.knots`` = its last row, ``.blocks`` = n_keys)."""
    knots, h = layout(r_max, target)
    r = L.f32c(r.detach().reshape(-1))
    L.require_cuda(r)
    e, dev = r.numel(), r.device
    lib = L.load()
    n_keys = int(n_keys) if key is not None else 1
    last = n_keys * (knots + 1) - 1            # last row of the (stacked) table
    sizes = [e, e, last + 2, last + 2, int(lib.e3k_rtable_bins_workspace_ints(e, last))]
    buf = torch.empty(sum(sizes), dtype=torch.int32, device=dev)      # one allocation: [bin | perm | ptr | seg | workspace]
    bin32, perm, ptr, seg, work = torch.split(buf, sizes)
    coef = torch.empty(e, 4, dtype=torch.float32, device=dev)
    if key is None:
        L.check(lib.e3k_rtable_bins(L.ptr(r), e, 1.0 / h, knots, L.ptr(bin32), L.ptr(coef), L.ptr(ptr), L.ptr(seg), L.ptr(perm),
                                    L.ptr(work), L.stream_ptr()), "e3k_rtable_bins")
    else:
        key = key.detach().reshape(-1)
        if key.dtype != torch.int64 or not key.is_contiguous():
            key = key.long().contiguous()
        if key.numel() != e:
            raise ValueError(f"{key.numel()} keys for {e} radii")
        from . import graph as _graph

        capturing = torch.cuda.is_current_stream_capturing()
        flag = _graph._capture_flags.get(dev.index) if capturing else _graph.persistent_flag(dev)
        L.check(lib.e3k_rtable_bins_keyed(L.ptr(r), L.ptr(key), n_keys, e, 1.0 / h, knots, L.ptr(bin32), L.ptr(coef), L.ptr(ptr),
                                          L.ptr(seg), L.ptr(perm), L.ptr(work), L.ptr(flag), L.stream_ptr()), "e3k_rtable_bins_keyed")
        if flag is not None:      # (a key outside [0, n_keys) interpolates in another bond type's table: reported like a bad edge endpoint)
            _graph.report_persistent(dev)
    bins = KnotBins(bin32, coef, ptr, perm, seg, last, h, buf)
    bins.blocks = n_keys
    return bins


class RadialSource:
    """What ``RadialBasisEncoding`` knows about an edge embedding it produced: the module, the radii, and the version
    counter of the embedding tensor when it was tagged (an in-place op on the embedding bumps it: the tag then no longer
    describes the tensor's contents and the table is not used)."""

    __slots__ = ("module", "r", "version", "knots", "knots_slope", "_bins", "_knot_basis", "_stack", "__weakref__")

    def __init__(self, module, r: torch.Tensor, version: int = 0, prepared: Optional[dict] = None):
        """``prepared``: {target knot count: KnotBins} built ahead of the step for exactly these radii (``prepare_bins``)."""
        self.module, self.r, self.version = weakref.ref(module), r, int(version)
        self.knots, self.knots_slope = int(KNOTS), int(KNOTS_SLOPE)      # this forward's resolution: a refinement that happens while
                                                                         # it runs (``_refine``) applies from the NEXT source on
        self._bins = dict(prepared) if prepared else {}
        self._knot_basis = {}
        self._stack = {}      # id(MessagePassing) -> (its radial MLP's rows on the knots, mode): nn/message_passing.py:_stack_rows

    def bins(self, knots: Optional[int] = None) -> KnotBins:
        """Once per batch and (target) knot count, shared by the layers."""
        knots = self.knots if knots is None else int(knots)
        hit = self._bins.get(knots)
        if hit is None:
            hit = self._bins[knots] = build_bins(self.r, float(self.module().basis.r_max), knots)
        return hit

    def knot_basis(self, knots: Optional[int] = None):
        """RadialBasisEncoding on the knots (differentiable w.r.t. the Bessel frequencies) -- once per forward."""
        knots = self.knots if knots is None else int(knots)
        kb = self._knot_basis.get(knots)
        if kb is None or kb[1] != torch.is_grad_enabled():
            mod = self.module()
            b, c = mod.basis, mod.cutoff
            basis = ops.radial_basis(knot_radii(float(b.r_max), knots, self.r.device), b.bessel_weights, b.r_max, b.r_min, c.p,
                                     b.one_over_r, c.cutoff.kind)
            kb = self._knot_basis[knots] = (basis, torch.is_grad_enabled())
        return kb[0]


def prepare_bins(r: torch.Tensor, r_max: float, target: int) -> KnotBins:
    """Knot bins of the radii ``r`` built NOW and left on the tensor (``SequentialGraphNetwork.prepare_data``: the batch's radii are
    known before the step that reads them); ``prepared_bins(r)`` hands them out while ``r`` has not been written to since."""
    bins = build_bins(r, r_max, target)
    r._e3k_bins = (r._version, float(r_max), {int(target): bins})
    return bins


def prepared_bins(r: torch.Tensor) -> Optional[dict]:
    hit = getattr(r, "_e3k_bins", None)
    return hit[2] if (hit is not None and hit[0] == r._version) else None


_KNOT_CACHE = {}


def knot_radii(r_max: float, target: int, device) -> torch.Tensor:
    """The radii of the table's rows for a target knot count: k h, exact in fp32 (h a power of two)."""
    key = (r_max, target, str(device))
    k = _KNOT_CACHE.get(key)
    if k is None:
        knots, h = layout(r_max, target)
        k = torch.arange(knots + 1, dtype=torch.float64) * h
        k[0] = 1e-6 * h                        # sin(w r) / r at r = 0: evaluate next to it (f is smooth there)
        k = _KNOT_CACHE[key] = k.float().to(device)
    return k


class KeyedRadialSource:
    """An edge embedding that is a ROW-WISE function of (radius, small categorical key): ``Concat(one_hot(bond type), RadialBasisEncoding(
    edge_length)) -> Linear`` in ``e3_layers/configs/config_diffusion.py:73-82``.  It is served by ``n_keys`` knot tables stacked into one:
    the radial MLP of a layer runs on ``n_keys (knots + 1)`` rows -- the embedding of every (key, knot radius) pair, ``rows_fn`` applied
    to the base source's knot basis -- and an edge interpolates inside the block of its key.  Same interface as ``RadialSource``."""

    __slots__ = ("base", "key", "n_keys", "rows_fn", "version", "_bins", "_knot_basis", "_stack", "__weakref__")

    def __init__(self, base: RadialSource, key: torch.Tensor, n_keys: int, rows_fn, version: int = 0):
        self.base, self.key, self.n_keys, self.rows_fn, self.version = base, key, int(n_keys), rows_fn, int(version)
        self._bins, self._knot_basis, self._stack = {}, {}, {}

    @property
    def r(self):
        return self.base.r

    @property
    def blocks(self) -> int:
        return self.n_keys

    def module(self):
        return self.base.module()

    def _fit(self, knots: int, floor: int) -> int:
        """A refined resolution (``_refine``) only as far as the stacked table still fits the bins kernel's 4 000-row histogram."""
        r_max = float(self.module().basis.r_max)
        while knots > floor and (layout(r_max, knots)[0] + 1) * self.n_keys > 4000:
            knots //= 2
        return knots

    @property
    def knots(self) -> int:
        return self._fit(self.base.knots, _KNOTS_AT_START[0])

    @property
    def knots_slope(self) -> int:
        return self._fit(self.base.knots_slope, _KNOTS_AT_START[1])

    def bins(self, knots: Optional[int] = None) -> KnotBins:
        knots = self.knots if knots is None else int(knots)
        hit = self._bins.get(knots)
        if hit is None:
            hit = self._bins[knots] = build_bins(self.base.r, float(self.module().basis.r_max), knots, self.key, self.n_keys)
        return hit

    def knot_basis(self, knots: Optional[int] = None):
        """[n_keys (knots + 1), width]: the embedding of every (key, knot) pair, key-major; differentiable w.r.t. the parameters of
        ``rows_fn`` (the Concat's Linear) and of the basis (the Bessel frequencies)."""
        knots = self.knots if knots is None else int(knots)
        kb = self._knot_basis.get(knots)
        if kb is None or kb[1] != torch.is_grad_enabled():
            base = self.base.knot_basis(knots)                               # [rows, n_basis]
            rows = base.shape[0]
            keys = torch.arange(self.n_keys, device=base.device).repeat_interleave(rows)
            rows_all = self.rows_fn(keys, base.repeat(self.n_keys, 1))
            rows_all._e3k_blocks = self.n_keys
            kb = self._knot_basis[knots] = (rows_all, torch.is_grad_enabled())
        return kb[0]


def source_of(edge_radial) -> Optional[RadialSource]:
    return getattr(edge_radial, "_e3k_radial_src", None)


def knots_for(edge_radial, w_last=None, allow_grad: bool = False) -> int:
    """The (target) knot count of the table that serves this edge embedding, or 0 when the table does not apply.  ``w_last``: the
    last-layer weight of the radial MLP that would run on the table -- its a-posteriori error guard (``guard`` below) can
    veto the table for that MLP.  ``allow_grad``: the caller can differentiate w.r.t. the radii through the slope table
    (the force block, ``backend/conv_force.py``); everyone else takes the per-edge path when the radii require grad."""
    if not ENABLED or not edge_radial.is_cuda:
        return 0
    src = source_of(edge_radial)
    if src is None or src.module() is None:
        return 0
    if isinstance(src, KeyedRadialSource) and not KEYED:
        return 0
    grad = bool(src.r.requires_grad)
    if grad and not allow_grad:
        return 0
    if edge_radial._version != src.version:
        return 0                           # modified in place since RadialBasisEncoding produced it
    if w_last is not None and not guard_ok(w_last, slope=grad):
        return 0
    knots = src.knots_slope if grad else src.knots
    rows = (layout(float(src.module().basis.r_max), knots)[0] + 1) * getattr(src, "blocks", 1)
    if rows > 4000:                        # (the stacked table of a keyed source: the bins kernel's LDS histogram)
        return 0
    return knots if edge_radial.shape[0] >= MIN_EDGES_PER_KNOT * rows else 0


def applicable(edge_radial, w_last=None) -> bool:
    return knots_for(edge_radial, w_last) > 0


# ---- a-posteriori guard of the interpolation error ---------------------------------------------------------------------
# Cubic Lagrange interpolation on knots h apart, evaluated in the middle interval of its four knots, is off by at most
# 3/128 h^4 max|f(4)|; on the table itself h^4 f(4) is the fourth finite difference, so for weight column c
#     err_c <= 3/128 max_i |T[i+4,c] - 4 T[i+3,c] + 6 T[i+2,c] - 4 T[i+1,c] + T[i,c]|
# -- read off the table rows the forward has just computed (``e3k_rtable_guard``: one launch for all the tables of a radial
# stack).  Two ratios: TABLE-WIDE, max_c err_c / max|T| <= GUARD_TOL (1e-6: what the forward's parity feels -- a column's error
# against the scale of the weights it is summed with; 1.1-1.8e-7 for the shipped models at random init on 512 knots, mostly the
# fp32 rounding of the rows, whose fourth difference carries 16 eps), and PER COLUMN (round 5), err_c relative to the column's
# own largest entry, floored at GUARD_COL_FLOOR (2^-7) of the table's largest, <= GUARD_TOL_COL (1e-5 since round 6; 2e-5 in round 5) -- until round 5 only the
# table-wide ratio existed and a column a thousand times smaller than the largest could be off by 1e-3 relative and pass.  The
# per-column ratio has its own, wider tolerance because random combinations of the hidden units that cancel their smooth part
# have 20-40 x the typical relative curvature: 1-4e-6 at random init over 2-3 k columns (``tools/guard_probe.py``,
# ``profiles/r05_guard_probe.txt``) -- a tolerance of 1e-6 there would veto every table of every shipped model.  The kernel
# reports est = max(table-wide, per-column * GUARD_TOL / GUARD_TOL_COL), compared with GUARD_TOL.  Both grow like
# (frequency x weight scale)^4, so a 32-function basis, grown Bessel frequencies or large trained weights can push them towards
# the 1e-5 parity budget.
#
# Who looks at it:
#   eager steps     the estimate of the first build and of every GUARD_EVERY-th after travels to pinned memory without a sync and
#                   is read on a later call;
#   replayed steps  (``run/graph_step.CapturedStep``: a HIP graph never re-enters this module) the kernel is PART of the captured
#                   step: every replay folds its estimate into a persistent per-MLP device maximum, which the replay loop sends
#                   to the host every GUARD_EVERY-th replay (``poll_replay``) -- the weights move under Adam while the graph
#                   replays, and the bound grows like scale^4.
# Above GUARD_TOL the table is switched off for that MLP (per-edge evaluation from then on, a warning; a
# CapturedStep whose graph evaluates that guard records itself again, without the table).  The slope table of force training has a guard of its own (same rule on
# D, relative to its own columns: what is interpolated there is the slope).
GUARD_TOL = _knob("E3K_RADIAL_TABLE_TOL")
GUARD_EVERY = _knob("E3K_RADIAL_TABLE_CHECK_EVERY")
GUARD_COL_FLOOR = _knob("E3K_RADIAL_TABLE_COL_FLOOR")
GUARD_TOL_COL = _knob("E3K_RADIAL_TABLE_TOL_COL")


class _Guard:
    __slots__ = ("calls", "pending", "ok", "last", "dev", "scratch", "what", "keyed", "__weakref__")

    def __init__(self, what: str):
        self.calls, self.pending, self.ok, self.last, self.what, self.keyed = 0, [], True, None, what, False
        self.dev = self.scratch = None      # device state [4] (running max, last estimate, ticket, -) and the kernel's column scratch

    def device_state(self, device, width: int):
        """Allocated OUTSIDE any capture (persistent across replays); None while capturing if it does not exist yet."""
        if self.dev is None or self.dev.device != device or self.scratch.numel() < 16 * width:
            if torch.cuda.is_current_stream_capturing():
                return None
            self.dev = torch.zeros(4, dtype=torch.float32, device=device)
            self.scratch = torch.empty(16 * width, dtype=torch.float32, device=device)      # (e3k.h: 8 row ranges x 2 x W)
        return self.dev


_GUARDS: dict = {}      # (id(weight), slope) -> (weak reference to the weight, its guard): tensors compare elementwise, so they
                        # cannot key a WeakKeyDictionary; the entry is dropped when the weight dies


def _guard_of(w_last, slope: bool = False, create: bool = False):
    key = (id(w_last), slope if isinstance(slope, tuple) else bool(slope))      # (slope, block) for the blocks > 0 of a stacked table
    hit = _GUARDS.get(key)
    if hit is not None and hit[0]() is w_last:
        return hit[1]
    if not create:
        return None
    g = _Guard("slope" if (slope[0] if isinstance(slope, tuple) else slope) else "value")
    _GUARDS[key] = (weakref.ref(w_last, lambda _r, key=key: _GUARDS.pop(key, None)), g)
    return g


def _poll(g: _Guard, what: str = None) -> None:
    if torch.cuda.is_current_stream_capturing():
        return                             # Event.query() is not allowed while a stream captures (it would invalidate the capture);
                                           # ``drain_guards()`` empties the lists before CapturedStep starts recording
    while g.pending and g.pending[0][0].query():
        _, host = g.pending.pop(0)
        g.last = float(host[0])
        if not (g.last <= GUARD_TOL):       # (also catches NaN)
            if g.ok:
                import warnings

                bound = g.last
                if _refine(g):
                    warnings.warn(f"radial knot table ({g.what}): interpolation error bound {bound:.2e} exceeds {GUARD_TOL:.0e}: the "
                                  f"tables are rebuilt on {KNOTS} (value) / {KNOTS_SLOPE} (value + slope) knots from the next step on")
                    return                 # (every guard's read-backs described the coarse tables: dropped by _refine)
                warnings.warn(f"radial knot table ({g.what}): interpolation error bound {g.last:.2e} exceeds {GUARD_TOL:.0e}: this "
                              "radial MLP is evaluated per edge from now on (E3K_RADIAL_KNOTS_MAX raises the finest resolution the "
                              "tables may take)")
            g.ok = False


def _refine(g: _Guard) -> bool:
    """Doubles both target knot counts if the table behind ``g`` can still get finer (see KNOTS_MAX above); every guard then starts
    over: what it has measured, on the device and on its way home, describes the coarse tables."""
    global KNOTS, KNOTS_SLOPE, REFINEMENTS
    if g.last != g.last:                    # NaN: not a matter of resolution
        return False
    if g.keyed:
        return False                        # a stacked (keyed) table: its rows x keys are bounded by the bins kernel's histogram
    if 2 * int(KNOTS_SLOPE if g.what == "slope" else KNOTS) > int(KNOTS_MAX):
        return False
    KNOTS = min(2 * int(KNOTS), max(int(KNOTS_MAX), int(KNOTS)))
    KNOTS_SLOPE = min(2 * int(KNOTS_SLOPE), max(int(KNOTS_MAX), int(KNOTS_SLOPE)))
    REFINEMENTS += 1
    for _, (ref, other) in list(_GUARDS.items()):
        if ref() is None:
            continue
        other.pending.clear()               # (_HOST_BUSY keeps the events: a slot is reused only after its copy has landed)
        other.calls = 0                     # the next eager build measures the fine table at once
        if other.ok:
            other.last = None
        if other.dev is not None:
            other.dev.zero_()
    return True


def drain_guards() -> None:
    """Read back every pending guard estimate (call after a device synchronisation, before a graph capture: no event may be
    queried while the stream records -- ``run/graph_step.CapturedStep`` does)."""
    for _, (ref, g) in list(_GUARDS.items()):
        if ref() is None:
            continue
        for ev, _ in g.pending:
            ev.synchronize()
        _poll(g)


def guard_ok(w_last, slope: bool = False) -> bool:
    """The value table's guard (all its blocks, for a stacked keyed table), and -- ``slope`` -- the slope table's too."""
    ok = True
    wid = id(w_last)
    for (kid, kind), (ref, g) in list(_GUARDS.items()):
        if kid != wid or ref() is not w_last:
            continue
        is_slope = kind[0] if isinstance(kind, tuple) else kind
        if is_slope and not slope:
            continue
        _poll(g)
        ok = ok and g.ok
    return ok


def guard_error(w_last, slope: bool = False):
    """Last error bound read back for this MLP's value (or slope) table (None before the first one arrived)."""
    g = _guard_of(w_last, slope)
    if g is None:
        return None
    _poll(g)
    return g.last


_HOST_RING = None      # pinned float slots for the read-backs, allocated once (a pinned allocation synchronises the device)
_HOST_NEXT = 0
_HOST_SLOTS = 512
_HOST_BUSY = {}        # slot index -> event of the read-back that last took it (a slot is not handed out again while
                       # that read-back has not landed and been looked at -- 512 outstanding read-backs mean nobody polls)


def _host_slot() -> torch.Tensor:
    global _HOST_RING, _HOST_NEXT
    if _HOST_RING is None:
        _HOST_RING = torch.zeros(_HOST_SLOTS, dtype=torch.float32).pin_memory()
    for _ in range(_HOST_SLOTS):
        i = _HOST_NEXT
        _HOST_NEXT = (_HOST_NEXT + 1) % _HOST_SLOTS
        ev = _HOST_BUSY.get(i)
        if ev is None or ev.query():
            _HOST_BUSY.pop(i, None)
            _HOST_SLOT_TAKEN[0] = i
            return _HOST_RING[i:i + 1]
    raise RuntimeError("radial table guard: 512 read-backs outstanding -- nothing polls the guards (drain_guards() after a sync)")


_HOST_SLOT_TAKEN = [0]


def _send(g: _Guard, index: int, reset: bool) -> None:
    """state[index] -> pinned memory behind everything enqueued on the current stream; ``reset``: the running maximum starts over"""
    host = _host_slot()
    host.copy_(g.dev[index:index + 1], non_blocking=True)
    if reset:
        g.dev[0:1].zero_()
    ev = torch.cuda.Event()
    ev.record()
    _HOST_BUSY[_HOST_SLOT_TAKEN[0]] = ev
    g.pending.append((ev, host))


def guard_many(items) -> None:
    """``items``: [(last-layer weight (the guard's key), table [rows, W], slope: bool)] -- tables of one row count, just computed on
    the current stream.  Eager: the first call and every GUARD_EVERY-th launch the estimate kernel and send its result home;
    while a stream captures: the kernel is recorded with EVERY build (each replay then updates the persistent maxima)."""
    capturing = torch.cuda.is_current_stream_capturing()
    todo = []
    flat = []
    for it in items:      # a stacked (keyed) table is guarded block by block: differences must not straddle two keys' tables
        w_last, table, slope = it[:3]
        blocks = int(it[3]) if len(it) > 3 else 1
        if blocks > 1 and table.shape[0] % blocks == 0:
            per = table.shape[0] // blocks
            flat += [(w_last, table[b * per:(b + 1) * per], slope, b, True) for b in range(blocks)]
        else:
            flat.append((w_last, table, slope, 0, False))
    for w_last, table, slope, block, keyed in flat:
        g = _guard_of(w_last, (slope, block) if block else slope, create=True)
        g.keyed = keyed
        g.calls += 1
        _poll(g)
        if table.shape[0] < 5 or not table.is_cuda:
            continue
        if not capturing and (g.calls - 1) % max(GUARD_EVERY, 1) != 0:
            continue
        if g.device_state(table.device, table.shape[1]) is None:
            continue                       # first seen inside a capture: CapturedStep's eager warm-up creates the states
        todo.append((g, table.detach()))
        if capturing and CAPTURE_LOG is not None and g not in CAPTURE_LOG:
            CAPTURE_LOG.append(g)          # (the step being recorded evaluates this guard with every replay)
    if not todo:
        return
    import ctypes as C

    lib = L.load()
    by_rows = {}
    for g, t in todo:                      # (one stack: one row count; value and slope tables of force training share it too)
        by_rows.setdefault(t.shape[0], []).append((g, t))
    for rows, group in by_rows.items():
        for base in range(0, len(group), 16):
            part = group[base:base + 16]
            n = len(part)
            tabs = (C.c_void_p * n)(*[t.data_ptr() for _, t in part])
            states = (C.c_void_p * n)(*[g.dev.data_ptr() for g, _ in part])
            scr = (C.c_void_p * n)(*[g.scratch.data_ptr() for g, _ in part])
            widths = (C.c_int32 * n)(*[t.shape[1] for _, t in part])
            # (packed = 1 always: the bound then also covers the fp16 rounding of the packed in-kernel form -- a few 1e-8 -- whether or not
            #  this table is read packed: one rule for every path, independent of which kernels a batch's size selects)
            L.check(lib.e3k_rtable_guard(tabs, states, scr, widths, n, rows, float(GUARD_COL_FLOOR), float(GUARD_TOL) / float(GUARD_TOL_COL),
                                         1, L.stream_ptr()), "e3k_rtable_guard")
    if not capturing:
        for g, _ in todo:
            _send(g, 1, reset=False)


def guard(w_last, table: torch.Tensor, slope: bool = False, blocks: int = 1) -> None:
    """Call with the table just computed (on the stream that computed it); ``blocks`` > 1: a stacked table of that many keys."""
    guard_many([(w_last, table, slope, blocks)])


CAPTURE_LOG = None      # while a CapturedStep records: the guards whose kernel went into the graph


def poll_replay(guards, count: int, every: int = None) -> bool:
    """For the loop that replays a captured step (``CapturedStep.__call__``, after replay number ``count`` (1-based), on its stream):
    looks at the estimates of ``guards`` (the ones recorded into that graph) that have arrived, and on the first replay and every
    ``every``-th (default GUARD_EVERY) sends each one's running maximum home and restarts it.  True when one of them has switched
    its table off (the caller records the step again)."""
    every = max(int(GUARD_EVERY if every is None else every), 1)
    vetoed = False
    for g in guards:
        _poll(g)
        if not g.ok:
            vetoed = True
        elif (count - 1) % every == 0:
            _send(g, 0, reset=True)
    return vetoed





def interp_fwd_raw(table: torch.Tensor, bins: KnotBins) -> torch.Tensor:
    e, width = bins.bin.numel(), table.shape[1]
    w = torch.empty(e, width, device=table.device, dtype=torch.float32)
    with ops.timed_launch("rtable_fwd", (e, bins.knots, width)):
        L.check(L.load().e3k_rtable_interp_fwd(L.ptr(table), L.ptr(bins.perm), L.ptr(bins.bin), L.ptr(bins.coef), e, bins.knots, width,
                                               L.ptr(w), L.stream_ptr()), "e3k_rtable_interp_fwd")
    return w


def pack_raw(table: torch.Tensor, knots: int) -> torch.Tensor:
    """P [knots + 1, W, 3] int32: the table packed for the tensor-product kernels (``e3k_rtable_pack``: 12-byte Taylor records)."""
    table = L.f32c(table)
    L.require_cuda(table)
    packed = torch.empty(table.shape[0], table.shape[1], 3, device=table.device, dtype=torch.int32)
    L.check(L.load().e3k_rtable_pack(L.ptr(table), int(knots), table.shape[1], L.ptr(packed), L.stream_ptr()), "e3k_rtable_pack")
    return packed


def interp_packed_raw(packed: torch.Tensor, bins: KnotBins) -> torch.Tensor:
    """w [E, W] from the packed table, with the arithmetic of the packed in-kernel form (bit-identical to it)."""
    e, width = bins.bin.numel(), packed.shape[1]
    w = torch.empty(e, width, device=packed.device, dtype=torch.float32)
    L.check(L.load().e3k_rtable_interp_packed(L.ptr(packed), L.ptr(bins.perm), L.ptr(bins.bin), L.ptr(bins.coef), e, bins.knots, width,
                                              L.ptr(w), L.stream_ptr()), "e3k_rtable_interp_packed")
    return w


def interp_bwd_raw(g_w: torch.Tensor, bins: KnotBins, scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """g_T = (interpolation)^T g_w; ``scale`` [E]: every edge's weights times scale[e] (the slope table's gradient in the double
    backward of force training)."""
    e, width = bins.bin.numel(), g_w.shape[1]
    lib = L.load()
    g_t = torch.empty(bins.knots + 1, width, device=g_w.device, dtype=torch.float32)
    work = torch.empty(lib.e3k_rtable_bwd_workspace_floats(e, bins.knots, width), device=g_w.device, dtype=torch.float32)
    with ops.timed_launch("rtable_bwd", (e, bins.knots, width)):
        L.check(lib.e3k_rtable_interp_bwd(L.ptr(g_w), L.ptr(bins.coef), L.ptr(scale), L.ptr(bins.ptr), L.ptr(bins.seg), L.ptr(bins.perm),
                                          e, bins.knots, width, L.ptr(work), L.ptr(g_t), 0, L.stream_ptr()), "e3k_rtable_interp_bwd")
    return g_t


class RadialTableFn(torch.autograd.Function):
    """w [E, W] = interpolation of the table T [knots + 1, W] at the edges' radii; backward: g_T (the radii are data)."""

    @staticmethod
    def forward(ctx, table, bins: KnotBins):
        ctx.bins = bins
        return interp_fwd_raw(L.f32c(table), bins)

    @staticmethod
    def backward(ctx, g_w):
        if torch.is_grad_enabled():
            raise RuntimeError("double backward through the materialised radial table is not built (force training runs the "
                               "force block, backend/conv_force.py, or the per-edge path): set E3K_RADIAL_TABLE=0")
        return interp_bwd_raw(L.f32c(g_w), ctx.bins), None


def last_weight(fc):
    """The last-layer weight Parameter of a FullyConnectedNet: the key of its guard."""
    return list(fc.children())[-1].weight


def table_weights(fc, edge_radial) -> torch.Tensor:
    """``fc(edge_radial)`` through the knot table (call only when ``applicable(edge_radial, last_weight(fc))``)."""
    src = source_of(edge_radial)
    table = fc(src.knot_basis())            # the MLP on KNOTS + 1 rows: its forward AND backward shrink with it
    guard(last_weight(fc), table, blocks=getattr(src, "blocks", 1))
    return RadialTableFn.apply(table, src.bins())
