"""Every switch and threshold of the Python side, in ONE documented table, read from the environment in ONE place.

VERDICT r3 counted 46 ``E3K_*`` environment reads scattered over eight modules and the C library.  Now:

* the C library reads NO environment variable in the product build (``csrc/e3k_common.h``: compile-time constants; ``make dbg``
  builds ``libe3k_dbg.so`` with the experiment knobs and the timing-only ablation mask, loaded only through ``E3K_LIB``);
* the Python modules ask ``knob(name)`` -- this module is the only one that touches ``os.environ``; ``table()`` lists every knob
  with its value, default and meaning (``python -m e3_layers_amd.backend.tuning``).  Modules keep their values as module-level
  constants (tests monkeypatch those), so the table is what the PROCESS started with.

Kinds: "path" = which of several equivalent code paths runs (results agree within rounding: each pair is pinned by a test, and
``tests/test_gpu_model.py::test_path_selection_thresholds_to_both_sides_meet_the_oracle`` sweeps the thresholds to both sides
against the float64 oracle); "threshold" = a size at which the default path changes; "accuracy" = changes numbers within a
documented bound; "debug".
"""
from __future__ import annotations

import os
from typing import Dict, Tuple

# name -> (default, kind, meaning)
KNOBS: Dict[str, Tuple[object, str, str]] = {
    # ---- which path a convolution layer takes (nn/message_passing.py) ----
    "E3K_CONV_BLOCK": (1, "path", "a layer as one autograd node (backend/conv_block.py); 0: one autograd.Function per kernel (the definition)"),
    "E3K_LAYER_NATIVE": (1, "path", "the fused layer's launch sequence issued by csrc/e3k_layer.hip; 0: the same sequence from Python"),
    "E3K_FORCE_BLOCK": (1, "path", "force training: a layer as three autograd nodes on the value + slope tables (backend/conv_force.py); 0: composed per-edge path"),
    "E3K_FORCE_MATERIALIZE": (1, "path", "force block: per-edge weights / slopes interpolated once per layer and streamed; 0: every kernel gathers the table rows"),
    "E3K_GEMM_CHAIN": (1, "path", "a Linear's terms that add onto one block (an irrep feeding two outputs: the second was an accumulating launch of its own) run as ONE K-chained GEMM problem; 0: one round of launches per term"),
    "E3K_FORCE_EDGE_ATOMICS": (0, "path", "force block: 1 = g_sh / g_r accumulated with float atomics across a plan's groups (rounds 4-5: forces not bit-reproducible); 0 = per-item partials combined in a fixed order"),
    "E3K_BLOCK_ADDEND": (1, "path", "layers with un-keyed node attributes run as fused blocks with the self-connection handed in as addend"),
    "E3K_ADDEND_INPLACE": (1, "path", "the addend tensor itself is the block's pre-gate buffer (no copy)"),
    "E3K_ADDEND_FORK": (1, "path", "addend blocks fork their radial branch under the same edge-count rule as keyed blocks"),
    "E3K_BLOCK_LOOK_AHEAD": (1, "path", "a forked layer issues the next layer's radial branch behind its own tensor product"),
    "E3K_CF_CHAIN": (1, "path", "consecutive MessagePassing layers hand their features over channel-fastest"),
    "E3K_CF_CHAIN_NORM": (1, "path", "... also through LayerNormalization"),
    "E3K_FUSED_MLP": (1, "path", "hidden chain of the radial MLP in one launch (csrc/e3k_mlp.hip); 0: one GEMM + activation per layer"),
    "E3K_RADIAL_STACK": (1, "path", "the radial MLPs of all layers on one edge embedding evaluated as one batch"),
    "E3K_KW_STACK": (1, "path", "the per-key self-connection weights of all layers formed as one batch"),
    "E3K_TP_TABLE": (1, "path", "table layers interpolate their path weights inside the tensor-product kernels (no w[E, W]); 0: interpolation pass"),
    "E3K_TP_TABLE_PACKED": (1, "accuracy", "... from the table packed into 12-byte Taylor records (two small coefficients in fp16: bounded by the guard); 0: four fp32 rows"),
    "E3K_TP_BWD_FUSED": (1, "path", "the input-gradient walk of the tensor product also writes what shares its per-edge sums: the weight gradient (packed-table layers, one-stream layers with streamed weights), force training's edge gradients and dual weight gradients (csrc/e3k_tp.hip MODE 5-8); 0: one kernel per gradient"),
    # ---- streams ----
    "E3K_FWD_FORK": (1, "path", "0: one stream; 1: radial / self-connection / weight-gradient branches on side streams above the edge thresholds; 2: also inside a graph capture"),
    "E3K_FWD_FORK_SC": (1, "path", "composed path: the self-connection on a third stream"),
    "E3K_WGRAD_SIDE": (1, "path", "sunk weight gradients of forked layers run on a side stream"),
    # ---- thresholds ----
    "E3K_FORK_MIN_EDGES": (25000, "threshold", "per-edge radial layers fork from this many edges (in units of a 1920-weight layer)"),
    "E3K_FORK_MIN_EDGES_TABLE": (60000, "threshold", "table layers fork from this many edges"),
    "E3K_STACK_MAX_EDGES": (50000, "threshold", "forked layers use the radial stack up to this many edges (one-stream layers: always)"),
    "E3K_KW_STACK_MAX_EDGES": (10 ** 9, "threshold", "keyed-weight stack up to this many edges (measured: no upper limit pays)"),
    "E3K_WGRAD_SIDE_MIN_ROWS": (2048, "threshold", "composed path: weight gradients move to the side stream from this many rows"),
    "E3K_KEY_MAX": (256, "threshold", "keyed self-connection for at most this many distinct attribute rows"),
    # ---- radial knot table ----
    "E3K_RADIAL_TABLE": (1, "accuracy", "radial MLP on a knot table + cubic interpolation per edge (bound by the guard below); 0: per edge"),
    "E3K_RADIAL_KNOTS": (512, "accuracy", "target knot count over [0, r_max] (the spacing is the power of two at or below r_max / knots)"),
    "E3K_RADIAL_KNOTS_SLOPE": (512, "accuracy", "... of the value + slope tables of force training"),
    "E3K_RADIAL_KNOTS_MAX": (2048, "accuracy", "a guard whose bound passes the tolerance doubles both knot counts up to this; beyond it that MLP's table is switched off (= E3K_RADIAL_KNOTS: never refine)"),
    "E3K_RADIAL_TABLE_KEYED": (0, "path", "one knot table per value of a small categorical edge key beside the radius (config_diffusion's bond type); measured slower than the per-edge MLP for the 32-channel net it serves: off"),
    "E3K_RADIAL_MIN_EDGES_PER_KNOT": (4.0, "threshold", "the table applies from this many edges per table row"),
    "E3K_RADIAL_TABLE_TOL": (1e-6, "accuracy", "a-posteriori interpolation-error bound above which an MLP's table is switched off"),
    "E3K_RADIAL_TABLE_CHECK_EVERY": (64, "debug", "eager: the guard is evaluated every this-many table builds; replayed graphs: its running maximum is read every this-many replays"),
    "E3K_RADIAL_TABLE_TOL_COL": (1e-5, "accuracy", "... and the bound of a weight column's error relative to the column's own scale "
                                 "(round 6: 2e-5 -> 1e-5, the forward bound it protects; measured 1.1-2.5e-6 at random init, <= 4.1e-6 after "
                                 "200 Adam steps at lr 1e-2: profiles/r06_parity_measured.jsonl)"),
    "E3K_RADIAL_TABLE_COL_FLOOR": (2.0 ** -7, "accuracy", "the guard's bound is relative to each weight column's own scale, floored at this fraction of the table's largest entry"),
    # ---- debug ----
    "E3K_HOST_TIMING": (0, "debug", "host seconds inside the layer functions (tools/host_split.py)"),
    "E3K_LIB": ("", "debug", "path of the shared library to load instead of csrc/libe3k.so (make dbg: libe3k_dbg.so)"),
}

_SEEN: Dict[str, object] = {}


def knob(name: str):
    """The value of a knob: the environment's when set, else the default -- converted to the default's type."""
    default, _, _ = KNOBS[name]
    raw = os.environ.get(name)
    if raw is None or raw == "":
        value = default
    elif isinstance(default, float):
        value = float(raw)
    elif isinstance(default, int):
        value = int(raw)
    else:
        value = raw
    _SEEN[name] = value
    return value


def table() -> str:
    rows = []
    for name, (default, kind, doc) in KNOBS.items():
        value = _SEEN.get(name, knob(name))
        mark = "*" if value != default else " "
        rows.append(f"{mark} {name:32s} {str(value):>12s}  (default {default}; {kind})  {doc}")
    return "\n".join(rows)


if __name__ == "__main__":
    print(table())
