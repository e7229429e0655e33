"""The C ABI: libe3k.so loads (no GPU needed), exports every symbol include/e3k.h declares, and the
ctypes mirror of its structs has the layout the C compiler gives them.  No compute calls."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "e3k.h")


def _declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(e3k_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from e3_layers_amd.backend import lib as L

    if not os.path.exists(L.LIB_PATH):
        import __graft_entry__

        __graft_entry__.build()
    handle = L.load()
    names = _declared_functions()
    assert len(names) >= 25
    for name in names:
        assert hasattr(handle, name), f"{name} declared in include/e3k.h but not exported by libe3k.so"
        assert name in L.SIGNATURES, f"{name} has no ctypes signature in backend/lib.py"
    assert handle.e3k_strerror(-3).decode().startswith("degree")
    assert handle.e3k_version() >= 100


def test_struct_layouts_match_the_c_compiler(tmp_path):
    from e3_layers_amd.backend import lib as L

    src = tmp_path / "sizes.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "e3k.h"\n'
        "int main(void){\n"
        'printf("%zu %zu %zu %zu\\n", sizeof(e3k_gemm_problem), sizeof(e3k_tp_group), sizeof(e3k_block), sizeof(e3k_gate_seg));\n'
        'printf("%zu %zu %zu %zu\\n", offsetof(e3k_gemm_problem, M1), offsetof(e3k_gemm_problem, a_r1), offsetof(e3k_gemm_problem, alpha), offsetof(e3k_tp_group, coeff));\n'
        'printf("%zu %zu\\n", sizeof(e3k_kw_instr), offsetof(e3k_kw_instr, u));\n'
        'printf("%zu %zu %zu %zu %zu\\n", sizeof(e3k_gemm_segment), sizeof(e3k_layer_desc), sizeof(e3k_layer_radial), sizeof(e3k_layer_fwd_args), sizeof(e3k_layer_bwd_args));\n'
        'printf("%zu %zu %zu %zu %zu %zu\\n", offsetof(e3k_gemm_segment, M1), offsetof(e3k_layer_desc, gate), offsetof(e3k_layer_desc, alphas), offsetof(e3k_layer_desc, tp_bwd_x_overwrites), offsetof(e3k_layer_fwd_args, rad), offsetof(e3k_layer_bwd_args, gb_hidden));\n'
        'printf("%zu %zu %zu\\n", offsetof(e3k_layer_fwd_args, x_cf), offsetof(e3k_layer_bwd_args, kw_ws), offsetof(e3k_layer_radial, z));\n'
        'printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(e3k_radial_stack_item), sizeof(e3k_mlp_net), sizeof(e3k_kw_multi_item), sizeof(e3k_kw_stack_item), offsetof(e3k_radial_stack_item, g_rows), offsetof(e3k_layer_bwd_args, have_m));\n'
        "return 0;}\n")
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    sizes = [int(v) for v in out]
    assert sizes[:4] == [C.sizeof(L.GemmProblem), C.sizeof(L.TpGroup), C.sizeof(L.Block), C.sizeof(L.GateSeg)]
    assert sizes[4:8] == [L.GemmProblem.M1.offset, L.GemmProblem.a_r1.offset, L.GemmProblem.alpha.offset, L.TpGroup.coeff.offset]
    assert sizes[8:10] == [C.sizeof(L.KwInstr), L.KwInstr.u.offset]
    assert sizes[10:15] == [C.sizeof(L.GemmSegment), C.sizeof(L.LayerDesc), C.sizeof(L.LayerRadial), C.sizeof(L.LayerFwdArgs),
                            C.sizeof(L.LayerBwdArgs)]
    assert sizes[15:21] == [L.GemmSegment.M1.offset, L.LayerDesc.gate.offset, L.LayerDesc.alphas.offset,
                            L.LayerDesc.tp_bwd_x_overwrites.offset, L.LayerFwdArgs.rad.offset, L.LayerBwdArgs.gb_hidden.offset]
    assert sizes[21:24] == [L.LayerFwdArgs.x_cf.offset, L.LayerBwdArgs.kw_ws.offset, L.LayerRadial.z.offset]
    assert sizes[24:30] == [C.sizeof(L.RadialStackItem), C.sizeof(L.MlpNet), C.sizeof(L.KwMultiItem), C.sizeof(L.KwStackItem),
                            L.RadialStackItem.g_rows.offset, L.LayerBwdArgs.have_m.offset]


def test_limits_agree_with_generated_header():
    gen = open(os.path.join(ROOT, "equivariant-nn-zoo_amd", "csrc", "e3k_cg_gen.h")).read()
    vals = {k: int(v) for k, v in re.findall(r"#define (E3K_L[123]MAX|E3K_MAXQ) (\d+)", gen)}
    from e3_layers_amd.backend import lib as L
    from e3_layers_amd.nn import core

    assert (vals["E3K_L1MAX"], vals["E3K_L2MAX"], vals["E3K_L3MAX"]) == (core.TP_L1MAX, core.TP_L2MAX, core.TP_L3MAX)
    assert vals["E3K_MAXQ"] == L.TP_MAXQ
    for l1 in range(core.TP_L1MAX + 1):
        m = re.search(r"struct Slots<%d> \{.*?L2\[\d+\] = \{([^}]*)\};.*?L3\[\d+\] = \{([^}]*)\};" % l1, gen, re.S)
        l2s = [int(v) for v in m.group(1).split(",")]
        l3s = [int(v) for v in m.group(2).split(",")]
        assert list(zip(l2s, l3s)) == core.tp_slots(l1)


def test_product_path_has_no_oracle_import():
    """The shipped package must never import the oracle (or the reference)."""
    pkg = os.path.join(ROOT, "equivariant-nn-zoo_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(base, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), os.path.join(base, f)
                assert "/root/reference" not in text, os.path.join(base, f)


def test_switches_live_in_one_table_and_the_product_library_reads_no_environment():
    """the C library's tuning constants are compile-time in the product build (getenv only behind
    E3K_DEBUG_KNOBS: `make dbg`), and the Python package reads the environment in backend/tuning.py only -- every knob it reads is
    documented there."""
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "equivariant-nn-zoo_amd", "csrc")
    for fn in sorted(os.listdir(csrc)):
        if not fn.endswith((".hip", ".h")):
            continue
        text = open(os.path.join(csrc, fn)).read()
        if fn == "e3k_common.h":
            assert text.count("getenv") >= 1 and "#ifdef E3K_DEBUG_KNOBS" in text
            continue
        assert "getenv" not in text, fn
    pkg = os.path.join(root, "equivariant-nn-zoo_amd", "e3_layers_amd")
    used = set()
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if not fn.endswith(".py"):
                continue
            text = open(os.path.join(dirpath, fn)).read()
            if fn != "tuning.py":
                assert "os.environ" not in text and "getenv" not in text, os.path.join(dirpath, fn)
            used |= set(re.findall(r'_knob\("(E3K_[A-Z0-9_]+)"\)', text))
    sys.path.insert(0, os.path.join(root, "equivariant-nn-zoo_amd"))
    from e3_layers_amd.backend import tuning

    assert used and used <= set(tuning.KNOBS), used - set(tuning.KNOBS)
    assert set(tuning.KNOBS) - used == set(), set(tuning.KNOBS) - used      # no documented knob that nothing reads
    assert all(len(doc) > 10 and kind in ("path", "threshold", "accuracy", "debug") for _, kind, doc in tuning.KNOBS.values())
