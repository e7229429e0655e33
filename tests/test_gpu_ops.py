"""GPU parity: every HIP operator (through the C ABI) against the float64 oracle on the same
seeded inputs.  Tolerances are normwise relative errors; fp32 kernels vs a float64 reference,
1e-5 is the north-star bound (BASELINE.json), most ops sit near 1e-7."""
import math

import pytest
import torch

from oracle import e3ref
from tests.util import from_cf, rel_err, to_cf

pytestmark = pytest.mark.gpu

TOL = 1e-5
GTOL = 2e-5


def _grads(outs, ins, seeds):
    return torch.autograd.grad(outs, ins, seeds, allow_unused=True)


# ------------------------------------------------------------------------------------------
def test_library_loads_and_limits(dev):
    from e3_layers_amd.backend import lib as L
    import ctypes as C

    lib = L.load()
    assert lib.e3k_version() >= 100
    a, b, c = C.c_int(), C.c_int(), C.c_int()
    lib.e3k_tp_limits(C.byref(a), C.byref(b), C.byref(c))
    from e3_layers_amd.nn.core import TP_L1MAX, TP_L2MAX, TP_L3MAX

    assert (a.value, b.value, c.value) == (TP_L1MAX, TP_L2MAX, TP_L3MAX)


def test_cpu_tensor_fails_loudly():
    from e3_layers_amd.nn import Linear

    lin = Linear("4x0e", "4x0e")
    with pytest.raises(RuntimeError, match="GPU only"):
        lin(torch.randn(3, 4))


@pytest.mark.parametrize("rows", [1, 300])
def test_relayout(dev, rows):
    from e3_layers_amd.backend import ops
    from e3_layers_amd.nn.core import irreps_blocks
    from e3_layers_amd.o3 import Irreps

    irreps = Irreps("8x0e+4x1o+3x1o+5x2e+64x3o")
    x = torch.randn(rows, irreps.dim, device=dev)
    y = ops.relayout(x, irreps_blocks(irreps), True)
    assert torch.equal(y.cpu(), to_cf(x.cpu(), irreps))
    assert torch.equal(ops.relayout(y, irreps_blocks(irreps), False), x)


@pytest.mark.parametrize("in_layout,out_layout", [("e3nn", "e3nn"), ("cf", "cf"), ("e3nn", "cf"), ("cf", "e3nn")])
def test_linear(dev, in_layout, out_layout):
    from e3_layers_amd.nn import Linear

    torch.manual_seed(0)
    ir_in, ir_out = "8x0e+4x1o+3x1o+5x2e+2x0o", "6x0e+7x1o+2x2e+3x3o+5x0e"
    lin = Linear(ir_in, ir_out, biases=True).to(dev)
    with torch.no_grad():
        lin.bias.normal_()
    ref = e3ref.Linear(ir_in, ir_out, biases=True).double()
    ref.load_state_dict({k: v.cpu() for k, v in lin.state_dict().items()})
    rows = 333
    x = torch.randn(rows, lin.irreps_in.dim, dtype=torch.float64)
    xin = (to_cf(x, ir_in) if in_layout == "cf" else x).float().to(dev).requires_grad_(True)
    y = lin(xin, in_layout=in_layout, out_layout=out_layout)
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)
    y_cmp = from_cf(y.cpu(), ir_out) if out_layout == "cf" else y.cpu()
    assert rel_err(y_cmp, yr) < TOL
    seed = torch.randn_like(yr)
    seed_dev = (to_cf(seed, ir_out) if out_layout == "cf" else seed).float().to(dev)
    gx, gw, gb = _grads(y, [xin, lin.weight, lin.bias], seed_dev)
    rx, rw, rb = _grads(yr, [xr, ref.weight, ref.bias], seed)
    gx_cmp = from_cf(gx.cpu(), ir_in) if in_layout == "cf" else gx.cpu()
    assert rel_err(gx_cmp, rx) < GTOL
    assert rel_err(gw, rw) < GTOL
    assert rel_err(gb, rb) < GTOL


def test_linear_large_shapes(dev):
    """The shapes of config_energy layer 3's node linear: 64-channel blocks, vector loads."""
    from e3_layers_amd.nn import Linear

    torch.manual_seed(1)
    ir = "64x0e+64x0o+64x1e+64x1o+64x2e+64x2o"
    lin = Linear(ir, ir).to(dev)
    ref = e3ref.Linear(ir, ir).double()
    ref.load_state_dict({k: v.cpu() for k, v in lin.state_dict().items()})
    x = torch.randn(1000, lin.irreps_in.dim, dtype=torch.float64)
    xin = to_cf(x, ir).float().to(dev).requires_grad_(True)
    y = lin(xin, in_layout="cf", out_layout="cf")
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)
    assert rel_err(from_cf(y.cpu(), ir), yr) < TOL
    seed = torch.randn_like(yr)
    gx, gw = _grads(y, [xin, lin.weight], to_cf(seed, ir).float().to(dev))
    rx, rw = _grads(yr, [xr, ref.weight], seed)
    assert rel_err(from_cf(gx.cpu(), ir), rx) < GTOL
    assert rel_err(gw, rw) < GTOL


@pytest.mark.parametrize("m1,m2,k,n,gathered", [
    (1000, 3, 384, 64, False),     # the trailing Linear's shape: three 128-wide k tiles, rows = (node, m)
    (777, 5, 192, 256, False),     # second k tile half empty, four column tiles
    (333, 1, 64, 64, False),       # narrow form (K <= 64)
    (95, 3, 448, 192, False),      # k tail (448 = 3.5 tiles), rows not a multiple of the 32-row chunk
    (1, 1, 64, 64, False),         # a single row
    (1200, 3, 64, 128, True),      # gathered rows (the keyed self-connection's groups): node indices fetched one chunk ahead
    (640, 5, 192, 64, True),
    (4704, 1, 64, 1, False),       # one output column (the energy head's Linear): the weighted column sum, not a tile kernel
    (333, 3, 100, 1, False),
    (500, 1, 64, 1, True),         # ... gathered rows keep the tile kernel
])
def test_gemm_wgrad_through_the_c_abi(dev, m1, m2, k, n, gathered):
    """e3k_gemm_wgrad: B[k, n] += alpha * sum_rows A[row, k] G[row, n] for strided (node, component) rows, against float64
    torch -- the pipelined kernel's tile shapes, tails and the gathered-row form (csrc/e3k_gemm.hip: gemm_wgrad2_kernel)."""
    from e3_layers_amd.backend import lib as L

    torch.manual_seed(m1 + k + n)
    nodes = m1 + 37 if gathered else m1
    d_a, d_g = m2 * k + 16, m2 * n + 8        # row widths with slack: the blocks do not fill the rows
    a = torch.randn(nodes, d_a, device=dev)
    g = torch.randn(nodes, d_g, device=dev)
    out = torch.randn(k, n, device=dev)
    want = out.double().cpu()
    idx = torch.randperm(nodes)[:m1].to(torch.int32) if gathered else None
    rows = idx.long() if gathered else torch.arange(m1)
    a_blk = a.cpu().double()[rows][:, 8:8 + m2 * k].reshape(m1 * m2, k)          # cf layout inside the block: [m][k]
    g_blk = g.cpu().double()[rows][:, 4:4 + m2 * n].reshape(m1 * m2, n)
    want = want + 0.37 * a_blk.t() @ g_blk
    p = L.GemmProblem()
    p.A, p.A2, p.B, p.C, p.bias = a.data_ptr() + 4 * 8, None, out.data_ptr(), g.data_ptr() + 4 * 4, None
    idx_dev = idx.to(dev) if gathered else None
    p.row_index = idx_dev.data_ptr() if gathered else None
    p.group_dev = None
    p.M1, p.M2, p.N, p.K, p.V, p.accumulate = m1, m2, n, k, 0, 1
    p.a_r1, p.a_r2, p.a_k = d_a, k, 1
    p.b_k, p.b_n = n, 1
    p.c_r1, p.c_r2, p.c_n = d_g, n, 1
    p.alpha, p.act, p.act_cst = 0.37, 0, 1.0
    arr = (L.GemmProblem * 1)(p)
    L.check(L.load().e3k_gemm_wgrad(arr, 1, L.stream_ptr()), "e3k_gemm_wgrad")
    torch.cuda.synchronize()
    assert rel_err(out, want) < 2e-6


@pytest.mark.parametrize("m1,m2,k,n,accumulate,bias", [
    (1000, 3, 192, 64, 1, False),     # interior tiles, accumulating epilogue (all old values requested, then all stored)
    (1000, 3, 192, 64, 0, True),
    (333, 5, 96, 200, 1, True),       # edge tiles in rows and columns (per-element predicates), K tail
    (130, 1, 64, 72, 1, False),       # 64-row tiles of the small-grid form, column tail
    (4700, 1, 384, 64, 1, False),     # enough tiles for the 128-row form
    (4704, 1, 64, 1, 0, True),        # one output column (the energy head): a wave per row, not a tile kernel
    (333, 3, 100, 1, 1, False),
])
def test_gemm_forward_epilogues_through_the_c_abi(dev, m1, m2, k, n, accumulate, bias):
    """e3k_gemm: C[(r1, r2), n] = alpha * A B (+ C) (+ bias) on strided rows against float64 torch: interior and edge tiles of
    store_acc (csrc/e3k_gemm.hip), with and without the read-modify-write."""
    from e3_layers_amd.backend import lib as L

    torch.manual_seed(m1 + k + n + accumulate)
    d_a, d_c = m2 * k + 12, m2 * n + 20
    a = torch.randn(m1, d_a, device=dev)
    b = torch.randn(k, n, device=dev)
    c = torch.randn(m1, d_c, device=dev)
    bv = torch.randn(n, device=dev) if bias else None
    want = c.double().cpu().clone()
    a_blk = a.cpu().double()[:, 4:4 + m2 * k].reshape(m1 * m2, k)
    prod = 0.71 * a_blk @ b.cpu().double()
    if bias:
        prod = prod + bv.cpu().double()
    blk = want[:, 8:8 + m2 * n].reshape(m1 * m2, n)
    want[:, 8:8 + m2 * n] = ((blk + prod) if accumulate else prod).reshape(m1, m2 * n)
    p = L.GemmProblem()
    p.A, p.A2, p.B, p.C = a.data_ptr() + 4 * 4, None, b.data_ptr(), c.data_ptr() + 4 * 8
    p.bias = bv.data_ptr() if bias else None
    p.row_index, p.group_dev = None, None
    p.M1, p.M2, p.N, p.K, p.V, p.accumulate = m1, m2, n, k, 0, accumulate
    p.a_r1, p.a_r2, p.a_k = d_a, k, 1
    p.b_k, p.b_n = n, 1
    p.c_r1, p.c_r2, p.c_n = d_c, n, 1
    p.alpha, p.act, p.act_cst = 0.71, 0, 1.0
    arr = (L.GemmProblem * 1)(p)
    L.check(L.load().e3k_gemm(arr, 1, L.stream_ptr()), "e3k_gemm")
    torch.cuda.synchronize()
    assert rel_err(c, want) < 2e-6      # (the columns outside the block are part of the comparison: they must be untouched)


@pytest.mark.parametrize("m1,m2,n,ks,accumulate,bias,kdgrad", [
    (1000, 1, 192, (64, 256), 0, False, True),      # the trailing Linear's input gradient of `0e`: scalars' and gates' blocks (W^T: k-contiguous B)
    (1000, 3, 64, (64, 128, 32), 1, True, False),   # three links, accumulating head, bias, n-contiguous B
    (333, 5, 200, (96, 40), 0, False, True),        # edge tiles in rows and columns, K tails
    (130, 1, 72, (64, 64), 1, False, False),
])
def test_gemm_k_chain_through_the_c_abi(dev, m1, m2, n, ks, accumulate, bias, kdgrad):
    """e3k_gemm with a K-chain (round 6): C = sum_j alpha_j A_j B_j (+ C) (+ bias) in one pass -- the followers continue the head's K loop
    into the same accumulators -- against float64 torch.  A second, unchained problem shares the launch (the batch's tile ranges must
    skip the followers)."""
    from e3_layers_amd.backend import lib as L

    torch.manual_seed(m1 + n + sum(ks))
    d_a, d_c = m2 * sum(ks) + 12, m2 * n + 20
    a = torch.randn(m1, d_a, device=dev)
    c = torch.randn(m1, d_c, device=dev)
    bv = torch.randn(n, device=dev) if bias else None
    alphas = [0.71, 0.33, 1.9][:len(ks)]
    want = c.double().cpu().clone()
    total = torch.zeros(m1 * m2, n, dtype=torch.float64)
    probs, keep, pos = [], [], 4
    for j, (k, al) in enumerate(zip(ks, alphas)):
        b = torch.randn(n, k, device=dev) if kdgrad else torch.randn(k, n, device=dev)      # dgrad reads W^T: B[k, n] = W[n, k]
        keep.append(b)
        a_blk = a.cpu().double()[:, pos:pos + m2 * k].reshape(m1 * m2, k)
        total += al * a_blk @ (b.cpu().double().T if kdgrad else b.cpu().double())
        p = L.GemmProblem()
        p.A, p.A2, p.B, p.C = a.data_ptr() + 4 * pos, None, b.data_ptr(), c.data_ptr() + 4 * 8
        p.bias = bv.data_ptr() if (bias and j == 0) else None
        p.row_index, p.group_dev = None, None
        p.M1, p.M2, p.N, p.K, p.V, p.accumulate = m1, m2, n, k, 0, accumulate
        p.a_r1, p.a_r2, p.a_k = d_a, k, 1
        p.b_k, p.b_n = (1, k) if kdgrad else (n, 1)
        p.c_r1, p.c_r2, p.c_n = d_c, n, 1
        p.alpha, p.act, p.act_cst = al, 0, 1.0
        p.chain = len(ks) - 1 if j == 0 else 0
        probs.append(p)
        pos += m2 * k
    if bias:
        total = total + bv.cpu().double()
    blk = want[:, 8:8 + m2 * n].reshape(m1 * m2, n)
    want[:, 8:8 + m2 * n] = ((blk + total) if accumulate else total).reshape(m1, m2 * n)
    # the bystander: its own output, after the chain in the array
    a2, b2, c2 = torch.randn(500, 64, device=dev), torch.randn(64, 128, device=dev), torch.zeros(500, 128, device=dev)
    q = L.GemmProblem()
    q.A, q.A2, q.B, q.C, q.bias, q.row_index, q.group_dev = a2.data_ptr(), None, b2.data_ptr(), c2.data_ptr(), None, None, None
    q.M1, q.M2, q.N, q.K, q.V, q.accumulate = 500, 1, 128, 64, 0, 0
    q.a_r1, q.a_r2, q.a_k, q.b_k, q.b_n, q.c_r1, q.c_r2, q.c_n = 64, 64, 1, 128, 1, 128, 128, 1
    q.alpha, q.act, q.act_cst, q.chain = 1.0, 0, 1.0, 0
    arr = (L.GemmProblem * (len(probs) + 1))(*probs, q)
    L.check(L.load().e3k_gemm(arr, len(probs) + 1, L.stream_ptr()), "e3k_gemm")
    torch.cuda.synchronize()
    assert rel_err(c, want) < 2e-6
    assert rel_err(c2, a2.double() @ b2.double()) < 2e-6
    # a follower that does not repeat its head's output is refused
    probs[1].N = n + 4
    arr = (L.GemmProblem * len(probs))(*probs)
    assert L.load().e3k_gemm(arr, len(probs), L.stream_ptr()) != 0


@pytest.mark.parametrize("rows,width", [(4097, 1920), (700, 960), (33, 260)])
def test_fully_connected_net_few_rows_wide_output(dev, rows, width):
    """The radial MLP on the knot table: few rows, a wide last layer -- its dgrad is the split-K kernel (K = width),
    its forward the small-K kernel with one column tile per workgroup."""
    from e3_layers_amd.nn import FullyConnectedNet
    from e3_layers_amd.utils import activations

    torch.manual_seed(7)
    hs = [8, 64, 64, 64, width]
    net = FullyConnectedNet(hs, activations["ssp"]).to(dev)
    ref = e3ref.FullyConnectedNet(hs, "ssp").double()
    ref.load_state_dict({k: v.cpu() for k, v in net.state_dict().items()})
    x = torch.randn(rows, 8, dtype=torch.float64)
    xin = x.float().to(dev).requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    y, yr = net(xin), ref(xr)
    assert rel_err(y, yr) < TOL
    seed = torch.randn_like(yr)
    g = _grads(y, [xin] + list(net.parameters()), seed.float().to(dev))
    r = _grads(yr, [xr] + list(ref.parameters()), seed)
    for a, b in zip(g, r):
        assert rel_err(a, b) < GTOL


def test_fully_connected_net(dev):
    from e3_layers_amd.nn import FullyConnectedNet
    from e3_layers_amd.utils import activations

    torch.manual_seed(2)
    hs = [8, 64, 64, 64, 300]
    net = FullyConnectedNet(hs, activations["ssp"]).to(dev)
    ref = e3ref.FullyConnectedNet(hs, "ssp").double()
    ref.load_state_dict({k: v.cpu() for k, v in net.state_dict().items()})
    x = torch.randn(1234, 8, dtype=torch.float64)
    xin = x.float().to(dev).requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    y, yr = net(xin), ref(xr)
    assert rel_err(y, yr) < TOL
    seed = torch.randn_like(yr)
    params = list(net.parameters())
    g = _grads(y, [xin] + params, seed.float().to(dev))
    r = _grads(yr, [xr] + list(ref.parameters()), seed)
    for a, b in zip(g, r):
        assert rel_err(a, b) < GTOL


def test_fctp_self_connection(dev):
    from e3_layers_amd.nn import FullyConnectedTensorProduct

    torch.manual_seed(3)
    in1, in2, out = "16x0e+16x1o+8x2e+4x0o", "20x0e", "24x0e+16x0e+16x1o+8x2e+4x0o+3x3e"
    tp = FullyConnectedTensorProduct(in1, in2, out).to(dev)
    ref = e3ref.FullyConnectedTensorProduct(in1, in2, out).double()
    assert ref.weight.numel() == tp.weight.numel()
    ref.load_state_dict({k: v.cpu() for k, v in tp.state_dict().items()})
    rows = 257
    x = torch.randn(rows, tp.irreps_in1.dim, dtype=torch.float64)
    a = torch.randn(rows, 20, dtype=torch.float64)
    xin = to_cf(x, in1).float().to(dev).requires_grad_(True)
    ain = a.float().to(dev).requires_grad_(True)
    xr, ar = x.clone().requires_grad_(True), a.clone().requires_grad_(True)
    y, yr = tp(xin, ain), ref(xr, ar)
    assert rel_err(from_cf(y.cpu(), out), yr) < TOL
    seed = torch.randn_like(yr)
    gx, ga, gw = _grads(y, [xin, ain, tp.weight], to_cf(seed, out).float().to(dev))
    rx, ra, rw = _grads(yr, [xr, ar, ref.weight], seed)
    assert rel_err(from_cf(gx.cpu(), in1), rx) < GTOL
    assert rel_err(ga, ra) < GTOL
    assert rel_err(gw, rw) < GTOL


def test_gate(dev):
    from e3_layers_amd.nn import Gate

    torch.manual_seed(4)
    sc, gt, gd = "8x0e+8x0o", "4x0e+6x0e+5x0e", "4x1o+6x2e+5x1e"
    args = (sc, ["silu", "tanhlu"], gt, ["silu", "silu", "silu"], gd)
    g = Gate(*args)
    ref = e3ref.Gate(*args)
    assert str(g.irreps_in) == e3ref.irreps_str(ref.irreps_in)
    assert str(g.irreps_out) == e3ref.irreps_str(ref.irreps_out)
    x = torch.randn(100, g.irreps_in.dim, dtype=torch.float64)
    xin = to_cf(x, g.irreps_in).float().to(dev).requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    y, yr = g(xin), ref(xr)
    assert rel_err(y, yr) < TOL
    seed = torch.randn_like(yr)
    (gx,) = _grads(y, [xin], seed.float().to(dev))
    (rx,) = _grads(yr, [xr], seed)
    assert rel_err(from_cf(gx.cpu(), g.irreps_in), rx) < GTOL


@pytest.mark.parametrize("ls", [[0, 1, 2], [0, 1, 2, 3], [2, 1]])
@pytest.mark.parametrize("normalize", [True, False])
@pytest.mark.parametrize("normalization", ["component", "integral", "norm"])
def test_spherical_harmonics(dev, ls, normalize, normalization):
    from e3_layers_amd.backend import ops

    torch.manual_seed(5)
    v = torch.randn(500, 3, dtype=torch.float64) * 2.0
    vin = v.float().to(dev).requires_grad_(True)
    vr = v.clone().requires_grad_(True)
    y = ops.spherical_harmonics(vin, ls, normalize, normalization)
    yr = e3ref.spherical_harmonics(ls, vr, normalize, normalization)
    assert rel_err(y, yr) < TOL
    seed = torch.randn_like(yr)
    (g,) = _grads(y, [vin], seed.float().to(dev))
    (r,) = _grads(yr, [vr], seed)
    assert rel_err(g, r) < GTOL


@pytest.mark.parametrize("one_over_r,cutoff", [(True, "_poly_cutoff"), (False, "_poly_cutoff"), (False, "symmetricCutoff")])
def test_radial_basis(dev, one_over_r, cutoff):
    from e3_layers_amd import nn as pnn

    torch.manual_seed(6)
    mod = pnn.RadialBasisEncoding(4.0, True, "8x0e", cutoff=getattr(pnn, cutoff), one_over_r=one_over_r).to(dev)
    ref = e3ref.RadialBasisEncoding(4.0, True, "8x0e", cutoff=cutoff, one_over_r=one_over_r).double()
    ref.load_state_dict({k: v.cpu() for k, v in mod.state_dict().items()})
    r = torch.rand(777, dtype=torch.float64) * 4.6 + 0.4  # includes r > r_max
    if cutoff == "symmetricCutoff":
        r = r - 2.5  # negative arguments too
    rin = r.float().to(dev).requires_grad_(True)
    rr = r.clone().requires_grad_(True)
    y = mod({"input": rin}, {"input": ("edge", "1x0e")})[0]["radial_embedding"]
    yr = ref({"input": rr}, {"input": ("edge", "1x0e")})[0]["radial_embedding"]
    assert rel_err(y, yr) < TOL
    seed = torch.randn_like(yr)
    g = _grads(y, [rin, mod.basis.bessel_weights], seed.float().to(dev))
    rg = _grads(yr, [rr, ref.basis.bessel_weights], seed)
    assert rel_err(g[0], rg[0]) < 5e-5  # derivative of the degree-8 envelope near r_max loses digits in fp32
    assert rel_err(g[1], rg[1]) < GTOL


def _random_graph(n_nodes, avg_deg, seed):
    gen = torch.Generator().manual_seed(seed)
    e = n_nodes * avg_deg
    src = torch.randint(n_nodes, (e,), generator=gen)
    dst = torch.randint(n_nodes, (e,), generator=gen)
    dst[dst == src] = (dst[dst == src] + 1) % n_nodes
    # leave some nodes without in-edges / out-edges
    dst[dst == 0] = 1
    src[src == n_nodes - 1] = 2
    return torch.stack([src, dst])


def test_edge_vector(dev):
    from e3_layers_amd.backend import ops
    from e3_layers_amd.backend.graph import build_topology

    torch.manual_seed(7)
    n = 50
    ei = _random_graph(n, 9, 7)
    pos = torch.randn(n, 3, dtype=torch.float64)
    pin = pos.float().to(dev).requires_grad_(True)
    pr = pos.clone().requires_grad_(True)
    topo = build_topology(ei.to(dev), n)
    vec, length = ops.edge_vector(pin, topo)
    vr = pr[ei[1]] - pr[ei[0]]
    lr = torch.linalg.norm(vr, dim=-1)
    assert rel_err(vec, vr) < TOL and rel_err(length, lr) < TOL
    s1, s2 = torch.randn_like(vr), torch.randn_like(lr)
    (g,) = _grads([vec, length], [pin], [s1.float().to(dev), s2.float().to(dev)])
    (r,) = _grads([vr, lr], [pr], [s1, s2])
    assert rel_err(g, r) < GTOL


def test_topology_matches_stable_sort(dev):
    from e3_layers_amd.backend.graph import build_topology

    ei = _random_graph(40, 6, 3)
    t = build_topology(ei.to(dev), 40)
    perm = t.dst_perm.cpu().long()
    assert torch.equal(ei[1][perm], torch.sort(ei[1], stable=True).values)
    # ascending edge ids inside every destination segment
    ptr = t.dst_ptr.cpu().long()
    for n in range(40):
        seg = perm[ptr[n]:ptr[n + 1]]
        assert torch.all(seg[1:] > seg[:-1])
        assert torch.all(ei[1][seg] == n)
    assert int(ptr[-1]) == ei.shape[1]


@pytest.mark.parametrize("n,deg,shuffle", [(37, 7, False), (500, 15, True), (3, 150, True), (64, 1, True), (9, 0, False), (40, 300, True)])
def test_device_csr_build_is_the_stable_sort(dev, n, deg, shuffle):
    """csrc/e3k_graph.hip (count -> scan -> fill -> per-row rank sort) against the torch construction (stable argsort /
    bincount / cumsum / searchsorted) on the host: every list bit-identical, whatever order the fill atomics landed in.
    Cases: sparse rows, shuffled edge lists, rows longer than a wave (150 > 64), isolated nodes, no edges at all."""
    from e3_layers_amd.backend import graph as G

    gen = torch.Generator().manual_seed(n * 131 + deg)
    if deg == 0:
        ei = torch.zeros(2, 0, dtype=torch.long)
    else:
        e = n * deg
        ei = torch.stack([torch.randint(0, n, (e,), generator=gen), torch.randint(0, n, (e,), generator=gen)])
        ei[1, : e // 3] = ei[1, 0]                      # a hub: one long destination row
        if not shuffle:
            ei = ei[:, torch.argsort(ei[0] * n + ei[1], stable=True)]
    ref = G.build_topology(ei, n + 2)                   # two trailing isolated nodes
    got = G.build_topology(ei.to(dev), n + 2)
    G.check_indices()
    for key in G.TOPO_KEYS:
        a, b = getattr(ref, key[len("_e3k_"):]), getattr(got, key[len("_e3k_"):])
        assert b.dtype == torch.int32 and b.is_cuda
        assert torch.equal(a, b.cpu()), key
    # out-of-range endpoints are reported (asynchronously: by the next check)
    if deg:
        bad = ei.clone()
        bad[1, 0] = n + 5
        G.build_topology(bad.to(dev), n + 2)
        with pytest.raises(ValueError):
            G.check_indices()


@pytest.mark.parametrize("rows,n_keys", [(4623, 10), (1000, 64), (70, 3), (1, 5), (5000, 1), (1536, 80), (3000, 256), (2049, 65)])
def test_group_rows_kernel_is_the_stable_sort(dev, rows, n_keys):
    """e3k_group_rows (one single-workgroup launch) against the torch construction (stable argsort + counts + cumsum):
    the same permutation, bounds and representatives -- absent keys included."""
    from e3_layers_amd.nn.core import row_groups

    gen = torch.Generator().manual_seed(rows + n_keys)
    key = torch.randint(0, n_keys, (rows,), generator=gen)
    if n_keys > 2:
        key[key == 1] = 0                      # key 1 is absent
    ref = row_groups(key.clone(), n_keys)      # CPU tensor: the torch path
    got = row_groups(key.to(dev), n_keys)
    assert torch.equal(ref.perm, got.perm.cpu()) and torch.equal(ref.bounds, got.bounds.cpu())
    present = ref.bounds[:, 1] > 0
    assert torch.equal(ref.reps[present], got.reps.cpu()[present])
    assert int(got.reps.min()) >= 0 and int(got.reps.max()) < rows


def test_radial_table_matches_the_per_edge_mlp(dev, monkeypatch):
    """backend/radial_table.py: the radial MLP evaluated on a knot table + quadratic interpolation per edge against the
    MLP evaluated per edge (HIP, fp32) and against the float64 oracle; and the backward (ordered per-knot sums + the MLP's
    backward on the knots) against the per-edge backward, for the MLP weights and the Bessel frequencies."""
    from e3_layers_amd import nn as pnn
    from e3_layers_amd.backend import radial_table
    from e3_layers_amd.nn.core import FullyConnectedNet
    from e3_layers_amd.utils.utils import activations

    torch.manual_seed(5)
    e, width = 40_000, 192
    enc = pnn.RadialBasisEncoding(r_max=4.0, trainable=True, irreps_out=("8x0e", "edge_radial"), irreps_in=("1x0e", "edge_length")).to(dev)
    fc = FullyConnectedNet([8, 64, 64, 64, width], activations["ssp"]).to(dev)
    gen = torch.Generator().manual_seed(1)
    r = (torch.rand(e, generator=gen) * 3.3 + 0.7)
    r[:5] = torch.tensor([1e-5, 1e-4, 3.99999, 4.0, 4.7])         # the ends of the table and beyond the cutoff
    rd = r.to(dev)
    seed = torch.randn(e, width, generator=gen).to(dev)
    params = [enc.basis.bessel_weights] + list(fc.parameters())

    def run(table):
        monkeypatch.setattr(radial_table, "ENABLED", 1 if table else 0)
        emb, _ = enc({"input": rd}, {"input": ("edge", "1x0e")})
        emb = emb["radial_embedding"]
        assert radial_table.applicable(emb) == bool(table)
        w = radial_table.table_weights(fc, emb) if table else fc(emb)
        grads = torch.autograd.grad(w, params, seed)
        return w.detach(), [g.detach() for g in grads]

    w_ref, g_ref = run(False)
    w_tab, g_tab = run(True)
    assert bool(torch.isfinite(w_tab).all())
    assert rel_err(w_tab, w_ref) < 5e-6                  # two fp32 evaluations of the same function (the oracle decides below)
    assert float((w_tab[4] - w_tab[3]).abs().max()) == 0.0        # constant beyond r_max
    for a, b, p in zip(g_tab, g_ref, params):
        assert rel_err(a, b) < 2e-5, tuple(p.shape)
    # float64 oracle of the same function
    orc_enc = e3ref.RadialBasisEncoding(r_max=4.0, trainable=True, irreps_out=("8x0e", "edge_radial"), irreps_in=("1x0e", "edge_length")).double()
    orc_fc = e3ref.FullyConnectedNet([8, 64, 64, 64, width], "ssp").double()
    orc_enc.load_state_dict({k: v.detach().cpu().double() for k, v in enc.state_dict().items()})
    orc_fc.load_state_dict({k: v.detach().cpu().double() for k, v in fc.state_dict().items()})
    out, _ = orc_enc({"input": r.double()[5:]}, {"input": ("edge", "1x0e")})
    w_orc = orc_fc(out[next(iter(out))])
    err_tab, err_edge = rel_err(w_tab[5:], w_orc), rel_err(w_ref[5:], w_orc)
    assert err_tab < 5e-6 and err_tab < 1.5 * err_edge + 1e-7       # interpolation adds nothing to the fp32 noise of the MLP


def test_batch_from_device_resident_samples(dev):
    """Batch.from_data_list on samples that already live in HBM: same Batch as collating on the host and copying."""
    from e3_layers_amd.data import Batch
    from e3_layers_amd.data.synthetic import synth_qm9_list

    lst, attrs = synth_qm9_list(5, 9, None, 4.0)
    host = Batch.from_data_list([dict(s) for s in lst], dict(attrs))
    on_dev = Batch.from_data_list([{k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in s.items()} for s in lst],
                                  dict(attrs))
    assert set(host.keys()) == set(on_dev.keys())
    for key in host.keys():
        assert on_dev[key].is_cuda and torch.equal(host[key], on_dev[key].cpu()), key
    assert torch.equal(on_dev[3]["pos"].cpu(), host[3]["pos"])


@pytest.mark.parametrize("mul,left,out,sh_grad", [
    (16, "16x0e+16x1o+16x2e", "16x0e+16x1o+16x2e+16x1e+16x3o", True),
    # <= 32 channels: the two lane halves of a wave walk two edges at a time (odd degrees leave the last trip half empty);
    # without a gradient for sh the weight-gradient kernel takes that form too
    (32, "32x0e+32x1o+32x1e+32x2e", "32x0e+32x1o+32x1e+32x2e+32x2o", False),
    (24, "24x0e+24x1o+24x2e", "24x0e+24x1o+24x2e+24x3o", False),
    (64, "64x0e+64x0o+64x1e+64x1o+64x2e+64x2o", "64x0e+64x0o+64x1e+64x1o+64x2e+64x2o", True),
    (96, "96x1o+96x3e", "96x0e+96x1o+96x2e+96x3e+96x2o", True),
    (64, "64x0e+64x0o+64x1e+64x1o+64x2e+64x2o+64x3e+64x3o", "64x0e+64x0o+64x1e+64x1o+64x2e+64x2o+64x3e+64x3o", True),
])
def test_tp_fused_against_unfused_oracle(dev, mul, left, out, sh_grad):
    """fused gather + uvu product + per-destination reduce + node-side Linear  ==  the
    reference's gather -> TensorProduct -> per-edge Linear -> scatter (oracle, float64)."""
    from e3_layers_amd.backend import ops
    from e3_layers_amd.backend.graph import build_topology
    from e3_layers_amd.nn import TensorProductExpansion

    torch.manual_seed(8)
    sh_ir = "1x0e+1x1o+1x2e"
    n = 37
    ei = _random_graph(n, 7, 11)
    e = ei.shape[1]
    mod = TensorProductExpansion(left, (sh_ir, "edge_spherical"), (out, "edge_features"), "uvu", internal_weight=False).to(dev)
    ref = e3ref.TensorProductExpansion(left, (sh_ir, "edge_spherical"), (out, "edge_features"), "uvu", internal_weight=False).double()
    assert ref.tp.weight_numel == mod.tp.weight_numel
    ref.load_state_dict({k: v.cpu() for k, v in mod.state_dict().items()})
    x = torch.randn(n, mod.tp.irreps_in1.dim, dtype=torch.float64)
    vec = torch.randn(e, 3, dtype=torch.float64)
    sh = e3ref.spherical_harmonics([0, 1, 2], vec)
    w = torch.randn(e, mod.tp.weight_numel, dtype=torch.float64)
    xin = to_cf(x, left).float().to(dev).requires_grad_(True)
    shin = sh.float().to(dev).requires_grad_(sh_grad)
    win = w.float().to(dev).requires_grad_(True)
    topo = build_topology(ei.to(dev), n)
    mid = mod.tp.fused(xin, shin, win, topo)
    y = mod.linear(mid, in_layout="cf", out_layout="e3nn")
    xr, shr, wr = x.clone().requires_grad_(True), sh.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = e3ref.scatter(ref(left=xr[ei[0]], right=shr, weight=wr), ei[1], dim_size=n)
    assert rel_err(y, yr) < TOL
    seed = torch.randn_like(yr)
    rx, rsh, rw, rl = _grads(yr, [xr, shr, wr, ref.linear.weight], seed)
    if sh_grad:
        gx, gsh, gw, gl = _grads(y, [xin, shin, win, mod.linear.weight], seed.float().to(dev))
        assert rel_err(gsh, rsh) < GTOL
    else:
        gx, gw, gl = _grads(y, [xin, win, mod.linear.weight], seed.float().to(dev))
    assert rel_err(from_cf(gx.cpu(), left), rx) < GTOL
    assert rel_err(gw, rw) < GTOL
    assert rel_err(gl, rl) < GTOL
    # the per-sample module API (no reduction) agrees with the oracle too
    y_edges = mod(left=x[ei[0]].float().to(dev), right=sh.float().to(dev), weight=w.float().to(dev))
    yr_edges = ref(left=x[ei[0]], right=sh, weight=w)
    assert rel_err(y_edges, yr_edges) < TOL


def _cubic_coef64(r, r_max, target):
    """float64 restatement of e3k_rtable_bins' per-edge part: knot i and the four cubic Lagrange weights (on the power-of-two
    spacing radial_table.layout picks for the target knot count)"""
    from e3_layers_amd.backend.radial_table import layout

    knots, h = layout(r_max, target)
    x = (r.double() / h).clamp(0, knots)
    i = x.floor().clamp(1, knots - 2)
    t = (x - i).unsqueeze(1)
    c = torch.cat([-t * (t - 1) * (t - 2) / 6, (t + 1) * (t - 1) * (t - 2) / 2, -(t + 1) * t * (t - 2) / 2, (t + 1) * t * (t - 1) / 6], 1)
    return i.long(), c


@pytest.mark.parametrize("clustered", [False, True])
def test_knot_bins_are_the_stable_counting_sort(dev, clustered):
    """e3k_rtable_bins: knot, the four cubic weights, and the edges grouped by knot == a stable argsort (ascending edge id inside a
    knot), its row pointers and the <= 64-edge segment list -- also when thousands of edges share a handful of knots (real
    molecules: C-H 1.09 A, C-C 1.52 A; VERDICT r3: the rank sort of round 3 was O(len^2) there), and bit-identical run to run."""
    from e3_layers_amd.backend import radial_table

    gen = torch.Generator().manual_seed(3)
    e, knots, r_max = 70_001, 512, 4.0
    if clustered:
        centres = torch.tensor([1.09, 1.52, 1.43, 1.21, 2.5])
        r = centres[torch.randint(0, 5, (e,), generator=gen)] + 0.004 * torch.randn(e, generator=gen)
    else:
        r = torch.rand(e, generator=gen) * 4.4                      # some beyond r_max
    r[:4] = torch.tensor([0.0, 4.0, 3.99999, 1e-4])
    a = radial_table.build_bins(r.to(dev), r_max, knots)
    b = radial_table.build_bins(r.to(dev), r_max, knots)
    i_ref, c_ref = _cubic_coef64(r, r_max, knots)
    knots = a.knots
    assert (knots, a.spacing) == radial_table.layout(r_max, 512) and knots * a.spacing >= r_max
    bin_cpu = a.bin.cpu().long()
    # the spacing is a power of two: r / h and the offset inside the interval are exact in fp32 -- the knot is THE knot
    assert torch.equal(bin_cpu, i_ref)
    assert float((a.coef.cpu().double() - c_ref).abs().max()) < 1e-6
    assert float((a.coef.sum(1) - 1).abs().max()) < 1e-6                             # the weights interpolate constants exactly
    assert torch.equal(a.coef[1].cpu(), torch.tensor([0.0, 0.0, 0.0, 1.0]))          # r = r_max: the last knot
    perm_ref = torch.argsort(bin_cpu, stable=True).int()
    assert torch.equal(a.perm.cpu(), perm_ref)
    cnt = torch.bincount(bin_cpu, minlength=knots + 1)            # (knots: the table's actual interval count from here on)
    ptr_ref = torch.zeros(knots + 2, dtype=torch.int64)
    ptr_ref[1:] = torch.cumsum(cnt, 0)
    assert torch.equal(a.ptr.cpu().long(), ptr_ref)
    seg_ref = torch.zeros(knots + 2, dtype=torch.int64)
    seg_ref[1:] = torch.cumsum((cnt + 63) // 64, 0)
    assert torch.equal(a.seg.cpu().long(), seg_ref)
    for name in ("bin", "coef", "ptr", "perm", "seg"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    if clustered:
        assert int(cnt.max()) > 5000
    # the transposed interpolation over these bins: an ordered sum per knot (segments of <= 64 edges combined in order) -- the same
    # bits run to run, and the float64 index_add within rounding
    width = 64
    g_w = torch.randn(e, width, generator=gen).to(dev)
    gt_a, gt_b = radial_table.interp_bwd_raw(g_w, a), radial_table.interp_bwd_raw(g_w, b)
    assert torch.equal(gt_a, gt_b)
    gt_ref = torch.zeros(knots + 1, width, dtype=torch.float64)
    for k in range(4):
        gt_ref.index_add_(0, bin_cpu - 1 + k, c_ref[:, k:k + 1] * g_w.cpu().double())
    assert rel_err(gt_a, gt_ref) < 2e-6


@pytest.mark.parametrize("left,out", [
    ("64x0e+64x0o+64x1e+64x1o+64x2e+64x2o", "64x0e+64x0o+64x1e+64x1o+64x2e+64x2o"),
    ("64x0e", "64x0e+64x1o+64x2e"),
    # l_max 3: the split (two waves per group) kernels
    ("64x0e+64x0o+64x1e+64x1o+64x2e+64x2o+64x3e+64x3o", "64x0e+64x0o+64x1e+64x1o+64x2e+64x2o+64x3e+64x3o"),
])
def test_tp_with_in_kernel_knot_table_is_interpolate_then_tp(dev, left, out):
    """e3k_tp_fwd_table / e3k_tp_bwd_x_table (the path weights interpolated from the radial knot table inside the kernel) ==
    e3k_rtable_interp_fwd into w[E, W] followed by e3k_tp_fwd / e3k_tp_bwd_x, bit for bit (same interpolation arithmetic, same
    order of accumulation); the interpolation and its transpose against float64; plans without the form say so."""
    from e3_layers_amd.backend import lib as L
    from e3_layers_amd.backend import ops, radial_table
    from e3_layers_amd.backend.graph import build_topology
    from e3_layers_amd.nn import TensorProductExpansion

    torch.manual_seed(21)
    n, knots, r_max = 300, 256, 5.0
    ei = _random_graph(n, 9, 14)
    e = ei.shape[1]
    mod = TensorProductExpansion(left, ("1x0e+1x1o+1x2e", "edge_spherical"), (out, "edge_features"), "uvu", internal_weight=False).to(dev)
    plan = mod.tp.plan
    lib = L.load()
    assert lib.e3k_tp_table_supported(plan.handle(dev)) == 1
    topo = build_topology(ei.to(dev), n)
    x = torch.randn(n, plan.d_in, device=dev)
    sh = torch.randn(e, 9, device=dev)
    g_out = torch.randn(n, plan.d_mid, device=dev)
    r = torch.rand(e) * 5.2
    r[:4] = torch.tensor([0.0, 5.0, 4.99999, 0.01])                                  # the ends of the table
    bins = radial_table.build_bins(r.to(dev), r_max, knots)
    table = torch.randn(bins.knots + 1, plan.w_numel, device=dev)
    w = radial_table.interp_fwd_raw(table, bins)
    i_ref, c_ref = _cubic_coef64(r, r_max, knots)
    assert torch.equal(bins.bin.cpu().long(), i_ref)
    t64 = table.cpu().double()
    w_ref = sum(c_ref[:, k:k + 1] * t64[i_ref - 1 + k] for k in range(4))
    assert rel_err(w, w_ref) < 1e-6
    ref_out = ops._tp_fwd_raw(x, sh, w, topo, plan)
    ref_gx = ops._tp_bwd_x_raw(sh, w, g_out, topo, plan)
    out_t = torch.empty_like(ref_out)
    gx_t = (torch.empty if plan.bwd_x_overwrites(dev) else torch.zeros)(n, plan.d_in, device=dev)
    L.check(lib.e3k_tp_fwd_table(plan.handle(dev), L.ptr(x), L.ptr(sh), L.ptr(table), L.ptr(bins.bin), L.ptr(bins.coef), L.ptr(topo.src),
                                 L.ptr(topo.dst_ptr), L.ptr(topo.dst_perm), n, e, L.ptr(out_t), L.stream_ptr()), "e3k_tp_fwd_table")
    L.check(lib.e3k_tp_bwd_x_table(plan.handle(dev), L.ptr(sh), L.ptr(table), L.ptr(bins.bin), L.ptr(bins.coef), L.ptr(g_out), L.ptr(topo.dst),
                                   L.ptr(topo.src_ptr), L.ptr(topo.src_perm), n, e, L.ptr(gx_t), L.stream_ptr()), "e3k_tp_bwd_x_table")
    torch.cuda.synchronize()
    assert torch.equal(out_t, ref_out)
    if plan.bwd_x_overwrites(dev):
        assert torch.equal(gx_t, ref_gx)
    else:                                   # (atomic accumulation across groups: the order is not fixed)
        assert rel_err(gx_t, ref_gx) < 1e-6
    # ---- the PACKED table (round 5: e3k_rtable_pack -> e3k_tp_fwd_ptable / e3k_tp_bwd_x_ptable): one 12-byte Taylor record per (knot,
    # weight), the cubic's two small coefficients in fp16.  (i) the packed in-kernel forms == interpolate-from-the-packed-table, then
    # multiply, bit for bit (the evaluation order this form is pinned to); (ii) the packed interpolation against the float64 cubic:
    # within the fp16 rounding of the two small terms, relative to the table's scale; on a SMOOTH table (what the kernels see) 2e-7.
    from e3_layers_amd.backend import conv_force

    # (on a SMOOTH table -- what the kernels see; the fixed fp16 scales of the record assume third differences ~1e-5 of the values,
    #  a table of independent normal rows overflows them, and the guard would veto it)
    radii = torch.arange(bins.knots + 1, dtype=torch.float64) * bins.spacing
    cols = torch.arange(plan.w_numel, dtype=torch.float64)
    smooth = (torch.sin(radii[:, None] * (1.0 + 5.0 * cols[None, :] / plan.w_numel)) * torch.exp(-0.2 * radii[:, None])).float().to(dev)
    packed = radial_table.pack_raw(smooth, bins.knots)
    w_p = radial_table.interp_packed_raw(packed, bins)
    assert bool(torch.isfinite(w_p).all())
    out_p = conv_force._tp_fwd_ptable(x, sh, packed, bins, topo, plan)
    gx_p = conv_force._tp_bwd_x_ptable(sh, packed, bins, g_out, topo, plan)
    torch.cuda.synchronize()
    assert torch.equal(out_p, ops._tp_fwd_raw(x, sh, w_p, topo, plan))
    if plan.bwd_x_overwrites(dev):
        assert torch.equal(gx_p, ops._tp_bwd_x_raw(sh, w_p, g_out, topo, plan))
    else:
        assert rel_err(gx_p, ops._tp_bwd_x_raw(sh, w_p, g_out, topo, plan)) < 1e-6
    s64 = smooth.cpu().double()
    ws_ref = sum(c_ref[:, k:k + 1] * s64[i_ref - 1 + k] for k in range(4))
    inner = (r > 0.1) & (r < r_max - 2 * bins.spacing)      # (in the first and the last interval the knot is clamped and the cubic is
                                                            #  evaluated outside its middle interval, |s| up to 3/2: the shipped
                                                            #  envelope is flat there, this test function is not)
    # (this table: 6 rad/A on knots 2^-6 A apart -- d2 = 4e-3 of the values, four times the shipped models' on their 2^-7 A knots:
    #  2^-11 d2 / 8 = 2.7e-7, plus the fp32 roundings)
    assert float((w_p.cpu().double() - ws_ref)[inner].abs().max()) < 5e-7 * float(s64.abs().max())
    assert float((w_p.cpu().double() - ws_ref).abs().max()) < 2e-6 * float(s64.abs().max())
    # the gradient of the table: tp_bwd_w -> g_w[E, W] -> transposed interpolation, against an index_add in float64; twice: same bits
    g_w, _ = ops._tp_bwd_w_raw(x, sh, None, g_out, topo, plan, False, True)
    # the FUSED input + weight gradient (e3k_tp_bwd_xw_ptable: what a packed-table layer's backward runs): g_x with the bits of
    # e3k_tp_bwd_x_ptable, g_w == tp_bwd_w's up to the order of the sums (<x, sum CG sh g> here, <sum CG x sh, g> there);
    # twice: same bits (no atomics on g_w)
    gx_f, gw_f = conv_force._tp_bwd_xw_ptable(x, sh, packed, bins, g_out, topo, plan)
    gx_f2, gw_f2 = conv_force._tp_bwd_xw_ptable(x, sh, packed, bins, g_out, topo, plan)
    torch.cuda.synchronize()
    assert torch.equal(gw_f, gw_f2)
    if plan.bwd_x_overwrites(dev):
        assert torch.equal(gx_f, gx_p) and torch.equal(gx_f, gx_f2)
    else:
        assert rel_err(gx_f, gx_p) < 1e-6
    assert rel_err(gw_f, g_w) < 1e-6      # (tp_bwd_w itself against the float64 oracle: test_tp_fused_against_unfused_oracle)
    # ... and with the weights streamed from w [E, W] (e3k_tp_bwd_xw: force training's materialised rows)
    gx_s, gw_s = ops._tp_bwd_xw_raw(x, sh, w_p, g_out, topo, plan)
    ref_gx_s = ops._tp_bwd_x_raw(sh, w_p, g_out, topo, plan)
    if plan.bwd_x_overwrites(dev):
        assert torch.equal(gx_s, ref_gx_s)
    else:
        assert rel_err(gx_s, ref_gx_s) < 1e-6
    assert torch.equal(gw_s, gw_f)      # (the same sums in the same order: the weights do not enter g_w)
    gt_a = radial_table.interp_bwd_raw(g_w, bins)
    gt_b = radial_table.interp_bwd_raw(g_w, bins)
    assert torch.equal(gt_a, gt_b)
    gt_ref = torch.zeros(bins.knots + 1, plan.w_numel, dtype=torch.float64)
    coef64, bin64, gw64 = bins.coef.cpu().double(), bins.bin.cpu().long(), g_w.cpu().double()
    for k in range(4):
        gt_ref.index_add_(0, bin64 - 1 + k, coef64[:, k:k + 1] * gw64)
    assert rel_err(gt_a, gt_ref) < 2e-6
    scale = torch.randn(e, device=dev)                                               # ... and with a per-edge factor (the slope table's)
    gt_s = radial_table.interp_bwd_raw(g_w, bins, scale=scale)
    gt_ref.zero_()
    for k in range(4):
        gt_ref.index_add_(0, bin64 - 1 + k, (coef64[:, k] * scale.cpu().double()).unsqueeze(1) * gw64)
    assert rel_err(gt_s, gt_ref) < 2e-6
    # a plan without the in-kernel form (odd channel count): refused, not silently wrong
    odd = TensorProductExpansion("24x0e+24x1o", ("1x0e+1x1o+1x2e", "edge_spherical"), ("24x0e+24x1o+24x2e", "edge_features"), "uvu",
                                 internal_weight=False).to(dev)
    assert lib.e3k_tp_table_supported(odd.tp.plan.handle(dev)) == 0
    rc = lib.e3k_tp_fwd_table(odd.tp.plan.handle(dev), L.ptr(x), L.ptr(sh), L.ptr(table), L.ptr(bins.bin), L.ptr(bins.coef), L.ptr(topo.src),
                              L.ptr(topo.dst_ptr), L.ptr(topo.dst_perm), n, e, L.ptr(out_t), L.stream_ptr())
    assert rc < 0


@pytest.mark.parametrize("left,out", [
    ("64x0e+64x0o+64x1e+64x1o+64x2e+64x2o", "64x0e+64x0o+64x1e+64x1o+64x2e+64x2o"),
    ("64x0e", "64x0e+64x1o+64x2e"),
])
def test_tp_table_second_order_forms_are_sums_of_first_order_kernels(dev, left, out):
    """The kernels of force training on the table (backend/conv_force.py) against compositions of the first-order kernels:
    e3k_tp_bwd_e_table (g_sh, g_r), e3k_tp_fwd_jvp_table, e3k_tp_bwd_x_dual_table, e3k_tp_bwd_w_dual."""
    from e3_layers_amd.backend import conv_force, ops, radial_table
    from e3_layers_amd.backend import lib as L
    from e3_layers_amd.backend.graph import build_topology
    from e3_layers_amd.nn import TensorProductExpansion

    torch.manual_seed(22)
    n, knots, r_max = 200, 128, 5.0
    ei = _random_graph(n, 9, 14)
    e = ei.shape[1]
    mod = TensorProductExpansion(left, ("1x0e+1x1o+1x2e", "edge_spherical"), (out, "edge_features"), "uvu", internal_weight=False).to(dev)
    tp = mod.tp.plan
    assert L.load().e3k_tp_table2_supported(tp.handle(dev)) == 1
    topo = build_topology(ei.to(dev), n)
    x, x2 = torch.randn(n, tp.d_in, device=dev), torch.randn(n, tp.d_in, device=dev)
    sh, sh2 = torch.randn(e, 9, device=dev), torch.randn(e, 9, device=dev)
    g = torch.randn(n, tp.d_mid, device=dev)
    s2 = torch.randn(e, device=dev)
    bins = radial_table.build_bins((torch.rand(e) * 5.0).to(dev), r_max, knots)
    T, D = torch.randn(bins.knots + 1, tp.w_numel, device=dev), torch.randn(bins.knots + 1, tp.w_numel, device=dev)
    w, dw = radial_table.interp_fwd_raw(T, bins), radial_table.interp_fwd_raw(D, bins)
    # first backward: d/dsh through w, d/dr through dw/dr
    g_sh, g_r = conv_force._tp_bwd_e_table(x, sh, T, D, bins, g, topo, tp)
    g_w_ref, g_sh_ref = ops._tp_bwd_w_raw(x, sh, w, g, topo, tp, True, True)
    assert rel_err(g_sh, g_sh_ref) < 1e-5
    assert rel_err(g_r, (g_w_ref.double() * dw.double()).sum(1)) < 1e-5
    # the product rule in one walk
    jvp = conv_force._tp_fwd_jvp(x, x2, sh, sh2, T, D, bins, s2, topo, tp)
    ref = (ops._tp_fwd_raw(x2, sh, w, topo, tp).double() + ops._tp_fwd_raw(x, sh2, w, topo, tp).double()
           + ops._tp_fwd_raw(x, sh, s2.unsqueeze(1) * dw, topo, tp).double())
    assert rel_err(jvp, ref) < 1e-5
    gx = conv_force._tp_bwd_x_dual(sh, sh2, T, D, bins, s2, g, topo, tp)
    ref = ops._tp_bwd_x_raw(sh2, w, g, topo, tp).double() + ops._tp_bwd_x_raw(sh, s2.unsqueeze(1) * dw, g, topo, tp).double()
    assert rel_err(gx, ref) < 1e-5
    gw = conv_force._tp_bwd_w_dual(x, x2, sh, sh2, g, topo, tp)
    ref = (ops._tp_bwd_w_raw(x2, sh, None, g, topo, tp, False, True)[0].double()
           + ops._tp_bwd_w_raw(x, sh2, None, g, topo, tp, False, True)[0].double())
    assert rel_err(gw, ref) < 1e-5
    # the three walks of the u-sweep as one (e3k_tp_bwd_xw_dual, streamed rows): the input gradient with the bits of the streamed
    # dual kernel, the dual and the plain weight gradient against the kernels they replace
    gx_s = conv_force._tp_bwd_x_dual(sh, sh2, w, dw, None, s2, g, topo, tp)
    gx_f, gw_f, gwp_f = conv_force._tp_bwd_xw_dual(x, x2, sh, sh2, w, dw, s2, g, topo, tp, True)
    if tp.bwd_x_overwrites(dev):
        assert torch.equal(gx_f, gx_s)
    else:
        assert rel_err(gx_f, gx_s) < 1e-6
    assert rel_err(gw_f, gw) < 1e-6
    assert rel_err(gwp_f, ops._tp_bwd_w_raw(x, sh, None, g, topo, tp, False, True)[0]) < 1e-6
    gx_n, gw_n, none = conv_force._tp_bwd_xw_dual(x, x2, sh, sh2, w, dw, s2, g, topo, tp, False)
    assert none is None and torch.equal(gw_n, gw_f)
    # the first backward's three walks as one (e3k_tp_bwd_xe, streamed rows): g_x bit for bit, g_w likewise the fused kernel's, the
    # edge gradients (atomics across the groups: order-free sums) against e3k_tp_bwd_e_table
    gx_e, gsh_e, gr_e, gw_e = conv_force._tp_bwd_xe(x, sh, w, dw, g, topo, tp, True)
    gx_ref = ops._tp_bwd_x_raw(sh, w, g, topo, tp)
    if tp.bwd_x_overwrites(dev):
        assert torch.equal(gx_e, gx_ref)
    else:
        assert rel_err(gx_e, gx_ref) < 1e-6
    assert rel_err(gw_e, g_w_ref) < 1e-6
    assert rel_err(gsh_e, g_sh_ref) < 1e-5
    assert rel_err(gr_e, (g_w_ref.double() * dw.double()).sum(1)) < 1e-5
    assert conv_force._tp_bwd_xe(x, sh, w, dw, g, topo, tp, False)[3] is None
    # round 6: the edge gradients are per-item partials combined in a fixed order -- bit-identical run to run, and equal (up to the
    # order of six additions) to the atomic form of rounds 4-5
    for _ in range(3):
        again = conv_force._tp_bwd_xe(x, sh, w, dw, g, topo, tp, False)
        assert torch.equal(again[1], gsh_e) and torch.equal(again[2], gr_e)
        a_sh, a_r = conv_force._tp_bwd_e_table(x, sh, T, D, bins, g, topo, tp)
        assert torch.equal(a_sh, g_sh) and torch.equal(a_r, g_r)
    monkey_atomics = conv_force.EDGE_ATOMICS
    conv_force.EDGE_ATOMICS = 1
    try:
        at = conv_force._tp_bwd_xe(x, sh, w, dw, g, topo, tp, False)
        assert rel_err(at[1], gsh_e) < 1e-6 and rel_err(at[2], gr_e) < 1e-6
        at_sh, at_r = conv_force._tp_bwd_e_table(x, sh, T, D, bins, g, topo, tp)
        assert rel_err(at_sh, g_sh) < 1e-6 and rel_err(at_r, g_r) < 1e-6
    finally:
        conv_force.EDGE_ATOMICS = monkey_atomics


@pytest.mark.parametrize("n_basis", [8, 48])
def test_slope_table_is_the_derivative_of_the_radial_mlp(dev, n_basis):
    """conv_native.RadialStackFn(slope=...): D = d/dr fc(basis(r)) on the knots (csrc/e3k_slope.hip: the hidden chain's forward-mode
    derivative per knot in float64, last layer in fp32) against the float64 oracle's autograd derivative, and its gradient w.r.t. the
    MLP weights and the Bessel frequencies against the oracle's double backward.  ``n_basis=48``: more than 32 trainable Bessel
    frequencies (the frequency level of the weight-gradient kernel summed only the first 32)."""
    from e3_layers_amd import nn as pnn
    from e3_layers_amd.backend import conv_native, radial_table
    from e3_layers_amd.configs import config_energy_force
    from e3_layers_amd.configs.layer_configs import addEnergyOutput, addForceOutput, featureModel
    from e3_layers_amd.utils import build

    torch.manual_seed(3)
    if n_basis == 8:
        tree = config_energy_force.get_config().model_config
    else:
        tree = addForceOutput(addEnergyOutput(featureModel(n_dim=64, l_max=2, edge_spherical="1x0e+1x1o+1x2e", node_attrs="16x0e",
                                                           edge_radial=f"{n_basis}x0e", num_types=20, num_layers=2, r_max=5.0),
                                              None, output_key="energy"))
    model = build(tree).to(dev)
    func = model.func
    layers = [getattr(func, f"layer{i}") for i in range(2)]
    enc = func.radial_basis
    knots, r_max = radial_table.KNOTS_SLOPE, 5.0
    b, c = enc.basis, enc.cutoff
    radii = radial_table.knot_radii(r_max, knots, dev)
    from e3_layers_amd.backend import ops

    rows = ops.radial_basis(radii, b.bessel_weights, b.r_max, b.r_min, c.p, b.one_over_r, c.cutoff.kind)
    plans = [l._block_plan() for l in layers]
    assert all(conv_native.native_layer(p) is not None for p in plans)
    weights = []
    for l in layers:
        fc = list(l.conv.fc.children())
        weights.append(fc[-1].weight)
        weights.extend(m.weight for m in fc[:-1])
    knots, spacing = radial_table.layout(r_max, knots)
    slope = (radii, r_max, float(b.r_min), float(c.p), int(b.one_over_r), int(c.cutoff.kind))
    outs = conv_native.RadialStackFn.apply(rows, plans, True, None, slope, b.bessel_weights, *weights)
    T, D = outs[:2], outs[2:]
    # float64 oracle of layer l's radial MLP and its derivative along r
    for li, l in enumerate(layers):
        orc_enc = e3ref.RadialBasisEncoding(r_max=r_max, trainable=True, irreps_out=(f"{n_basis}x0e", "edge_radial"), irreps_in=("1x0e", "edge_length")).double()
        orc_enc.load_state_dict({k: v.detach().cpu().double() for k, v in enc.state_dict().items()})
        hs = [n_basis, 64, 64, 64, plans[li].last_spec.d_out]
        orc_fc = e3ref.FullyConnectedNet(hs, "ssp").double()
        orc_fc.load_state_dict({k: v.detach().cpu().double() for k, v in l.conv.fc.state_dict().items()})
        rr = radii.cpu().double().clone().requires_grad_(True)
        o, _ = orc_enc({"input": rr}, {"input": ("edge", "1x0e")})
        f = orc_fc(o[next(iter(o))])
        cols = torch.randn(f.shape[1], dtype=torch.float64)
        (df_c,) = torch.autograd.grad((f * cols).sum(), rr, create_graph=True)        # d/dr of a random combination of the columns
        lo, hi = int(0.6 / spacing), int(r_max / spacing) - 1                          # (no edge is shorter than 0.6 A; from r_max on the rows are flat)
        assert rel_err(T[li][lo:hi], f[lo:hi].detach()) < 2e-6
        got = (D[li].double().cpu() * cols).sum(1)
        assert rel_err(got[lo:hi], df_c[lo:hi].detach()) < 2e-6, li
        # gradients of <g, D> w.r.t. the parameters
        gD = torch.zeros_like(D[li])
        gD[lo:hi] = torch.randn(hi - lo, D[li].shape[1], device=dev)
        params = [l.conv.fc.layer3.weight, l.conv.fc.layer0.weight, enc.basis.bessel_weights]
        got_g = torch.autograd.grad(D[li], params, gD, retain_graph=True)
        # oracle: D = d f / d r per column -> <gD, D> = sum_k sum_c gD[k, c] df_c/dr[k]: forward-mode via a dummy direction
        eps = torch.zeros_like(rr, requires_grad=True)
        o2, _ = orc_enc({"input": rr.detach() + eps}, {"input": ("edge", "1x0e")})
        f2 = orc_fc(o2[next(iter(o2))])
        (jv,) = torch.autograd.grad(f2, eps, gD.cpu().double(), create_graph=True)   # jv[k] = sum_c gD[k, c] d f_c / d r_k
        oparams = [orc_fc.layer3.weight, orc_fc.layer0.weight, orc_enc.basis.bessel_weights]
        ref_g = torch.autograd.grad(jv.sum(), oparams)
        for a_, b_, nm in zip(got_g, ref_g, ("last", "first hidden", "bessel")):
            assert rel_err(a_, b_) < 5e-5, (li, nm)


def test_tp_repeated_sh_degree_shares_an_input_block(dev):
    """An edge_spherical with a degree that repeats ('1x1o+1x1e') opens a second group on the same input block: the
    backward w.r.t. x must ADD the groups' contributions (it stored, so the last group won: ADVICE r1)."""
    from e3_layers_amd.backend.graph import build_topology
    from e3_layers_amd.nn import TensorProductExpansion

    torch.manual_seed(12)
    left, sh_ir, out = "16x0e+16x1o+16x1e", "1x0e+1x1o+1x1e", "16x0e+16x0o+16x1o+16x1e+16x2e"
    n = 23
    ei = _random_graph(n, 5, 9)
    e = ei.shape[1]
    mod = TensorProductExpansion(left, (sh_ir, "edge_spherical"), (out, "edge_features"), "uvu", internal_weight=False).to(dev)
    ref = e3ref.TensorProductExpansion(left, (sh_ir, "edge_spherical"), (out, "edge_features"), "uvu", internal_weight=False).double()
    assert ref.tp.weight_numel == mod.tp.weight_numel
    ref.load_state_dict({k: v.cpu() for k, v in mod.state_dict().items()})
    x = torch.randn(n, mod.tp.irreps_in1.dim, dtype=torch.float64)
    sh = torch.randn(e, 7, dtype=torch.float64)
    w = torch.randn(e, mod.tp.weight_numel, dtype=torch.float64)
    xin = to_cf(x, left).float().to(dev).requires_grad_(True)
    shin = sh.float().to(dev).requires_grad_(True)
    win = w.float().to(dev).requires_grad_(True)
    topo = build_topology(ei.to(dev), n)
    y = mod.linear(mod.tp.fused(xin, shin, win, topo), in_layout="cf", out_layout="e3nn")
    xr, shr, wr = x.clone().requires_grad_(True), sh.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = e3ref.scatter(ref(left=xr[ei[0]], right=shr, weight=wr), ei[1], dim_size=n)
    assert rel_err(y, yr) < TOL
    seed = torch.randn_like(yr)
    gx, gsh, gw = _grads(y, [xin, shin, win], seed.float().to(dev))
    rx, rsh, rw = _grads(yr, [xr, shr, wr], seed)
    assert rel_err(from_cf(gx.cpu(), left), rx) < GTOL
    assert rel_err(gw, rw) < GTOL
    assert rel_err(gsh, rsh) < GTOL


def test_tp_empty_and_isolated(dev):
    """No edges at all, and nodes without in-edges: outputs are exact zeros."""
    from e3_layers_amd.backend.graph import build_topology
    from e3_layers_amd.nn import TensorProductExpansion

    left, sh_ir = "8x0e+8x1o", "1x0e+1x1o+1x2e"
    mod = TensorProductExpansion(left, (sh_ir, "s"), (left, "o"), "uvu", internal_weight=False).to(dev)
    n = 5
    x = torch.randn(n, 32, device=dev)
    ei = torch.zeros(2, 0, dtype=torch.long, device=dev)
    topo = build_topology(ei, n)
    mid = mod.tp.fused(x, torch.zeros(0, 9, device=dev), torch.zeros(0, mod.tp.weight_numel, device=dev), topo)
    assert mid.shape == (n, mod.tp.d_mid) and float(mid.abs().max()) == 0.0


def test_layer_norm(dev):
    from e3_layers_amd import nn as pnn

    torch.manual_seed(9)
    ir = "8x0e+8x1o+4x2e"
    mod = pnn.LayerNormalization(ir, ir).to(dev)
    ref = e3ref.LayerNormalization(ir, ir).double()
    with torch.no_grad():
        mod.std.uniform_(0.5, 1.5)
    ref.load_state_dict({k: v.cpu() for k, v in mod.state_dict().items()})
    x = torch.randn(64, 52, dtype=torch.float64)
    xin, xr = x.float().to(dev).requires_grad_(True), x.clone().requires_grad_(True)
    y = mod({"input": xin}, {})[0]["output"]
    yr = ref({"input": xr}, {})[0]["output"]
    assert rel_err(y, yr) < TOL
    seed = torch.randn_like(yr)
    g = _grads(y, [xin, mod.std], seed.float().to(dev))
    r = _grads(yr, [xr, ref.std], seed)
    assert rel_err(g[0], r[0]) < GTOL and rel_err(g[1], r[1]) < GTOL


def test_pooling(dev):
    from e3_layers_amd import nn as pnn

    n_nodes = torch.tensor([[3], [1], [5], [2]])
    seg = torch.repeat_interleave(torch.arange(4), n_nodes.view(-1))
    x = torch.randn(11, 1, dtype=torch.float64)
    for reduce in ("sum", "mean"):
        mod = pnn.Pooling("1x0e", "1x0e", reduce)
        xin = x.float().to(dev).requires_grad_(True)
        y = mod({"input": xin, "_n_nodes": n_nodes.to(dev), "_node_segment": seg.to(dev)}, {"input": ("node", "1x0e")})[0]["output"]
        yr = e3ref.scatter(x, seg, dim_size=4, reduce=reduce)
        assert rel_err(y, yr) < TOL
        (g,) = _grads(y, [xin], torch.ones_like(y))
        expect = torch.ones(11, 1) if reduce == "sum" else (1.0 / n_nodes.view(-1)[seg].double()).view(-1, 1)
        assert rel_err(g, expect) < TOL


def test_fctp_keyed_attrs_matches_generic_and_oracle(dev):
    """Self-connection over keyed node attributes (rows = table[species]): the per-key contracted
    path equals the generic outer-product path and the oracle; parameter gradients agree, and the
    attrs gradient summed per key agrees (it is deposited on each key's representative row)."""
    from e3_layers_amd.nn import FullyConnectedTensorProduct
    from e3_layers_amd.nn.core import set_row_key

    torch.manual_seed(11)
    in1, in2, out = "32x0e+32x1o+32x2e+16x0o", "20x0e", "48x0e+32x0e+32x1o+32x2e+16x0o"
    tp = FullyConnectedTensorProduct(in1, in2, out).to(dev)
    ref = e3ref.FullyConnectedTensorProduct(in1, in2, out).double()
    ref.load_state_dict({k: v.cpu() for k, v in tp.state_dict().items()})
    rows, n_types = 700, 7
    species = torch.randint(0, n_types - 1, (rows,))          # the last type never occurs: empty group
    table = torch.randn(n_types, 20, dtype=torch.float64)
    x = torch.randn(rows, tp.irreps_in1.dim, dtype=torch.float64)
    a = table[species]
    xin = to_cf(x, in1).float().to(dev).requires_grad_(True)
    ain = a.float().to(dev).requires_grad_(True)
    set_row_key(ain, species.to(dev), n_types)
    y = tp(xin, ain)                                           # keyed path
    xr, ar = x.clone().requires_grad_(True), a.clone().requires_grad_(True)
    yr = ref(xr, ar)
    assert rel_err(from_cf(y.cpu(), out), yr) < TOL
    ain2 = a.float().to(dev).requires_grad_(True)              # no key: generic path
    y2 = tp(xin, ain2)
    assert rel_err(y, y2) < TOL
    seed = torch.randn_like(yr)
    gx, ga, gw = _grads(y, [xin, ain, tp.weight], to_cf(seed, out).float().to(dev))
    rx, ra, rw = _grads(yr, [xr, ar, ref.weight], seed)
    assert rel_err(from_cf(gx.cpu(), in1), rx) < GTOL
    assert rel_err(gw, rw) < GTOL
    per_key = torch.zeros(n_types, 20, dtype=torch.float64).index_add_(0, species, ra)
    got = torch.zeros(n_types, 20, dtype=torch.float64).index_add_(0, species, ga.cpu().double())
    assert rel_err(got, per_key) < GTOL


def test_fused_activation_epilogue_and_grad_sink(dev):
    """strided_linear(act='ssp') == linear followed by the activation kernel (values and gradients);
    with a registered gradient sink the weight gradient lands in the flat buffer instead of a temporary."""
    from e3_layers_amd.backend import ops
    from e3_layers_amd.run.parallel import FlatGradients
    from e3_layers_amd.utils import act_second_moment_const

    torch.manual_seed(12)
    rows, k, n = 2000, 64, 192
    spec = ops.LinearSpec(k, n, [ops.LinInstr(0, 0, k, n, 1, 0, 0.125)], "e3nn", "e3nn", [], True, True, k * n)
    x = torch.randn(rows, k, device=dev, requires_grad=True)
    w = torch.nn.Parameter(torch.randn(k * n, device=dev))
    cst = act_second_moment_const("ssp")
    seed = torch.randn(rows, n, device=dev)
    y_f = ops.strided_linear(x, w, None, spec, act="ssp", act_cst=cst)
    y_u = ops.activation(ops.strided_linear(x, w, None, spec), "ssp", cst)
    assert rel_err(y_f, y_u) < 1e-6
    gf = _grads(y_f, [x, w], seed)
    gu = _grads(y_u, [x, w], seed)
    assert rel_err(gf[0], gu[0]) < 1e-5 and rel_err(gf[1], gu[1]) < 1e-5
    (gw_plain,) = _grads(ops.strided_linear(x, w, None, spec), [w], seed)
    flat = FlatGradients([w])
    flat.enable_direct_accumulation()
    try:
        flat.zero()
        ops.strided_linear(x, w, None, spec).backward(seed)
    finally:
        flat.disable_direct_accumulation()
    assert rel_err(flat.gather(), gw_plain) < 1e-5        # accumulated in place by the wgrad kernel
    assert rel_err(w.grad, gw_plain) < 1e-5             # .grad is a view of the same buffer


@pytest.mark.parametrize("ema,clip,wd", [(None, None, 0.0), (0.999, 0.5, 0.0), (0.99, None, 0.01)])
def test_fused_adam_ema_matches_torch(dev, ema, clip, wd):
    """e3k_adam_ema_step == clip_grad_norm_ + torch.optim.Adam.step + torch_ema update (the reference's
    trainer.py:374-386 sequence), run in float64 on the CPU, over 12 steps of seeded gradients."""
    from e3_layers_amd.run.optim import FusedAdamEMA

    torch.manual_seed(13)
    shapes = [(64, 33), (7,), (1000,), (5, 5, 3)]          # offsets that need padding
    ps = [torch.nn.Parameter(torch.randn(s, device=dev)) for s in shapes]
    ref = [torch.nn.Parameter(p.detach().cpu().double().clone()) for p in ps]
    opt = FusedAdamEMA(ps, lr=1e-2, weight_decay=wd, ema_decay=ema, max_grad_norm=clip)
    ropt = torch.optim.Adam(ref, lr=1e-2, weight_decay=wd)
    shadow = [p.detach().clone() for p in ref]
    gen = torch.Generator().manual_seed(99)
    for k in range(1, 13):
        opt.zero_grad()
        for p, r in zip(ps, ref):
            g = torch.randn(r.shape, dtype=torch.float64, generator=gen) * (3.0 if k % 4 == 0 else 0.1)
            p.grad.copy_(g.float().to(dev))
            r.grad = g.float().double()
        if clip is not None:
            torch.nn.utils.clip_grad_norm_(ref, clip)
        ropt.step()
        opt.step()
        if ema is not None:
            d = min(ema, (1 + k) / (10 + k))
            for s, r in zip(shadow, ref):
                s.sub_((1 - d) * (s - r.detach()))
    for p, r in zip(ps, ref):
        assert rel_err(p, r) < 1e-5
    assert opt.steps_taken == 12
    if ema is not None:
        with opt.average_parameters():
            for p, s in zip(ps, shadow):
                assert rel_err(p, s) < 1e-5
        for p, r in zip(ps, ref):
            assert rel_err(p, r) < 1e-5               # restored


def test_fused_adam_skips_nonfinite_gradient(dev):
    from e3_layers_amd.run.optim import FusedAdamEMA

    p = torch.nn.Parameter(torch.randn(300, device=dev))
    opt = FusedAdamEMA([p], lr=1e-2, skip_nonfinite=True, ema_decay=0.9)
    before = p.detach().clone()
    opt.zero_grad()
    p.grad.fill_(1.0)
    p.grad[17] = float("nan")
    opt.step()
    assert torch.equal(p.detach(), before) and opt.steps_taken == 0
    p.grad.fill_(1.0)
    opt.step()
    assert opt.steps_taken == 1 and not torch.equal(p.detach(), before)
    assert abs(opt.last_grad_norm - 300 ** 0.5) < 1e-3


@pytest.mark.parametrize("r_max", [2.0, 4.0, 9999.0])
def test_radius_graph_kernel_bit_exact(dev, r_max):
    """Device computeEdgeIndex == the oracle's (and the CPU path's) edge list, bit for bit: same edges, same
    (graph, i, j) order, same per-graph counts — including graphs of 1 atom, > 64 atoms (several ballot rounds)
    and a pair sitting exactly on the cutoff (strict '<')."""
    from e3_layers_amd.data import computeEdgeIndex

    g = torch.Generator().manual_seed(0)
    n_nodes = [4, 1, 29, 150, 3, 70]
    pos = torch.randn(sum(n_nodes), 3, generator=g) * 2.5
    pos[1] = pos[0] + torch.tensor([2.0, 0.0, 0.0])
    nn_ = torch.tensor(n_nodes).view(-1, 1)
    odata = {"pos": pos.clone(), "_n_nodes": nn_.clone()}
    ref, _ = e3ref.compute_edge_index(odata, {}, r_max=r_max)
    ddata = {"pos": pos.to(dev), "_n_nodes": nn_.to(dev)}
    out, attrs = computeEdgeIndex(ddata, {}, r_max=r_max)
    assert out["edge_index"].is_cuda and out["edge_index"].dtype == torch.int64
    assert torch.equal(out["edge_index"].cpu(), ref["edge_index"])
    assert torch.equal(ddata["_n_edges"].cpu(), odata["_n_edges"])
    assert attrs["_n_edges"] == ("graph", "1x0e")


def test_radius_graph_kernel_keeps_existing_edges(dev):
    """Pre-existing (bond) edges longer than the cutoff survive and their attributes land on the new rows."""
    from e3_layers_amd.data import computeEdgeIndex
    from e3_layers_amd.data.synthetic import synth_qm9

    pos = torch.tensor([[0.0, 0, 0], [1.0, 0, 0], [5.0, 0, 0], [0.0, 0, 0], [0.5, 0, 0]])
    n_nodes = torch.tensor([[3], [2]])
    old = torch.tensor([[2, 0], [0, 2]])                      # unsorted on purpose
    bond = torch.tensor([[9.0], [7.0]])
    data = {"pos": pos.to(dev), "_n_nodes": n_nodes.to(dev), "edge_index": old.to(dev), "bond": bond.to(dev)}
    out, _ = computeEdgeIndex(data, {"bond": ("edge", "1x0e")}, r_max=1.5)
    assert out["edge_index"].tolist() == [[0, 0, 1, 2, 3, 4], [1, 2, 0, 0, 4, 3]]
    assert data["bond"].view(-1).tolist() == [0.0, 7.0, 0.0, 9.0, 0.0, 0.0]
    assert data["_n_edges"].view(-1).tolist() == [4, 2]
    # a realistic batch: the device list equals the list the CPU preprocessing produced
    batch = synth_qm9(3, 40)
    data = {"pos": batch["pos"].to(dev), "_n_nodes": batch["_n_nodes"].to(dev)}
    out, _ = computeEdgeIndex(data, {}, r_max=4.0)
    assert torch.equal(out["edge_index"].cpu(), batch["edge_index"])
    bad = {"pos": pos.to(dev), "_n_nodes": n_nodes.to(dev), "edge_index": torch.tensor([[0], [4]], device=dev)}
    with pytest.raises(ValueError, match="different graphs"):
        computeEdgeIndex(bad, {}, r_max=1.5)


@pytest.mark.parametrize("hs,act,rows", [([8, 64, 64, 64, 96], "ssp", 1234), ([32, 32, 32, 40], "silu", 130),
                                         ([6, 64, 20], "ssp", 1), ([8, 32, 32, 32, 32, 24], "ssp", 777)])
def test_fused_radial_mlp_hidden_chain(dev, hs, act, rows, monkeypatch):
    """The one-launch hidden chain (csrc/e3k_mlp.hip) == the oracle's FullyConnectedNet and == the per-layer
    kernels (E3K_FUSED_MLP=0), values and every gradient; ragged row counts, input widths 6/8/32, 1-4 hidden layers."""
    from e3_layers_amd.nn import FullyConnectedNet
    from e3_layers_amd.utils import activations

    torch.manual_seed(21)
    net = FullyConnectedNet(hs, activations[act]).to(dev)
    assert net.fused_hidden
    ref = e3ref.FullyConnectedNet(hs, act).double()
    ref.load_state_dict({k: v.cpu() for k, v in net.state_dict().items()})
    x = torch.randn(rows, hs[0], dtype=torch.float64)
    xin, xr = x.float().to(dev).requires_grad_(True), x.clone().requires_grad_(True)
    y, yr = net(xin), ref(xr)
    assert rel_err(y, yr) < TOL
    seed = torch.randn_like(yr)
    params = list(net.parameters())
    g = _grads(y, [xin] + params, seed.float().to(dev))
    r = _grads(yr, [xr] + list(ref.parameters()), seed)
    for a, b in zip(g, r):
        assert rel_err(a, b) < GTOL
    net.fused_hidden = False                      # per-layer kernels on the same parameters
    y2 = net(xin)
    g2 = _grads(y2, [xin] + params, seed.float().to(dev))
    assert rel_err(y, y2) < 1e-6
    for a, b in zip(g, g2):
        assert rel_err(a, b) < 1e-5
    with torch.no_grad():                         # inference: no pre-activations kept
        net.fused_hidden = True
        assert rel_err(net(xin), yr) < TOL


def test_fused_radial_mlp_falls_back_for_other_widths(dev):
    from e3_layers_amd.nn import FullyConnectedNet
    from e3_layers_amd.utils import activations

    assert not FullyConnectedNet([8, 16, 16, 30], activations["ssp"]).fused_hidden      # width 16: per-layer path
    assert not FullyConnectedNet([8, 64, 32, 30], activations["ssp"]).fused_hidden      # mixed widths
    assert not FullyConnectedNet([8, 30], None).fused_hidden


@pytest.mark.parametrize("act", ["ssp", "silu"])
def test_norm_activation(dev, act):
    """NormActivation (the 'norm' nonlinearity_type, message_passing.py:212-219): values, gradient and double
    backward against the oracle; includes exactly-zero channels (the epsilon clamp) and scalar irreps."""
    from e3_layers_amd.nn import NormActivation

    torch.manual_seed(14)
    ir = "8x0e+8x0o+4x1o+6x2e+3x3o"
    mod = NormActivation(ir, act, normalize=True, epsilon=1e-8)
    ref = e3ref.NormActivation(ir, act, normalize=True, epsilon=1e-8)
    x = torch.randn(150, mod.irreps_in.dim, dtype=torch.float64)
    x[3, 16:28] = 0.0                                   # the 4x1o block of one row: |x| = 0 -> clamped, output 0
    xin, xr = x.float().to(dev).requires_grad_(True), x.clone().requires_grad_(True)
    y, yr = mod(to_cf(xin, ir)), ref(xr)
    assert rel_err(y, yr) < TOL
    assert float(y.detach()[3, 16:28].abs().max()) == 0.0
    seed = torch.randn_like(yr)
    # on the clamped channels the slope is act(eps)/eps: for ssp that is (softplus(1e-8) - ln 2) / 1e-8, which no fp32
    # evaluation (this kernel's or torch's) resolves -- the float64 oracle says 0.5; leave those 12 entries out
    seed[3, 16:28] = 0.0
    (g,) = torch.autograd.grad(y, xin, seed.float().to(dev), create_graph=True)
    (r,) = torch.autograd.grad(yr, xr, seed, create_graph=True)
    assert rel_err(g, r) < GTOL
    c = torch.randn_like(r)
    (gg,) = torch.autograd.grad((g * c.float().to(dev)).sum(), xin)
    (rr,) = torch.autograd.grad((r * c).sum(), xr)
    assert rel_err(gg, rr) < 1e-4
    # the second backward with respect to BOTH inputs of the first (e3k_norm_act_bwd2: x and the incoming gradient)
    sd, sr = seed.float().to(dev).requires_grad_(True), seed.clone().requires_grad_(True)
    (g2,) = torch.autograd.grad(mod(to_cf(xin, ir)), xin, sd, create_graph=True)
    (r2,) = torch.autograd.grad(ref(xr), xr, sr, create_graph=True)
    gg_x, gg_s = torch.autograd.grad((g2 * c.float().to(dev)).sum(), [xin, sd])
    rr_x, rr_s = torch.autograd.grad((r2 * c).sum(), [xr, sr])
    keep = torch.ones_like(rr_s, dtype=torch.bool)
    keep[3, 16:28] = False                              # (the clamped channels: slope act(eps) / eps, see above)
    assert rel_err(gg_x, rr_x) < 1e-4
    assert rel_err(gg_s.cpu()[keep], rr_s[keep]) < 1e-4


@pytest.mark.parametrize("in1,v,out,rows", [
    ("16x0e+16x1o+16x2e", 8, "24x0e+16x1o+8x2e", 300),       # every block its own instruction, mixed widths
    ("64x0e+64x0o+64x1e+64x1o+64x2e+64x2o", 32, "80x0e+16x0o+64x1e+64x1o+64x2e+64x2o", 700),   # protein-like, > 64 rows of keys
    ("32x0e+32x1o+32x2e", 16, "32x0e+96x0e+32x1o+32x2e", 300),   # two output blocks read the same input block (scalars + gates)
])
def test_unkeyed_self_connection_shapes(dev, in1, v, out, rows):
    """General (un-keyed) node attributes (the diffusion configs: time-dependent attrs): the outer-product GEMM path
    against the oracle, forward and all three gradients, on mixed widths / shared input blocks / protein-like sizes."""
    from e3_layers_amd.nn import FullyConnectedTensorProduct

    torch.manual_seed(3)
    in2 = f"{v}x0e"
    tp = FullyConnectedTensorProduct(in1, in2, out).to(dev)
    ref = e3ref.FullyConnectedTensorProduct(in1, in2, out).double()
    ref.load_state_dict({k: t.cpu() for k, t in tp.state_dict().items()})
    x = torch.randn(rows, tp.irreps_in1.dim, dtype=torch.float64)
    a = torch.randn(rows, v, dtype=torch.float64)
    xr, ar = x.clone().requires_grad_(True), a.clone().requires_grad_(True)
    yr = ref(xr, ar)
    seed = torch.randn_like(yr)
    rx, ra, rw = _grads(yr, [xr, ar, ref.weight], seed)
    xin = to_cf(x, in1).float().to(dev).requires_grad_(True)
    ain = a.float().to(dev).requires_grad_(True)
    y = tp(xin, ain)
    assert rel_err(from_cf(y.cpu(), out), yr) < TOL
    gx, ga, gw = _grads(y, [xin, ain, tp.weight], to_cf(seed, out).float().to(dev))
    assert rel_err(from_cf(gx.cpu(), in1), rx) < GTOL
    assert rel_err(ga, ra) < GTOL and rel_err(gw, rw) < GTOL


@pytest.mark.gpu
def test_small_step_plumbing_kernels_match_torch(dev):
    """Round 5's launch-count work: the one-launch forms of OneHotEncoding's rows, Pooling's row pointers and a squared-error loss
    term (value AND gradient) against the torch ops they replace (`nn/embedding.py:271-281`, `nn/output.py:66-74`,
    `run/loss.py` with MSELoss)."""
    from e3_layers_amd.backend import lib as L
    from e3_layers_amd.backend import ops

    lib = L.load()
    gen = torch.Generator().manual_seed(3)
    idx = torch.randint(0, 7, (1234,), generator=gen).to(dev)
    one = torch.empty(1234, 7, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    L.check(lib.e3k_onehot(L.ptr(idx), 1234, 7, L.ptr(one), L.ptr(flag), L.stream_ptr()), "e3k_onehot")
    assert int(flag) == 0
    bad = idx.clone()
    bad[17] = 7                                   # one index outside [0, 7): a zero row and bit 2 of the flag
    one_bad = torch.empty(1234, 7, device=dev)
    L.check(lib.e3k_onehot(L.ptr(bad), 1234, 7, L.ptr(one_bad), L.ptr(flag), L.stream_ptr()), "e3k_onehot")
    assert int(flag) == 4 and float(one_bad[17].abs().sum()) == 0.0
    host = torch.zeros(1, dtype=torch.int32).pin_memory()
    L.check(lib.e3k_flag_fetch_clear(L.ptr(flag), host.data_ptr(), L.stream_ptr()), "e3k_flag_fetch_clear")
    torch.cuda.synchronize()
    assert int(host[0]) == 4 and int(flag) == 0   # handed over and cleared by one launch
    assert torch.equal(one, torch.nn.functional.one_hot(idx, 7).float())
    for g in (1, 255, 256, 257, 1000):
        counts = torch.randint(0, 40, (g,), generator=gen).to(dev)
        ptr = torch.empty(g + 1, dtype=torch.int32, device=dev)
        L.check(lib.e3k_counts_to_ptr(L.ptr(counts), g, L.ptr(ptr), L.stream_ptr()), "e3k_counts_to_ptr")
        ref = torch.zeros(g + 1, dtype=torch.int64, device=dev)
        ref[1:] = torch.cumsum(counts, 0)
        assert torch.equal(ptr.long(), ref)
    for n, rows_w in ((257, None), (257, 1), (3 * 1500, 3)):      # mean; per-entry weights; one weight per row of three components
        pred = torch.randn(n, generator=gen).to(dev).requires_grad_(True)
        target = torch.randn(n, generator=gen).to(dev)
        w = None if rows_w is None else torch.rand(n // rows_w, generator=gen).to(dev)
        loss = ops.sq_error(pred.view(-1, rows_w or 1), target.view(-1, rows_w or 1), None if w is None else w.view(-1, 1), 1e3)
        (g_k,) = torch.autograd.grad(2.0 * loss, pred)
        p64 = pred.detach().double().requires_grad_(True)
        d2 = (p64 - target.double()) ** 2
        ref = 1e3 * (d2.mean() if w is None else (d2.view(-1, rows_w) * w.double().view(-1, 1)).sum())
        (g_r,) = torch.autograd.grad(2.0 * ref, p64)
        assert abs(float(loss) - float(ref)) < 1e-5 * abs(float(ref))
        assert rel_err(g_k, g_r) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("left,out", [
    ("32x0e+32x1o+32x2e", "32x0e+32x1o+32x2e"),                                          # two edges per wave (HALF)
    ("16x0e+16x1o", "16x0e+16x1o+16x2e"),
    ("64x0e+64x1o+64x2e", "64x0e+64x1o"),                                                # 64 channels, not every slot present
    ("48x0e+48x1e+48x1o", "48x0e+48x0o+48x1e+48x1o+48x2e"),                               # idle lanes in the last chunk
    ("64x0e+64x0o+64x1e+64x1o+64x2e+64x2o", "64x0e+64x0o+64x1e+64x1o+64x2e+64x2o"),       # channel-complete
])
def test_tp_input_and_weight_gradient_in_one_walk_streamed(dev, left, out):
    """e3k_tp_bwd_xw (weights streamed from w [E, W]; every plan shape) == e3k_tp_bwd_x + e3k_tp_bwd_w: the input gradient bit for bit
    where that kernel stores (no atomics), the weight gradient up to the order of its sums; twice: the same bits."""
    from e3_layers_amd.backend import ops
    from e3_layers_amd.backend.graph import build_topology
    from e3_layers_amd.nn import TensorProductExpansion

    torch.manual_seed(31)
    n = 190
    ei = _random_graph(n, 8, 5)
    e = ei.shape[1]
    mod = TensorProductExpansion(left, ("1x0e+1x1o+1x2e", "edge_spherical"), (out, "edge_features"), "uvu", internal_weight=False).to(dev)
    plan = mod.tp.plan
    topo = build_topology(ei.to(dev), n)
    x = torch.randn(n, plan.d_in, device=dev)
    sh = torch.randn(e, 9, device=dev)
    w = torch.randn(e, plan.w_numel, device=dev)
    g = torch.randn(n, plan.d_mid, device=dev)
    gx_ref = ops._tp_bwd_x_raw(sh, w, g, topo, plan)
    gw_ref, _ = ops._tp_bwd_w_raw(x, sh, None, g, topo, plan, False, True)
    gx, gw = ops._tp_bwd_xw_raw(x, sh, w, g, topo, plan)
    gx2, gw2 = ops._tp_bwd_xw_raw(x, sh, w, g, topo, plan)
    torch.cuda.synchronize()
    assert torch.equal(gw, gw2)
    if plan.bwd_x_overwrites(dev):
        assert torch.equal(gx, gx_ref) and torch.equal(gx, gx2)
    else:
        assert rel_err(gx, gx_ref) < 1e-6
    assert rel_err(gw, gw_ref) < 1e-6


@pytest.mark.gpu
def test_type_index_outside_the_one_hot_range_is_reported(dev):
    """``torch.nn.functional.one_hot`` raises on an index outside [0, num_types) (``e3_layers/nn/embedding.py:271-281``);
    the one-launch kernel wrote a zero row and said nothing.  Now it ORs a bit into the device's persistent error flag, which
    reaches the host without a sync and is raised by ``check_indices()`` (or the next build / optimizer step) -- once."""
    from e3_layers_amd.backend import graph as G
    from e3_layers_amd.nn import OneHotEncoding

    G.check_indices()
    enc = OneHotEncoding(num_types=5, irreps_out=("5x0e", "one_hot"), irreps_in=("1x0e", "input"))
    good = torch.randint(0, 5, (300, 1)).to(dev)
    out, _ = enc({"input": good}, {"input": ("node", "1x0e")})
    G.check_indices()                                             # nothing to report
    assert torch.equal(out["one_hot"].argmax(1), good.view(-1))
    bad = good.clone()
    bad[7, 0] = 5
    out, _ = enc({"input": bad}, {"input": ("node", "1x0e")})
    assert float(out["one_hot"][7].abs().sum()) == 0.0
    with pytest.raises(ValueError, match="type index outside"):
        G.check_indices()
    enc({"input": good.clone()}, {"input": ("node", "1x0e")})      # reported once: a later, valid batch is clean
    G.check_indices()
