#!/usr/bin/env python3
"""bench.py — molecules/s of the config_energy training step on MI355X (the BASELINE.json metric).

A step = one pass of the hot path over one synthetic QM9-like batch, exactly what the reference's
``Trainer.batch_step`` does (``e3_layers/run/trainer.py:358-399``): forward of the
SequentialGraphNetwork, loss ``1e3 * MSE(total_energy)`` (``e3_layers/configs/config_energy.py:27``),
backward, (gradient all-reduce when N > 1), Adam step.  Inputs are resident in HBM before the
timed region.  Workload at every N: BASELINE.json configs[1] — config_energy, l_max=2, n_dim 64,
5 layers, 256 molecules per GPU (weak scaling: graph-parallel data parallelism).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line carrying the metric plus
  roofline     — fused TP+reduce forward kernel (e3k::tp_fwd_kernel): algorithmic bytes (SURVEY.md §8d
                 variant A) / HIP-event time of its launches inside the timed region, vs 8 TB/s;
  cpu_baseline — the oracle (unfused PyTorch restatement, kind "port") timed on the host cores
                 on a bounded 32-molecule sample of the same workload (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "equivariant-nn-zoo_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="molecules per GPU")
    ap.add_argument("--lmax", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="capture the whole step (fwd+loss+bwd+all-reduce+Adam) in one HIP graph and replay it; the "
                         "roofline block then comes from an eager pass after the timed region")
    ap.add_argument("--cpu-sample", type=int, default=96, help="molecules in the CPU-baseline sample")
    return ap.parse_args()


def cpu_baseline(tree, shifts, n_mol, budget_s=20.0):
    """Oracle fwd+bwd on the host cores, fp32.  One timed step on a bounded sample: a 4-molecule
    probe step is timed first and the sample is sized so that the timed step takes about
    ``budget_s`` seconds (at most ``n_mol`` molecules)."""
    from e3_layers_amd.data.synthetic import synth_qm9
    from oracle import e3ref

    cores = min(os.cpu_count() or 1, 32)  # more threads than this only adds fork/join overhead on these op sizes
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    net = e3ref.build(tree).float()

    def step(batch):
        data = {k: v for k, v in batch.data.items()}
        t0 = time.perf_counter()
        out, _ = net(data, dict(batch.attrs))
        loss = 1e3 * torch.nn.functional.mse_loss(out["total_energy"], batch["total_energy"])
        t1 = time.perf_counter()
        net.zero_grad(set_to_none=True)
        loss.backward()
        return t1 - t0, time.perf_counter() - t0

    step(synth_qm9(1, 2, shifts))  # lazy-init warm-up, untimed
    _, probe = step(synth_qm9(2, 4, shifts))
    n_mol = int(max(4, min(n_mol, 4 * budget_s / max(probe, 1e-3))))
    fwd, total = step(synth_qm9(0, n_mol, shifts))
    return {
        "value": round(n_mol / total, 4),
        "unit": "molecules/s",
        "cores": cores,
        "kind": "port",
        "sample": (f"oracle/e3ref.py fp32 on {cores} threads, 1 fwd+bwd step on synth_qm9(seed 0, {n_mol} molecules), "
                   f"same model; {total:.1f} s total, forward-only {n_mol / fwd:.3f} molecules/s"),
        "forward_only_value": round(n_mol / fwd, 4),
    }


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # one process per GPU; E3K_DIST_BACKEND=gloo lets two ranks share one GPU to smoke-test the N>1 path
    backend = os.environ.get("E3K_DIST_BACKEND", "nccl")
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    if world > torch.cuda.device_count():
        # ranks share a GPU (smoke-testing the N>1 path on one device): several processes x several HIP streams on one
        # device time-slice pathologically (measured 18x), so keep each process on one stream there
        os.environ.setdefault("E3K_FWD_FORK", "0")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if args.gpus != world and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)

    from e3_layers_amd.backend import ops
    from e3_layers_amd.backend.graph import build_topology
    from e3_layers_amd.configs import config_energy
    from e3_layers_amd.data.synthetic import synth_qm9
    from e3_layers_amd.run.optim import FusedAdamEMA
    from e3_layers_amd.run.parallel import broadcast_parameters
    from e3_layers_amd.utils import build, countParameters

    cfg = config_energy.get_config(l_max=args.lmax)
    tree = cfg.model_config
    torch.manual_seed(0)
    model = build(tree).to(dev)
    broadcast_parameters(model)
    # parameters, gradients and Adam moments as flat vectors: one all-reduce, one fused optimizer launch
    opt = FusedAdamEMA(model.parameters(), lr=cfg.learning_rate)
    flat = opt.grads
    flat.enable_direct_accumulation()

    # every rank owns its own 256 molecules (weak scaling); seeded per rank, resident in HBM
    batch = synth_qm9(1000 + rank, args.batch, config_energy.QM9_SHIFTS).to(dev)
    batch.update(build_topology(batch["edge_index"], batch["pos"].shape[0]).as_dict())
    target = batch["total_energy"]
    n_nodes, n_edges = batch["pos"].shape[0], batch["edge_index"].shape[1]

    # setup, not a step of the workload: libe3k.so is loaded, the TP plans are created and the code objects of every
    # kernel on the path are paged in by one forward/backward over EIGHT molecules (no optimizer step, gradients zeroed
    # after) -- otherwise the first timed-or-warm-up step carries 0.2 s of lazy initialisation
    tiny = synth_qm9(7, 8, config_energy.QM9_SHIFTS).to(dev)
    (1e3 * torch.nn.functional.mse_loss(model(tiny)["total_energy"], tiny["total_energy"])).backward()
    ops.join_side_streams()
    flat.zero()
    torch.cuda.synchronize()
    del tiny

    def step():
        out = model(batch.view())   # fresh key dict over the resident tensors (the model adds keys, never mutates inputs)
        loss = 1e3 * torch.nn.functional.mse_loss(out["total_energy"], target)
        flat.zero()
        loss.backward()
        flat.all_reduce_mean()
        opt.step()
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run = step
    graph = None
    if args.graph:
        from e3_layers_amd.run.graph_step import CapturedStep

        captured = CapturedStep(step, warmup=3)
        graph = captured.graph
        run = captured

    def max_over_ranks(seconds: float) -> float:
        t = torch.tensor([seconds], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # W untimed warm-up steps; the last few are clocked only as a reference rate, to recognise a timed region that an
    # external stall (another tenant of the box, a clock dip) stretched several-fold
    n_ref = min(args.warmup, 3)
    for _ in range(args.warmup - n_ref):
        run()
    fence()
    t0 = time.perf_counter()
    for _ in range(n_ref):
        run()
    fence()
    ref_step = max_over_ranks(time.perf_counter() - t0) / n_ref if n_ref else None

    def timed_region():
        ops.PROFILE_TP = [] if graph is None else None
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = run()
        fence()
        seconds = time.perf_counter() - t0
        recs, ops.PROFILE_TP = ops.PROFILE_TP, None
        return max_over_ranks(seconds), recs, out

    elapsed, records, loss = timed_region()
    first_elapsed = None
    if ref_step is not None and elapsed / args.steps > 3.0 * ref_step:
        # exactly K steps are timed again, once; the line reports the repeat and says so (config.retimed_after_stall)
        first_elapsed = elapsed
        elapsed, records, loss = timed_region()
    if graph is not None:  # per-kernel events cannot be read back from a replayed graph: eager pass for the roofline block
        ops.PROFILE_TP = []
        for _ in range(min(args.steps, 5)):
            step()
        torch.cuda.synchronize()
        records, ops.PROFILE_TP = ops.PROFILE_TP, None

    # roofline of the fused TP+reduce forward kernel (rank 0's launches in the timed region)
    tot_bytes, tot_ms = 0.0, 0.0
    for start, end, n, e, plan in records:
        tot_ms += start.elapsed_time(end)
        tot_bytes += e * (4 * plan.d_in + 4 * plan.d_sh + 4 * plan.w_numel + 16) + n * 4 * plan.d_mid
    n_launch = max(len(records), 1)
    achieved = tot_bytes / (tot_ms * 1e-3) / 1e9 if tot_ms > 0 else 0.0
    # HBM bytes per launch from the PMC counters cannot be read in-process; they come from the committed
    # rocprofv3 --pmc passes over this same command (profiles/, tools/collect_profiles.sh) when the
    # workload matches, else null
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "r01_tp_fwd_traffic.json")
    if os.path.exists(tfile) and args.batch == 256 and args.lmax == 2:
        try:
            traffic = round(json.load(open(tfile))["traffic_bytes_per_launch"])
        except Exception:
            traffic = None
    roofline = {
        "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
        "kernel": "e3k::tp_fwd_kernel", "launches": len(records),
        "avg_launch_us": round(1e3 * tot_ms / n_launch, 2), "avg_launch_algorithmic_MB": round(tot_bytes / n_launch / 1e6, 2),
    }

    if rank == 0:
        result = {
            "metric": "molecules/s forward+backward, QM9 config_energy batch",
            "value": round(world * args.batch * args.steps / elapsed, 2),
            "unit": "molecules/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"config_energy QM9-like, l_max={args.lmax}, n_dim 64, 5 layers, {args.batch} molecules per GPU "
                            f"(rank 0: N={n_nodes} nodes, E={n_edges} edges), fwd + 1e3*MSE + bwd + Adam",
                "global_batch": world * args.batch, "parallelism": f"graph-parallel dp{world}", "launch": "hip-graph replay" if graph is not None else "eager",
                "parameters": countParameters(model), "final_loss": round(float(loss.detach()), 4),
            },
            "roofline": roofline,
        }
        if first_elapsed is not None:
            result["config"]["retimed_after_stall"] = {"first_ms_per_step": round(1e3 * first_elapsed / args.steps, 3),
                                                       "warmup_ms_per_step": round(1e3 * ref_step, 3)}
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(tree, config_energy.QM9_SHIFTS, args.cpu_sample)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
