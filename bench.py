#!/usr/bin/env python3
"""bench.py — molecules/s of the config_energy training step on MI355X (the BASELINE.json metric).

A step = one pass of the hot path over one synthetic QM9-like batch, what the reference's ``Trainer.batch_step``
does (``e3_layers/run/trainer.py:358-399``): forward of the SequentialGraphNetwork, loss ``1e3 * MSE(total_energy)``
(``e3_layers/configs/config_energy.py:27``), backward, (gradient all-reduce when N > 1), Adam step, EMA update
(``config_energy.py:18-20``: use_ema).  Workload at every N: BASELINE.json configs[1] — config_energy, l_max=2,
n_dim 64, 5 layers, 256 molecules per GPU (weak scaling: graph-parallel data parallelism).

Every step sees a NEW batch object: four distinct batches are resident in HBM and each step works on a fresh device
copy of the next one's tensors, so everything the framework derives per batch (CSR topology by destination and by
source, tile ownership, species key groups, one-hot indices) is rebuilt inside the timed step, as in training, where
a batch is never seen twice.  ``per_batch_prep_ms`` reports that part on its own.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line carrying the metric plus
  roofline     — the fused TP+reduce forward kernel (e3k::tp_fwd_kernel): algorithmic bytes (SURVEY.md §8d variant A)
                 / HIP-event time of its launches inside the timed region, vs 8 TB/s; ``kernels`` lists the other
                 edge kernels and the radial GEMM the same way (the step's weakest kernel is in there, not hidden);
  cpu_baseline — the oracle (unfused PyTorch restatement, kind "port") on the host cores at BASELINE configs[0]
                 (config_energy as shipped: l_max 3, 32 molecules): warm-up + median, forward and forward+backward
                 (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "equivariant-nn-zoo_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 at 2.4 GHz (the clock an MFMA loop sustains is lower)
# rocprofv3 --pmc passes over this command at the two l_max values (tools/collect_profiles_r04.sh, tools/summarize_profiles_r04.py)
TRAFFIC_FILES = {2: "profiles/r06_tp_traffic.json", 3: "profiles/r06_lmax3_tp_traffic.json"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="energy", choices=["energy", "energy_force", "diffusion", "diffusion_CA"],
                    help="BASELINE.json configuration: energy = configs[1] (the metric's workload, default); energy_force = "
                         "configs[2] (64 molecules, energy + force loss: double backward); diffusion = configs[3] (128 fully "
                         "connected molecules, VP-SDE loss); diffusion_CA = configs[4] (4 x 384 residues)")
    ap.add_argument("--batch", type=int, default=None, help="graphs per GPU (default: 256 / 64 / 128 / 4 by --config)")
    ap.add_argument("--lmax", type=int, default=2)
    ap.add_argument("--bonds", default="uniform", choices=["uniform", "clustered"],
                    help="synthetic geometry of the energy / energy_force batches: neighbour distances U(1.0, 1.55) A (SURVEY 8d, default) or "
                         "element-pair bond lengths +- 0.01 A with tetrahedral angles (the clustered distances of real molecules)")
    ap.add_argument("--loader", action="store_true",
                    help="feed the step from the prefetching loader (data/loader.py: collate of fresh samples on a worker thread, "
                         "pinned buffers, async H2D) instead of HBM-resident batches; reported beside the resident figure")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph-fresh", action="store_true",
                    help="HIP-graph replay with a NEW batch every step: the resident batches are padded to one size bucket with a "
                         "ghost graph (run/graph_step.py), each step copies the next one into the captured tensors and replays")
    ap.add_argument("--graph", action="store_true",
                    help="capture the whole step (fwd+loss+bwd+all-reduce+Adam) on ONE resident batch in a HIP graph and "
                         "replay it (no per-batch work in the replayed step: disclosed in config.launch)")
    ap.add_argument("--launch", default="fixed", choices=["fixed", "auto"],
                    help="fixed (default): ONE launch mode per workload, decided by the workload alone -- the HIP-graph replay of padded "
                         "fresh batches for config_energy (the metric's workload: since round 6 -- the replay does not depend on the host's "
                         "launch rate, the eager step on this pool's shared boxes does), config_energy_force and config_diffusion, the eager "
                         "multi-stream step for config_diffusion_CA (data-dependent edge counts); --eager pins the eager step; "
                         "auto: time the eager layouts (and the replay when the host is the limit) on this box and keep the fastest "
                         "(rounds 3-4's default; reported under config.launch_auto)")
    ap.add_argument("--eager", action="store_true", help="the eager multi-stream step (rounds 1-5's default for config_energy)")
    ap.add_argument("--cpu-budget", type=float, default=30.0, help="seconds of CPU work for the baseline leg")
    return ap.parse_args()


def _cpu_model_times(tree, batch, target, reps_fwd, reps_bwd, budget_s):
    """(median forward seconds, median forward+backward seconds, reps actually timed) of the oracle network."""
    from oracle import e3ref

    torch.manual_seed(0)
    net = e3ref.build(tree).float()

    def fwd():
        t0 = time.perf_counter()
        with torch.no_grad():
            net(dict(batch.data), dict(batch.attrs))
        return time.perf_counter() - t0

    def fwd_bwd():
        t0 = time.perf_counter()
        out, _ = net(dict(batch.data), dict(batch.attrs))
        loss = 1e3 * torch.nn.functional.mse_loss(out["total_energy"], target)
        net.zero_grad(set_to_none=True)
        loss.backward()
        return time.perf_counter() - t0

    t_start = time.perf_counter()
    for _ in range(3):                                       # 3 warm-ups (thread pool, allocator)
        fwd()
    f = [fwd() for _ in range(reps_fwd)]
    if reps_bwd == 0:
        return statistics.median(f), None, (len(f), 0)
    warm = fwd_bwd()                                         # warm-up
    left = budget_s - (time.perf_counter() - t_start)
    n = max(3, min(reps_bwd, int(left / max(warm, 1e-3))))   # bounded (the default run must finish within minutes), median of >= 3
    b = [fwd_bwd() for _ in range(n)]
    return statistics.median(f), statistics.median(b), (len(f), len(b))


def cpu_baseline(budget_s):
    """the oracle in fp32 on the host cores.  Primary figure: BASELINE configs[0] (config_energy as
    shipped, l_max 3, 32 molecules), forward and forward+backward, warm-up + median.  Second field: forward of the
    bench's own model (l_max 2) on 32 molecules, the denominator of the ">= 15x CPU forward" north-star clause."""
    from e3_layers_amd.configs import config_energy
    from e3_layers_amd.data.synthetic import synth_qm9

    host_cores = os.cpu_count() or 1
    cores = min(host_cores, 32)  # more threads than this only adds fork/join overhead on these op sizes
    torch.set_num_threads(cores)
    n_mol = 32
    batch = synth_qm9(0, n_mol, config_energy.QM9_SHIFTS)
    target = batch["total_energy"]
    f3, b3, (nf3, nb3) = _cpu_model_times(config_energy.get_config(l_max=3).model_config, batch, target, 10, 10, 0.75 * budget_s)
    f2, _, (nf2, _) = _cpu_model_times(config_energy.get_config(l_max=2).model_config, batch, target, 10, 0, 0.25 * budget_s)
    return {
        "value": round(n_mol / b3, 4), "unit": "molecules/s", "cores": cores, "host_cpu_count": host_cores, "kind": "port",
        "sample": (f"oracle/e3ref.py fp32 on {cores} threads (os.cpu_count() = {host_cores}), BASELINE configs[0]: config_energy l_max 3, synth_qm9(seed 0, "
                   f"{n_mol} molecules): forward 3 warm-ups + median of {nf3} ({f3:.2f} s: BASELINE.md section 3's protocol); fwd+bwd 1 warm-up + "
                   f"median of {nb3} ({b3:.2f} s/step -- fewer repetitions than the protocol's 10: ten more steps of {b3:.0f} s each would "
                   f"take the default run past its few-minutes budget); second field: forward of the l_max 2 model on the same molecules, "
                   f"3 warm-ups + median of {nf2}"),
        "forward_only_value": round(n_mol / f3, 4),
        "lmax2_forward_only_value": round(n_mol / f2, 4),
    }


def _knots_now():
    from e3_layers_amd.backend import radial_table

    return [int(radial_table.KNOTS), int(radial_table.KNOTS_SLOPE)]


def launch_ranks(args) -> int:
    """``python bench.py --gpus N`` as given (no torchrun around it): this process touches no GPU -- it starts N FRESH rank
    processes through ``python -m torch.distributed.run`` (one per GPU, rendezvous on 127.0.0.1, the reference's
    ``torch.multiprocessing.spawn`` of ``train.py:272,280-304``), relays their output, and prints rank 0's JSON line ONCE, with
    the CPU baseline of the same run (measured here, after the ranks have finished, so that it does not disturb their hosts'
    launch rates) added to it.  A rank that fails makes the elastic agent end its siblings; the agent's exit code is ours."""
    import subprocess

    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["E3K_BENCH_CHILD"] = "1"
    # rendezvous: the agent's own store on a port the OS hands out (endpoint port 0) -- nothing is picked here by binding and
    # closing a socket, which another job on a shared box could grab in between
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--rdzv-backend=c10d",
           "--rdzv-endpoint=127.0.0.1:0", "--local-addr", "127.0.0.1", os.path.abspath(__file__)] + [a for a in sys.argv[1:]]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True, bufsize=1)
    line = None
    try:
        for out in proc.stdout:
            text = out.rstrip("\n")
            if text.startswith("{") and '"metric"' in text:
                line = text                      # rank 0's result: held back until the baseline has been added
            else:
                print(text, flush=True)
        rc = proc.wait()
    except BaseException:
        proc.kill()                              # (exactly the process started here; the agent takes its ranks with it)
        proc.wait()
        raise
    if rc != 0 or line is None:
        if line is not None:
            print(line, flush=True)
        print(f"bench.py: the {args.gpus}-rank run failed (exit code {rc}" + ("" if line is not None else ", no result line") + ")",
              file=sys.stderr, flush=True)
        return rc if rc != 0 else 1
    result = json.loads(line)
    if not args.no_cpu_baseline and args.config == "energy" and os.environ.get("E3K_BENCH_DRY_RUN") is None:
        result["cpu_baseline"] = cpu_baseline(args.cpu_budget)
    result.setdefault("config", {})["launcher"] = f"bench.py started {args.gpus} rank processes itself (torch.distributed.run, c10d rendezvous on 127.0.0.1)"
    print(json.dumps(result), flush=True)
    return 0


def dry_run(args, world: int, rank: int) -> None:
    """E3K_BENCH_DRY_RUN: the launcher's plumbing without a GPU (tests/test_parallel_gloo.py) -- the ranks rendezvous over gloo,
    all-reduce one number and rank 0 prints a stub line; ``fail-rankK`` makes rank K exit with code 3 first."""
    mode = os.environ["E3K_BENCH_DRY_RUN"]
    if mode == f"fail-rank{rank}":
        raise SystemExit(3)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t)
        assert float(t) == world * (world + 1) / 2
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "dry-run", "value": 0.0, "unit": "molecules/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "config": {"workload": "launcher dry run (no GPU work)"}}), flush=True)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as `python bench.py --gpus N`: become the launcher BEFORE anything touches a GPU (fresh child processes only --
        # a process that has initialised HIP must never be re-executed)
        raise SystemExit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus != world:
        raise SystemExit(f"bench.py --gpus {args.gpus} was launched with WORLD_SIZE={world}: start it as `python bench.py --gpus "
                         f"{args.gpus}` (it launches its ranks itself) or under `python -m torch.distributed.run --nproc-per-node "
                         f"{args.gpus} bench.py --gpus {args.gpus} ...`")
    if os.environ.get("E3K_BENCH_DRY_RUN") is not None:
        return dry_run(args, world, rank)
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
        assert dist.get_world_size() == args.gpus

    from e3_layers_amd.backend import ops
    from e3_layers_amd.backend.graph import build_topology
    from e3_layers_amd.configs import config_diffusion, config_diffusion_CA, config_energy, config_energy_force
    from e3_layers_amd.data.synthetic import synth_protein, synth_qm9, synth_qm9_diffusion
    from e3_layers_amd.run.optim import FusedAdamEMA
    from e3_layers_amd.run.parallel import backward_parameters, broadcast_parameters, flat_param_order, param_names
    from e3_layers_amd.run.sde_utils import VPSDE, sde_loss, sde_loss_of, sde_perturb
    from e3_layers_amd.utils import build, countParameters

    # ---- the workload: BASELINE.json configs[1] by default; configs[2..4] by --config (tools/config_bench.py folded in) ----
    cfg_kind = args.config
    if args.batch is None:
        args.batch = {"energy": 256, "energy_force": 64, "diffusion": 128, "diffusion_CA": 4}[cfg_kind]
    n_res = 4
    unit = "proteins" if cfg_kind == "diffusion_CA" else "molecules"
    if cfg_kind == "energy":
        cfg = config_energy.get_config(l_max=args.lmax)
        make = lambda k: synth_qm9(1000 + 17 * k + rank, args.batch, config_energy.QM9_SHIFTS, bonds=args.bonds)
        opt_kw = dict(ema_decay=cfg.ema_decay if cfg.use_ema else None, ema_use_num_updates=cfg.ema_use_num_updates)
    elif cfg_kind == "energy_force":
        cfg = config_energy_force.get_config()
        make = lambda k: synth_qm9(2000 + 17 * k + rank, args.batch, config_energy_force.SHIFTS, r_max=5.0, bonds=args.bonds)
        opt_kw = {}
    elif cfg_kind == "diffusion":
        cfg = config_diffusion.get_config()
        make = lambda k: synth_qm9_diffusion(1 + 17 * k + rank, args.batch)
        opt_kw = dict(max_grad_norm=1.0)
    else:
        cfg = config_diffusion_CA.get_config()
        make = lambda k: synth_protein(1 + 17 * k + rank, args.batch, n_res=384)
        opt_kw = dict(max_grad_norm=1.0)
    tree = cfg.model_config
    torch.manual_seed(0)
    model = build(tree).to(dev)
    broadcast_parameters(model)
    # parameters, gradients, Adam moments and the EMA shadow as flat vectors: one all-reduce, one fused optimizer launch
    order = flat_param_order(model)
    opt = FusedAdamEMA(order, lr=cfg.learning_rate, names=param_names(model, order), **opt_kw)
    flat = opt.grads
    flat.enable_direct_accumulation()
    # launch mode (decided by the workload alone): see the --launch help
    if (args.launch == "fixed" and not args.eager and not (args.loader or args.graph)
            and (cfg_kind == "energy" or (cfg_kind in ("energy_force", "diffusion") and world == 1))):
        # the fixed mode of these workloads (no run-time choice in the default line; VERDICT r5 item 1: the
        # metric's workload too -- its eager step is host-bound on the driver's boxes (4.5-5.4 ms, regions 15 % apart), the replay
        # is 4.2 ms whatever the host does).  Several ranks (config_energy): forward + backward replayed, ONE flat all-reduce and
        # the optimizer launch behind it.
        args.graph_fresh = True
    if world > 1 and os.environ.get("E3K_OVERLAP_ALLREDUCE", "1") != "0" and not args.graph_fresh:
        flat.enable_overlapped_all_reduce(model)     # a layer's gradient slice is all-reduced while the backward goes on

    # every rank owns its own graphs (weak scaling): four distinct seeded batches per rank, resident in HBM
    host_batches = [make(k) for k in range(n_res)]
    resident = [b.to(dev) for b in host_batches]
    n_nodes = [b["pos"].shape[0] if "pos" in b else b["_n_nodes"].sum().item() for b in resident]
    n_edges = [b["edge_index"].shape[1] if "edge_index" in b else 0 for b in resident]
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    if cfg_kind == "energy_force":
        for b in resident:      # synthetic force targets (the generator has none): one fixed draw per resident batch
            b["forces_target"] = torch.randn(b["pos"].shape, device=dev, generator=gen)
            b.attrs["forces_target"] = ("node", "1x1o")
    sde = VPSDE({"pos": 3}) if cfg_kind == "diffusion" else (VPSDE({"CA": 3}) if cfg_kind == "diffusion_CA" else None)

    def loss_of(batch):
        if cfg_kind == "energy":
            target = batch["total_energy"]          # the model writes its prediction under the same key of the same Batch
            return ops.sq_error(model(batch)["total_energy"], target, None, 1e3)      # (= 1e3 * mse_loss: loss + gradient in one launch)
        if cfg_kind == "energy_force":                  # config_energy_force.py:18 loss_coeffs
            # (the model writes its graph energy under "energy", config_energy_force.py:73; the synthetic target travels as
            #  "total_energy" -- until round 5 this line compared the target with itself: the energy term was identically zero)
            e_t, f_t = batch["total_energy"], batch["forces_target"]
            out = model(batch)
            return ops.sq_error(out["energy"], e_t, None, 1e3) + ops.sq_error(out["forces"], f_t, None, 3e4)
        return sde_loss(sde, model, batch, generator=gen)[0]

    # setup, not a step of the workload: libe3k.so is loaded, the TP plans are created and the code objects of every
    # kernel on the path are paged in by one forward/backward over a SMALL batch (no optimizer step, gradients zeroed
    # after) -- otherwise the first timed-or-warm-up step carries 0.2 s of lazy initialisation
    if cfg_kind == "energy":
        tiny = synth_qm9(7, 8, config_energy.QM9_SHIFTS).to(dev)
        loss_of(tiny).backward()
        ops.join_side_streams()
        flat.zero()
        torch.cuda.synchronize()
        del tiny

    counter = [0]
    loader = None
    if args.loader:
        from e3_layers_amd.data.loader import PrefetchLoader, samples_of

        pool = [s for b in host_batches for s in samples_of(b)]
        loader = iter(PrefetchLoader(pool, batch_size=args.batch, device=dev, shuffle=True, seed=rank, epochs=None))

    def next_batch():
        """A batch the framework has never seen: fresh device copies of the next resident batch's tensors (what a
        collated batch arriving from the loader is), so no per-batch memo (topology, key groups) can hit; with
        --loader: the next batch collated by the loader's worker from individual samples."""
        if loader is not None:
            return next(loader)
        b = resident[counter[0] % n_res].clone()
        counter[0] += 1
        return b

    def step(batch=None):
        batch = next_batch() if batch is None else batch
        loss = loss_of(batch)
        flat.zero()
        if cfg_kind == "energy":
            loss.backward(gradient=ops.unit_gradient(loss))      # (a persistent 1.0: no ones_like fill, no multiplication by it)
        else:
            backward_parameters(loss, opt.params)
        flat.all_reduce_mean()
        opt.step()
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run = step
    # config_diffusion_CA rebuilds its edge list inside the model from the noised coordinates and reads the edge count back: in the
    # plain loop that read-back waits behind the previous step's whole backward.  Software-pipelined loop: the NEXT batch is noised
    # and its leading data-only layers run (SequentialGraphNetwork.prepare) between this batch's forward and backward -- the same
    # kernels per step on the same stream, the read-back only waits for the forward.  E3K_BENCH_PIPELINE=0: the plain loop.
    pipelined = (cfg_kind == "diffusion_CA" and hasattr(model, "prepare") and os.environ.get("E3K_BENCH_PIPELINE", "1") != "0"
                 and not args.loader)
    if pipelined:
        pending = [None]

        def noised_and_prepared():
            pert, misc = sde_perturb(sde, next_batch(), generator=gen)
            model.prepare(pert)
            return pert, misc

        def step_pipelined():
            if pending[0] is None:
                pending[0] = noised_and_prepared()
            pert, misc = pending[0]
            loss = sde_loss_of(sde, model, pert, misc)[0]
            pending[0] = noised_and_prepared()
            flat.zero()
            backward_parameters(loss, opt.params)
            flat.all_reduce_mean()
            opt.step()
            return loss

        run = step_pipelined
    graph = None
    if args.graph:
        if cfg_kind != "energy" or args.loader:
            raise SystemExit("--graph replays the config_energy step on one resident batch")
        from e3_layers_amd.run.graph_step import CapturedStep

        fixed = resident[0]
        fixed.update(build_topology(fixed["edge_index"], fixed["pos"].shape[0]).as_dict())
        captured = CapturedStep(lambda: step(fixed.view()), warmup=3)
        graph = captured.graph
        run = captured

    bucket = None
    n_cap = e_cap = 0
    auto = None

    def make_bucket():
        from e3_layers_amd.run.graph_step import BucketedStep, PipelinedBucketedStep, bucket_capacity, pad_batch

        # the bucket: here the capacity of the resident batches (a training run takes it from the dataset's statistics and
        # steps a batch that does not fit eagerly)
        n_cap, e_cap = bucket_capacity(list(zip(n_nodes, n_edges)))
        padded = [pad_batch(b, n_cap, e_cap).to(dev) for b in host_batches]
        if cfg_kind == "energy_force":
            for p in padded:
                p["forces_target"] = torch.randn(p["pos"].shape, device=dev, generator=gen)
                p.attrs["forces_target"] = ("node", "1x1o")

        def finish():
            flat.all_reduce_mean()
            opt.step()

        def gradients_of(batch):
            if cfg_kind == "diffusion":       # VP-SDE denoising loss: a mean over the REAL nodes
                loss = sde_loss(sde, model, batch, generator=gen, node_weight=batch["_node_weight"])[0]
                flat.zero()
                backward_parameters(loss, opt.params)
                return loss
            target, weight = batch["total_energy"], batch["_graph_weight"]      # weight: 1 / G for the real graphs, 0 for the ghost
            if cfg_kind == "energy":
                loss = ops.sq_error(model(batch)["total_energy"], target, weight, 1e3)
                flat.zero()
                loss.backward(gradient=ops.unit_gradient(loss))
            else:      # config_energy_force.py:18 loss_coeffs; the force term is a mean over the REAL nodes' components
                f_t, wn = batch["forces_target"], batch["_node_weight"]
                out = model(batch)
                loss = ops.sq_error(out["energy"], target, weight, 1e3) + ops.sq_error(out["forces"], f_t, wn, 3e4 / 3.0)
                flat.zero()
                backward_parameters(loss, opt.params)
            return loss

        gens = (gen,) if cfg_kind == "diffusion" else ()
        if world == 1:      # the whole step is the graph

            def train_on(batch):
                loss = gradients_of(batch)
                finish()
                return loss

            if pipelined_prep:
                bucket_ = PipelinedBucketedStep(prep_fn, train_on, padded[0], warmup=3, generators=gens)
            else:
                bucket_ = BucketedStep(train_on, padded[0], warmup=3, generators=gens)
        else:
            # several ranks: forward + backward are the graph, the flat all-reduce (the same fixed sequence of slices as in the
            # eager step, issued in one go) and the fused optimizer launch follow it eagerly -- RCCL stays outside the capture
            def backward_on(batch):
                loss = gradients_of(batch)
                ops.join_side_streams()      # the sunk weight gradients land inside the capture
                return loss

            flat.early_start = False
            if pipelined_prep:
                bucket_ = PipelinedBucketedStep(prep_fn, backward_on, padded[0], warmup=3, generators=gens, tail=finish)
            else:
                bucket_ = BucketedStep(backward_on, padded[0], warmup=3, generators=gens, tail=finish)

        def run_():
            b = padded[counter[0] % n_res]
            counter[0] += 1
            if pipelined_prep:      # ... and the NEXT batch is handed to the preparation stream (what a prefetching loader does)
                return bucket_(b, nxt=padded[counter[0] % n_res])
            return bucket_(b)

        return bucket_, run_, n_cap, e_cap

    replay_error = None
    # batch preparation (device copy, CSR views, edge geometry, species groups, knot bins, edge records) as its own graph on a second
    # stream beside the previous step (run/graph_step.PipelinedBucketedStep); E3K_BENCH_PREP_PIPELINE=0: inside the step's graph
    pipelined_prep = bool(args.graph_fresh and hasattr(model, "prepare_data") and os.environ.get("E3K_BENCH_PREP_PIPELINE", "1") != "0")
    # (the denoising loss noises ``pos`` on a clone of the batch: nothing that reads the positions may be computed ahead of it)
    prep_fn = (lambda b: model.prepare_data(b, exclude=("pos",))) if cfg_kind == "diffusion" else getattr(model, "prepare_data", None)
    if args.graph_fresh:
        if cfg_kind not in ("energy", "energy_force", "diffusion") or args.loader or args.graph:
            raise SystemExit("--graph-fresh replays the config_energy / config_energy_force / config_diffusion step on padded resident "
                             "batches (the protein net rebuilds its edge list inside the model: data-dependent sizes)")
        if world == 1:
            bucket, run, n_cap, e_cap = make_bucket()
        else:
            # several ranks: a rank whose capture fails must not leave the others inside a collective -- the capture and its warm-up
            # issue none (BucketedStep), the ranks agree on the outcome here, and all of them fall back to the eager step together
            made = None
            try:
                made = make_bucket()
            except Exception as ex:
                replay_error = f"{type(ex).__name__}: {ex}"[:200]
            failed = torch.tensor([0.0 if made is not None else 1.0], device=dev)
            dist.all_reduce(failed, op=dist.ReduceOp.MAX)
            if float(failed.item()) > 0.0:
                replay_error = replay_error or "another rank could not capture"
                flat.zero()
            else:
                bucket, run, n_cap, e_cap = made
        if bucket is not None:
            graph = bucket.captured.graph

    def max_over_ranks(seconds: float) -> float:
        t = torch.tensor([seconds], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    replicas = []

    def check_replicas(tag: str) -> None:
        """Several ranks: every rank's parameters (and EMA shadow) must be bit-identical after a step -- the ranks start from rank
        0's weights and apply the same all-reduced gradient.  A diverged collective schedule, a rank that skipped a step or a
        reduction that paired the wrong buffers shows up here on the first multi-GPU run, not as a slowly diverging loss."""
        if world == 1:
            return
        torch.cuda.synchronize()
        words = opt.flat.view(torch.int32).to(torch.int64)
        cs = torch.stack([words.sum(), (words * (torch.arange(words.numel(), device=dev) % 8191 + 1)).sum()])
        if opt.ema is not None:
            cs = torch.cat([cs, opt.ema.view(torch.int32).to(torch.int64).sum().view(1)])
        every = [torch.empty_like(cs) for _ in range(world)]
        dist.all_gather(every, cs)
        same = all(bool((c == every[0]).all()) for c in every)
        replicas.append({"after": tag, "equal": same})
        if not same:
            raise SystemExit(f"bench.py: the ranks' parameters differ {tag} (checksums {[c.tolist() for c in every]}): the "
                             "data-parallel step is broken (collective schedule / reduction)")

    # W untimed warm-up steps; the last few are clocked only as a reference rate, to recognise a timed region that an
    # external stall (another tenant of the box, a clock dip) stretched several-fold
    n_first = 1 if (world > 1 and args.warmup >= 1) else 0
    if n_first:
        run()
        check_replicas("after step 1")
    n_ref = min(args.warmup - n_first, 3)
    for _ in range(args.warmup - n_first - n_ref):
        run()
    fence()
    waited_ref = opt.waited_seconds
    t0 = time.perf_counter()
    for _ in range(n_ref):
        run()
    host_ref = (time.perf_counter() - t0) - (opt.waited_seconds - waited_ref)
    fence()
    ref_step = max_over_ranks(time.perf_counter() - t0) / n_ref if n_ref else None

    # Launch mode.  The eager multi-stream step is the fastest one as long as the host keeps ahead of the GPU (3.9-4.4 ms of
    # Python + launches per 5.0-5.4 ms step at 256 molecules on an idle host); on a loaded host -- the boxes of this pool are
    # shared -- it becomes host-bound (7 ms seen).  The HIP-graph replay with a new padded batch every step
    # (run/graph_step.py) does not depend on the host at all and costs 5.5 ms there.  So: when the reference steps were bound
    # by the host, capture the bucketed step, time it, and keep whichever is faster for the timed region.  One rank; the
    # config_energy, config_energy_force and config_diffusion workloads (the ones the bucketed replay serves);
    # E3K_BENCH_AUTO=0 pins the eager step.
    if (args.launch == "auto" and cfg_kind in ("energy", "energy_force", "diffusion") and not (args.loader or args.graph or args.graph_fresh)
            and n_ref and os.environ.get("E3K_BENCH_AUTO", "1") != "0"):
        # (several ranks: only the choice between the two eager layouts -- they issue the same collectives, so the ranks cannot
        #  diverge; the times compared are the maxima over the ranks, so every rank takes the same decision)
        auto = {"eager_ms_per_step": round(1e3 * ref_step, 3), "eager_host_busy_ms_per_step": round(1e3 * host_ref / n_ref, 3),
                "chosen": "eager"}
        # the eager step on ONE stream: as fast as the four-stream layout when the host has no slack (256 molecules: 5.24 vs
        # 5.23 ms on a loaded host, 5.26 vs 5.00 on an idle one) with a third less host work
        from e3_layers_amd.nn import message_passing as _mp

        forked = (_mp.FORK_MIN_EDGES, _mp.FORK_MIN_EDGES_TABLE)
        _mp.FORK_MIN_EDGES = _mp.FORK_MIN_EDGES_TABLE = 10 ** 12
        for _ in range(2):
            run()
        fence()
        waited_ref = opt.waited_seconds
        t0 = time.perf_counter()
        for _ in range(n_ref):
            run()
        host_one = (time.perf_counter() - t0) - (opt.waited_seconds - waited_ref)
        fence()
        one_step = max_over_ranks(time.perf_counter() - t0) / n_ref
        auto["eager_one_stream_ms_per_step"] = round(1e3 * one_step, 3)
        auto["eager_one_stream_host_busy_ms_per_step"] = round(1e3 * host_one / n_ref, 3)
        if one_step < 0.98 * ref_step:
            ref_step, host_ref = one_step, host_one
            auto["chosen"] = "eager, one stream"
        else:
            _mp.FORK_MIN_EDGES, _mp.FORK_MIN_EDGES_TABLE = forked
        # (several ranks: every quantity a rank decides on is a maximum over the ranks, the capture itself issues no collective,
        #  and the ranks agree on its outcome before the first replayed step: they cannot take different branches)
        # (several ranks: the replay has only been exercised with two ranks on one GPU over gloo, never over RCCL on a multi-GPU
        #  box -- there it is tried on request only: E3K_BENCH_AUTO=try-graph, or --graph-fresh to pin it)
        host_bound = world == 1 and host_ref / n_ref >= 0.85 * ref_step
        # (force training issues ~360 launches from Python per step: the replay is tried whatever the host's share was in the few
        #  reference steps -- on a shared box that share swings between 0.7 and 0.95 from run to run)
        if host_bound or (world == 1 and cfg_kind == "energy_force") or os.environ.get("E3K_BENCH_AUTO") == "try-graph":
            made = None
            try:
                made = make_bucket()
            except Exception as ex:      # (the eager step stays: the capture is an optimisation of the measurement, not part of it)
                auto["graph_fresh_error"] = f"{type(ex).__name__}: {ex}"[:200]
            if max_over_ranks(0.0 if made is not None else 1.0) > 0.0:
                made = None
                auto.setdefault("graph_fresh_error", "another rank could not capture")
            if made is not None:
                bucket_c, run_c, n_cap_c, e_cap_c = made
                try:
                    for _ in range(2):
                        run_c()
                    fence()
                    t0 = time.perf_counter()
                    for _ in range(5):
                        run_c()
                    fence()
                    graph_step = max_over_ranks(time.perf_counter() - t0) / 5
                    auto["graph_fresh_ms_per_step"] = round(1e3 * graph_step, 3)
                    if graph_step < 0.95 * ref_step:
                        bucket, run, graph, ref_step = bucket_c, run_c, bucket_c.captured.graph, graph_step
                        n_cap, e_cap = n_cap_c, e_cap_c
                        auto["chosen"] = "graph-fresh"
                except Exception as ex:
                    if world > 1:
                        raise      # the ranks are past the point where they could agree to go back
                    auto["graph_fresh_error"] = f"{type(ex).__name__}: {ex}"[:200]
            if bucket is None:
                flat.early_start = True

    from e3_layers_amd.backend import conv_native, radial_table

    class _Ms:
        """An elapsed time read back from the native layer executor's event pairs, in the shape of a torch event pair."""

        def __init__(self, ms):
            self.ms = ms

        def elapsed_time(self, other):
            return other.ms

    def native_records(recs):
        """Launches issued by csrc/e3k_layer.hip (HIP event pairs recorded there, on the launching stream)."""
        for nl in conv_native._LAYERS:
            plan = nl.plan
            for kind in ("tp_fwd", "tp_bwd_x", "tp_bwd_w"):
                for ms, n, e in nl.profile_read(kind):
                    recs.setdefault(kind, []).append((_Ms(0.0), _Ms(ms), (n, e, plan.tp_plan)))
            for kind in ("rtable_fwd", "rtable_bwd"):
                for ms, rows, e in nl.profile_read(kind):
                    recs.setdefault(kind, []).append((_Ms(0.0), _Ms(ms), (e, rows - 1, plan.last_spec.d_out)))
            for ms, rows, e in nl.profile_read("radial_last_fwd"):
                recs.setdefault("radial_last_fwd", []).append((_Ms(0.0), _Ms(ms), (rows, plan.last_spec.d_in, plan.last_spec.d_out)))
            nl.profile(0)
        return recs

    host_enqueue = [0.0]

    # The timed region records HIP event pairs around the DOMINANT kernel only (`tp_fwd`: the `roofline` block's
    # achieved figure is measured live, on the launching stream); event pairs around every other launch of interest cost the
    # eager step 3 % (4.85 vs 4.69 ms: a region without them, `ms_per_step_repeats`) -- the secondary kernels' durations
    # come from a short eager pass behind the timed regions, as they do for a replayed graph.
    DOMINANT = ("tp_fwd",)

    def timed_region():
        ops.PROFILE = {} if graph is None else None
        ops.PROFILE_ONLY = set(DOMINANT)
        if graph is None:
            for nl in conv_native._LAYERS:
                nl.profile(args.steps + 8, kinds=DOMINANT)
        fence()
        waited0 = opt.waited_seconds
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = run()
        # the host's share: everything enqueued, minus what it spent waiting for the GPU to come within two steps
        host_enqueue[0] = (time.perf_counter() - t0) - (opt.waited_seconds - waited0)
        fence()
        seconds = time.perf_counter() - t0
        recs, ops.PROFILE = ops.PROFILE, None
        ops.PROFILE_ONLY = None
        if graph is None:
            recs = native_records(recs)
        return max_over_ranks(seconds), recs, out

    elapsed, records, loss = timed_region()
    check_replicas(f"after the timed region's step {args.steps}")
    retimed = None
    if ref_step is not None and elapsed / args.steps > 3.0 * ref_step:
        # exactly K steps are timed again, once; the line reports the repeat and the discarded figure
        retimed = {"first_ms_per_step": round(1e3 * elapsed / args.steps, 3), "warmup_ms_per_step": round(1e3 * ref_step, 3)}
        elapsed, records, loss = timed_region()
    # the K-step region four more times (0.1 s is a short window on a shared box): reported as extra keys, `value` stays the first
    # region's (the contract: exactly K timed steps)
    repeats = [elapsed / args.steps]
    for _ in range(4):
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run()
        fence()
        repeats.append(max_over_ranks(time.perf_counter() - t0) / args.steps)
    # eager pass for the roofline block: every kernel kind when the step was a replayed graph (per-kernel events cannot be
    # read back from a replay), the secondary kinds otherwise (the dominant kernel keeps the timed region's records)
    live = {k: v for k, v in (records or {}).items() if k in DOMINANT} if graph is None else {}
    ops.PROFILE = {}
    for nl in conv_native._LAYERS:
        nl.profile(16)
    from e3_layers_amd.nn import message_passing as _mp_

    forked_ = (_mp_.FORK_MIN_EDGES, _mp_.FORK_MIN_EDGES_TABLE)
    if graph is not None:
        # A replayed step is ONE in-order queue.  Event pairs cannot bracket a kernel inside a graph on this runtime (the HIP 7.0
        # runtime torch ships rejects hipEventRecordWithFlags(hipEventRecordExternal) under a capture -- tools/micro/ext_event_torch.py;
        # ROCm 7.2's own accepts it: tools/micro/ext_event.hip), so the kernels' durations are taken from eager steps laid out like
        # the replay: one stream, every kernel alone on the chip.  profiles/r06_bench_kernel_stats.csv (rocprofv3 over this very
        # command: it does see the replayed kernels) carries the in-graph durations to compare with.
        _mp_.FORK_MIN_EDGES = _mp_.FORK_MIN_EDGES_TABLE = 10 ** 12
    for _ in range(min(args.steps, 5) + (2 if graph is not None else 0)):
        step()
    torch.cuda.synchronize()
    _mp_.FORK_MIN_EDGES, _mp_.FORK_MIN_EDGES_TABLE = forked_
    records, ops.PROFILE = ops.PROFILE, None
    records = native_records(records)
    records.update(live)

    # ---- outside the timed region: the pieces of a step on their own (rank 0 reports them) -------------------------
    def event_ms(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n

    def forward_only():
        with torch.no_grad():
            model(next_batch())

    def prep_only():
        b = next_batch()
        build_topology(b["edge_index"], b["pos"].shape[0])

    fwd_ms = prep_ms = None
    if cfg_kind == "energy":      # (the other configurations' forward needs the loss wrapper's inputs: t, noise)
        forward_only()
        fwd_ms = event_ms(forward_only, 10)
        prep_only()
        prep_ms = event_ms(prep_only, 10)

    if loader is not None:
        loader.close()

    # ---- roofline blocks (rank 0's launches in the timed region) ---------------------------------------------------
    table_rows = radial_table.layout(float(getattr(tree, "r_max", 4.0)), radial_table.KNOTS)[0] + 1      # rows of a layer's knot table

    def in_kernel(e, plan):
        """This launch interpolated its path weights from the knot table inside the kernel (csrc/e3k_tp.hip TABLE forms)."""
        return bool(conv_native.TP_TABLE and radial_table.ENABLED and getattr(plan, "_e3k_table_form", False)
                    and e >= radial_table.MIN_EDGES_PER_KNOT * table_rows)

    def fused_bwd(e, plan):
        """This input-gradient launch also wrote the per-edge weight gradient (csrc/e3k_tp.hip MODE 5: packed-table layers)."""
        return bool(in_kernel(e, plan) and conv_native.TP_TABLE_PACKED and conv_native.TP_BWD_FUSED and conv_native.ENABLED)

    def tp_bytes(kind, n, e, plan, variant="A"):
        """Algorithmic bytes of one launch.  Variant A = the module-API operation
        tp(x[src], sh, w) + scatter with the per-edge weights supplied (E x W streamed); variant B = the radial-fused
        operation (weights produced in the kernel: here interpolated from the (K + 1) x W knot table, counted once)."""
        w_term = 4 * plan.w_numel
        extra = 0
        if variant == "B" and kind != "tp_bwd_w":
            # knot (4 B) + four interpolation weights (16 B) per edge; the table once -- 12-byte records when the kernels read it packed
            w_term, extra = 20, (12 if conv_native.TP_TABLE_PACKED else 4) * table_rows * plan.w_numel
        if kind == "tp_fwd":        # x[src] gather + sh + w + out rows
            return e * (4 * plan.d_in + 4 * plan.d_sh + w_term + 16) + n * 4 * plan.d_mid + extra
        if kind == "tp_bwd_w":      # x[src] gather + sh + g_w stream + g_mid rows
            return e * (4 * plan.d_in + 4 * plan.d_sh + 4 * plan.w_numel + 16) + n * 4 * plan.d_mid
        if kind == "tp_bwd_x":      # compulsory: w + sh + g_mid once + g_x rows (the g_mid[dst] gather is not counted)
            fused = e * 4 * plan.w_numel + n * 4 * plan.d_in if (variant == "B" and fused_bwd(e, plan)) else 0      # + g_w [E, W] out, x rows in
            return e * (4 * plan.d_sh + w_term + 16) + n * 4 * (plan.d_mid + plan.d_in) + extra + fused
        raise KeyError(kind)

    def summarise(kind):
        recs = (records or {}).get(kind, [])
        tot_b = sum(tp_bytes(kind, *meta) for _, _, meta in recs)
        tot_ms = sum(ev0.elapsed_time(ev1) for ev0, ev1, _ in recs)
        n = max(len(recs), 1)
        ach = tot_b / (tot_ms * 1e-3) / 1e9 if tot_ms > 0 else 0.0
        out = {"kernel": f"e3k::{kind}_kernel", "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": round(ach / HBM_PEAK_GBS, 4), "launches": len(recs), "avg_launch_us": round(1e3 * tot_ms / n, 2),
               "avg_launch_algorithmic_MB": round(tot_b / n / 1e6, 2)}
        n_tab = sum(1 for _, _, (nn, e, plan) in recs if in_kernel(e, plan))
        if n_tab and kind != "tp_bwd_w" and tot_ms > 0:
            # These launches never read w[E, W] from HBM: they gather four rows of the L2-resident knot table per edge.  A kernel
            # that produces its own weights is SURVEY 8d's variant B: `achieved` / `frac` count the bytes this form must move
            # (x[src], sh, knot + four weights per edge, the output rows, the table once); variant A (the module-API operation's
            # bytes, E x W streamed: what rounds 1-3 led with) stays as a secondary key for comparison with the earlier rounds.
            tot_bb = sum(tp_bytes(kind, *meta, variant="B" if in_kernel(meta[1], meta[2]) else "A") for _, _, meta in recs)
            ach_b = tot_bb / (tot_ms * 1e-3) / 1e9
            out.update({"variant_A_achieved": out["achieved"], "variant_A_frac": out["frac"],
                        "variant_A_avg_launch_algorithmic_MB": out["avg_launch_algorithmic_MB"]})
            out.update({"achieved": round(ach_b, 1), "frac": round(ach_b / HBM_PEAK_GBS, 4),
                        "avg_launch_algorithmic_MB": round(tot_bb / n / 1e6, 2), "launches_with_in_kernel_table": n_tab,
                        "bytes_model": "SURVEY 8d variant B (weights produced in the kernel from the knot table)",
                        "note": "in-kernel knot-table form: E x W is not streamed from HBM (L2 / Infinity-Cache gathers of a "
                                f"<= {(12 if conv_native.TP_TABLE_PACKED else 4) * table_rows * max(m[2].w_numel for _, _, m in recs) / 1e6:.1f} MB "
                                + ("packed table (one 12-byte record per knot and weight) per layer" if conv_native.TP_TABLE_PACKED else "table per layer")
                                + "); frac = the bytes "
                                "this form must move / time / 8 TB/s; variant_A_frac = the module-API operation's bytes (E x W streamed) "
                                "/ time, as reported in rounds 1-3"})
        return out

    traffic = {}
    TRAFFIC_FILE = TRAFFIC_FILES.get(args.lmax, "")
    tfile = os.path.join(ROOT, TRAFFIC_FILE)
    if TRAFFIC_FILE and os.path.exists(tfile) and cfg_kind == "energy" and args.batch == 256:
        try:
            traffic = json.load(open(tfile))
        except Exception:
            traffic = {}
    main_k = summarise("tp_fwd")
    roofline = {k: main_k[k] for k in ("bound", "achieved", "peak", "unit", "frac")}
    t_fwd = traffic.get("tp_fwd", {}).get("traffic_bytes_per_launch")
    roofline["traffic"] = round(t_fwd) if t_fwd else None
    roofline["traffic_source"] = (f"{TRAFFIC_FILE} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this command, "
                                  "NOT this run; the counters sit on the L2's fabric side: Infinity-Cache hits -- the knot-table "
                                  "rows and the gathered node rows that miss an XCD's L2 -- are counted with the HBM bytes, so this "
                                  "is an upper bound on DRAM traffic)") if t_fwd else None
    if t_fwd and main_k["avg_launch_us"] > 0:   # DRAM-side rate: counter bytes / this run's launch time / peak
        roofline["dram_frac"] = round(t_fwd / (main_k["avg_launch_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
    roofline.update({k: main_k[k] for k in main_k if k not in roofline})
    kernels = []
    for kind in ("tp_bwd_x", "tp_bwd_w"):
        k = summarise(kind)
        if kind == "tp_bwd_w" and not k["launches"]:
            continue      # (every layer formed its weight gradient inside the input-gradient walk)
        if kind == "tp_bwd_x":
            n_f = sum(1 for _, _, (nn, e, plan) in (records or {}).get(kind, []) if fused_bwd(e, plan))
            if n_f:
                k["launches_with_fused_weight_gradient"] = n_f
                k["kernel"] += " (MODE 5: the per-edge weight gradient g_w [E, W] written in the same walk -- its bytes are in the model)"
        t_k = traffic.get(kind, {}).get("traffic_bytes_per_launch")
        k["traffic"] = round(t_k) if t_k else None
        if t_k and k["avg_launch_us"] > 0:
            k["dram_frac"] = round(t_k / (k["avg_launch_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
        kernels.append(k)
    for kind, label in (("rtable_fwd", "interpolation of the radial knot table, forward: E x W written + the table"),
                        ("rtable_bwd", "its transpose: E x W read + per-knot partial sums written and combined")):
        recs = (records or {}).get(kind, [])
        if not recs:
            continue
        nbytes = sum((4.0 * e * w + 4.0 * (k + 1) * w * (1 if kind == "rtable_fwd" else 7)) for _, _, (e, k, w) in recs)
        ms = sum(ev0.elapsed_time(ev1) for ev0, ev1, _ in recs)
        ach = nbytes / (ms * 1e-3) / 1e9
        kernels.append({"kernel": f"e3k::{kind.replace('rtable_', 'rtable_interp_')}_kernel ({label})", "bound": "hbm",
                        "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                        "launches": len(recs), "avg_launch_us": round(1e3 * ms / len(recs), 2),
                        "avg_launch_algorithmic_MB": round(nbytes / len(recs) / 1e6, 2)})
    recs = (records or {}).get("radial_last_fwd", [])
    if recs:
        flops = sum(2.0 * e * k * w for _, _, (e, k, w) in recs)
        ms = sum(ev0.elapsed_time(ev1) for ev0, ev1, _ in recs)
        rows = sorted({e for _, _, (e, k, w) in recs})
        kernels.append({"kernel": "e3k::gemm_kernel<2, false> (radial MLP last layer, forward; rows per launch: "
                                  + ("the knot table" if max(rows) < 10000 else "one per edge") + ")", "bound": "mfma",
                        "achieved": round(flops / (ms * 1e-3) / 1e12, 1), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                        "frac": round(flops / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, 4), "launches": len(recs),
                        "avg_launch_us": round(1e3 * ms / len(recs), 2)})
    roofline["kernels"] = kernels

    if rank == 0:
        workloads = {
            "energy": f"config_energy QM9-like, l_max={args.lmax}, n_dim 64, 5 layers, {args.batch} molecules per GPU",
            "energy_force": f"config_energy_force (BASELINE configs[2]): n_dim 64, l_max 2, r_max 5, {args.batch} molecules per GPU, "
                            "loss on energies and forces (double backward through every kernel)",
            "diffusion": f"config_diffusion (BASELINE configs[3]): VP-SDE score net, n_dim 32, 4 layers, {args.batch} fully "
                         "connected molecules per GPU",
            "diffusion_CA": f"config_diffusion_CA (BASELINE configs[4]): residue-level score net, n_dim 64, 8 layers, {args.batch} x 384 "
                            "residues per GPU",
        }
        result = {
            "metric": ("molecules/s forward+backward, QM9 config_energy batch" if cfg_kind == "energy"
                       else f"{unit}/s forward+backward, {cfg_kind} batch (not the BASELINE metric's workload)"),
            "value": round(world * args.batch * args.steps / elapsed, 2),
            "unit": f"{unit}/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "ms_per_step_median": round(1e3 * statistics.median(repeats), 3),
            "ms_per_step_repeats": {"regions": len(repeats), "min": round(1e3 * min(repeats), 3),
                                    "median": round(1e3 * statistics.median(repeats), 3), "max": round(1e3 * max(repeats), 3),
                                    "note": "the K-step region timed five times back to back; value / ms_per_step are the first region's"},
            "host_busy_ms_per_step": round(1e3 * host_enqueue[0] / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": workloads[cfg_kind] + (f" (rank 0: {n_res} resident batches, a fresh copy per step, N={min(n_nodes)}-{max(n_nodes)} "
                                               f"nodes, E={min(n_edges)}-{max(n_edges)} edges), fwd + loss + bwd + Adam"
                                               + (" + EMA" if cfg_kind == "energy" else "")),
                "geometry": args.bonds,
                "input": "prefetching loader (collate of fresh samples + pinned H2D inside the loop)" if args.loader else "HBM-resident batches",
                "global_batch": world * args.batch, "parallelism": f"graph-parallel dp{world}",
                "ranks": dist.get_world_size() if world > 1 else 1,
                "replica_parameter_checksums": replicas if world > 1 else None,
                "all_reduce": (None if world == 1 else "one flat all-reduce (sum, / world) behind the replayed forward + backward" if bucket is not None
                               else "layer slices of the flat gradient, overlapped with the backward (static schedule)"),
                "table": ("radial MLP on a cubic knot table (2^-7 A spacing), read packed: d0, d1 fp32 + (d2, d3) as an fp16 pair per (knot, "
                          f"weight), a-posteriori guard {radial_table.GUARD_TOL:g} table-wide / {radial_table.GUARD_TOL_COL:g} per column "
                          "evaluated on the device with every replay; arithmetic fp32"),
                "replay_error": replay_error,
                "knot_table_recaptures": (getattr(bucket, "recaptures", None) if bucket is not None else None),      # the table guard's
                "knot_table_knots": _knots_now(),      # refinements (the knot count doubles) and vetoes made the step record itself again
                "streams": ("one (captured)" if graph is not None else
                            "one" if (auto or {}).get("chosen") == "eager, one stream" else "per size: four from 60 000 edges (table layers)"),
                "launch": ((f"hip-graph replay, a NEW batch every step: padded to the bucket ({n_cap} nodes, {e_cap} edges) with a ghost "
                            "graph of zero loss weight, copied into the captured tensors, CSR build / species groups / knot bins "
                            + ("in a second graph on a preparation stream, beside the previous batch's step (two buffers)" if pipelined_prep
                               else "inside the graph")
                            + ("; forward + backward replayed, the flat all-reduce and the optimizer launch follow eagerly" if world > 1 else ""))
                           if bucket is not None else
                           "hip-graph replay of ONE resident batch (no per-batch work in the replayed step)" if graph is not None else "eager"),
                "launch_auto": auto,
                "loop": ("software-pipelined: the next batch's edge list (one count read back) is built between this batch's forward and backward"
                         if pipelined else "plain"),
                "parameters": countParameters(model), "final_loss": round(float(loss.detach()), 4),
                "retimed_after_stall": retimed,
            },
            "roofline": roofline,
        }
        if fwd_ms is not None:
            result.update({"per_batch_prep_ms": round(prep_ms, 3), "gpu_forward_only_ms": round(fwd_ms, 3),
                           "gpu_forward_only_molecules_per_s": round(args.batch / (fwd_ms * 1e-3), 1)})
        if world == 1 and not args.no_cpu_baseline and cfg_kind == "energy":
            cb = cpu_baseline(args.cpu_budget)
            result["cpu_baseline"] = cb
            result["gpu_vs_cpu_forward"] = round(result["gpu_forward_only_molecules_per_s"] / cb["lmax2_forward_only_value"], 1)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
