"""The N > 1 path with the REAL model on the GPU: two ranks (gloo rendezvous; both on cuda:0, single-stream convolutions:
several processes x several HIP streams on one device time-slice pathologically) run the config_energy network on their
shard of a batch through exactly what bench.py drives over RCCL -- FlatGradients with direct accumulation (the gradient
sink, side-stream weight gradients), all_reduce_mean, FusedAdamEMA -- and must reproduce the single-process gradient
and parameter update of the union batch.  The 8-GPU RCCL run itself is the driver's (SCALE_rNN.json); this pins the
sink / stream-join / flat-buffer logic under a process group.  Reference: train.py:99,272, run/trainer.py:138-139."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
ROOT = sys.argv[1]
for p in (ROOT, os.path.join(ROOT, "equivariant-nn-zoo_amd")):
    sys.path.insert(0, p)
import torch
import torch.distributed as dist

rank, world, out = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), sys.argv[2]
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
if world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)
from e3_layers_amd.backend import ops
from e3_layers_amd.configs.layer_configs import addEnergyOutput, featureModel
from e3_layers_amd.data.synthetic import synth_qm9
from e3_layers_amd.run.optim import FusedAdamEMA
from e3_layers_amd.run.parallel import broadcast_parameters, shard_batch
from e3_layers_amd.utils import build

tree = addEnergyOutput(featureModel(n_dim=64, l_max=2, edge_spherical="1x0e+1x1o+1x2e", node_attrs="20x0e", edge_radial="8x0e",
                                    num_types=10, num_layers=3, r_max=4.0), None)
torch.manual_seed(100 + rank)               # different initial weights per rank ...
model = build(tree).to(dev)
broadcast_parameters(model)                 # ... made identical, as DDP does at construction
opt = FusedAdamEMA(model.parameters(), lr=1e-2, ema_decay=0.99)
flat = opt.grads
flat.enable_direct_accumulation()
if len(sys.argv) > 3 and sys.argv[3] == "overlap":
    flat.enable_overlapped_all_reduce()
start = opt.flat.clone()
batch = synth_qm9(5, 40)      # > 256 nodes per rank: the keyed self-connection, so the layers run as fused blocks
mine = shard_batch(batch, rank, world).to(dev)
target = mine["total_energy"]
n_mine = len(mine)
res = model(mine)
# per-rank SUM of squared errors scaled by world / total graphs: the all-reduce MEAN of these is the global mean loss
loss = (res["total_energy"] - target).square().sum() * (world / len(batch))
flat.zero()
loss.backward()
flat.all_reduce_mean()
grad = flat.gather().clone()
opt.step()
torch.cuda.synchronize()
torch.save({"overlapped": flat.overlapped_slices, "grad": grad.cpu(), "start": start.cpu(), "after": opt.flat.detach().cpu().clone(), "n": n_mine,
            "sink_entries": len(ops.GRAD_SINK)}, out)
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(world, tmp_path, mode="plain"):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs, outs = [], []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   E3K_FWD_FORK="0" if world > 1 else os.environ.get("E3K_FWD_FORK", "1"), HSA_ENABLE_IPC_MODE_LEGACY="0")
        out = tmp_path / f"w{world}_{mode}_r{rank}.pt"
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(out), mode], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            log, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(log.decode(errors="replace"))
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log[-3000:]
    return [torch.load(o) for o in outs]


@pytest.mark.parametrize("mode", ["plain", "overlap"])
def test_two_ranks_on_one_gpu_match_the_single_process_step(dev, tmp_path, mode):
    """mode "overlap": every fused layer's slice of the flat gradient is all-reduced on a communication stream as soon as
    its weight-gradient kernels are enqueued (FlatGradients.enable_overlapped_all_reduce), the rest at the end."""
    two = _launch(2, tmp_path, mode)
    one = _launch(1, tmp_path)[0]
    r0, r1 = two
    assert r0["overlapped"] == r1["overlapped"] == (3 if mode == "overlap" else 0)      # the three convolution layers
    assert r0["n"] + r1["n"] == 40 and r0["n"] >= 1 and r1["n"] >= 1
    assert r0["sink_entries"] > 20                              # the weight-gradient kernels wrote into the flat buffer
    assert torch.equal(r0["start"], r1["start"])                # broadcast made the replicas identical
    assert torch.equal(r0["grad"], r1["grad"])                  # one all-reduce: both ranks hold the same mean gradient
    assert torch.equal(r0["after"], r1["after"])
    # the single process started from ITS seed: compare through the gradient of the same parameters instead -- rank 0's
    # parameters equal the single process's (both seed 100), so gradients and updates must agree to rounding
    assert torch.equal(r0["start"], one["start"])
    denom = float(one["grad"].norm())
    assert denom > 0
    assert float((r0["grad"] - one["grad"]).norm()) / denom < 2e-5
    assert float((r0["after"] - one["after"]).norm()) / float(one["after"].norm()) < 1e-6
