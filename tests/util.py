"""Shared helpers for the parity tests (product HIP path vs the float64 oracle)."""
import torch

from oracle import e3ref


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """Normwise relative error ||a-b|| / ||b|| in float64."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    denom = float(b.norm())
    return float((a - b).norm()) / (denom if denom > 0 else 1.0)


def to_cf(x: torch.Tensor, irreps) -> torch.Tensor:
    """[mul][2l+1] blocks -> [2l+1][mul] blocks (pure torch, for checking the relayout kernel)."""
    cols, pos = [], 0
    for mul, l, _ in e3ref.parse_irreps(str(irreps)):
        d = 2 * l + 1
        cols.append(x[:, pos:pos + mul * d].reshape(-1, mul, d).transpose(1, 2).reshape(-1, mul * d))
        pos += mul * d
    return torch.cat(cols, dim=1)


def from_cf(x: torch.Tensor, irreps) -> torch.Tensor:
    cols, pos = [], 0
    for mul, l, _ in e3ref.parse_irreps(str(irreps)):
        d = 2 * l + 1
        cols.append(x[:, pos:pos + mul * d].reshape(-1, d, mul).transpose(1, 2).reshape(-1, mul * d))
        pos += mul * d
    return torch.cat(cols, dim=1)


def oracle_like(product_model, config_tree, dtype=torch.float64):
    """Oracle network built from the same config tree, carrying the product's parameters."""
    orc = e3ref.build(config_tree)
    sd = {("mods." + k if not k.startswith("func.") else k): v.detach().cpu() for k, v in product_model.state_dict().items()}
    orc.load_state_dict(_rename(sd, orc))
    return orc.to(dtype)


def _rename(sd, orc):
    want = set(orc.state_dict().keys())
    out = {}
    for k, v in sd.items():
        if k in want:
            out[k] = v
        elif k.startswith("mods.func."):
            out["func.mods." + k[len("mods.func."):]] = v
        else:
            out[k] = v
    return out


def zero_shifts(*models) -> None:
    """Zeroes the per-species energy shifts of ``PerTypeScaleShift`` layers.  The QM9 shifts are about -1e4 eV per molecule: with them
    in, a NORMWISE relative bound of 1e-5 on ``total_energy`` allows 0.1 eV of absolute error on a learned part of O(1) (VERDICT r5,
    weak item 3).  Tests that compare energies with the oracle zero them in both models first: the bound is then on the network."""
    for model in models:
        for m in model.modules():
            s = getattr(m, "shifts", None)
            if isinstance(s, torch.Tensor):
                with torch.no_grad():
                    s.zero_()


def batch_to_oracle(batch, dtype=torch.float64):
    data = {k: (v.detach().cpu().to(dtype) if v.is_floating_point() else v.detach().cpu())
            for k, v in batch.data.items() if not k.startswith("_e3k_")}
    return data, dict(batch.attrs)


def record_measured(test: str, **values) -> None:
    """Appends measured errors to the JSON-lines file named by E3K_PARITY_LOG (DESIGN.md quotes them); no-op otherwise."""
    import json
    import os

    path = os.environ.get("E3K_PARITY_LOG")
    if path:
        with open(path, "a") as f:
            f.write(json.dumps({"test": test, **{k: (float(v) if isinstance(v, (int, float)) else v) for k, v in values.items()}}) + "\n")
