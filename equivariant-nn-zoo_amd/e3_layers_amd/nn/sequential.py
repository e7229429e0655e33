"""Layer protocol and container.

``Module.init_irreps`` and ``SequentialGraphNetwork`` follow the interface of
``e3_layers/nn/sequential.py:12-39,42-88``: a layer is constructed with keyword irreps given
either as ``irreps`` or as ``(irreps, custom_key)``; its ``forward(data, attrs)`` receives the
batch dict with custom keys renamed to the layer's canonical names and returns only the new
entries, which the container renames back and merges.  Plain callables ``(data, attrs) ->
(data, attrs)`` are accepted as layers.  TorchScript (``config['jit']``) is accepted and ignored:
the arithmetic is in HIP kernels.
"""
from __future__ import annotations

from collections import OrderedDict

import torch
from torch.profiler import record_function

from ..data import Batch
from ..o3 import Irreps
from ..utils.utils import _is_mapping, build, keyMap


class Module(torch.nn.Module):
    def init_irreps(self, output_keys=(), **kwargs):
        if isinstance(output_keys, str):
            output_keys = [output_keys]
        self.irreps_in, self.irreps_out = {}, {}
        self.input_key_mapping, self.output_key_mapping = {}, {}
        for key, value in kwargs.items():
            if value is None:
                continue
            if isinstance(value, (str, Irreps)):
                irreps, custom = value, key
            elif isinstance(value, (list, tuple)) and len(value) == 2:
                irreps, custom = value
            else:
                raise TypeError(f"{key}: expected irreps or (irreps, key), got {value!r}")
            if key in output_keys:
                self.irreps_out[key] = irreps
                self.output_key_mapping[key] = custom
            else:
                self.irreps_in[key] = irreps
                self.input_key_mapping[custom] = key

    def inputKeyMap(self, x):
        return keyMap(x, self.input_key_mapping)

    def outputKeyMap(self, x):
        return keyMap(x, self.output_key_mapping)


class SequentialGraphNetwork(torch.nn.Sequential):
    def __init__(self, **config):
        self.layers = []
        self.layer_configs = config["layers"]
        modules = OrderedDict()
        for key, value in self.layer_configs:
            if _is_mapping(value):
                module = build(value)
                modules[key] = module
                self.layers.append((key, module))
            elif callable(value):
                self.layers.append((key, value))
            else:
                raise TypeError(f"invalid config node for layer {key!r}")
        super().__init__(modules)
        # radial look-ahead (nn/message_passing.py): every convolution knows the next one whose radial MLP reads the
        # same edge embedding, so that MLP can be issued one layer early on the side stream
        prev = None
        for _, layer in self.layers:
            conv = getattr(layer, "conv", None)
            if conv is None or not hasattr(conv, "fc") or not getattr(conv, "reduce", False):
                continue
            src = next((g for g, loc in getattr(layer, "input_key_mapping", {}).items() if loc == "edge_radial"), None)
            if prev is not None and prev[1] == src and src is not None:
                object.__setattr__(prev[0], "_next_conv", conv)     # a plain reference, not a registered submodule
                prev[2].__dict__["_next_mp"] = layer                # (the fused block's look-ahead wants the layer)
            prev = (conv, src, layer)

        # cf hand-over between directly consecutive MessagePassing layers on one feature key (nn/message_passing.py):
        # layer i may emit the channel-fastest layout when layer i+1 is the only reader of that key and overwrites it
        for (_, a), (_, b) in zip(self.layers[:-1], self.layers[1:]):
            ok = getattr(a, "cf_chain_ok", None)
            if ok is None or not isinstance(b, Module) or not ok(b):
                continue
            key = a.output_key_mapping.get("output_features")
            reads = [g for g, loc in b.input_key_mapping.items() if loc == "input_features"]
            other_reads = [g for g, loc in b.input_key_mapping.items() if loc != "input_features"]
            if key is not None and reads == [key] and key not in other_reads and b.output_key_mapping.get("output_features") == key:
                a._emit_cf = True

    def prepare(self, batch) -> int:
        """Runs the LEADING layers that are plain callables -- parameter-free data preparation by construction, e.g. the
        protein nets' ``computeEdgeIndex``, which reads the edge count back from the device -- on ``batch`` now and marks it, so
        that ``forward(batch)`` starts behind them (same object; a clone forgets the mark and simply runs them again).  A training
        loop calls this for the NEXT batch before it enqueues the current batch's backward: the read-back then waits for the
        forward alone instead of for a whole step queued in front of it.  Same stream, same kernels, only earlier."""
        data, attrs = batch.data, batch.attrs
        done = 0       # (a second prepare() on the same object starts over: whatever the first one derived may be stale by now)
        for _, layer in self.layers[done:]:
            if isinstance(layer, torch.nn.Module):
                break
            self._run_layer(layer, data, attrs)
            done += 1
        batch._e3k_prepared = done
        return done

    # ---- batch preparation ahead of the step (round 6) -----------------------------------------------------------------
    # Everything a forward derives from the batch ALONE -- CSR topology, edge vectors, spherical harmonics, one-hot species, the
    # species groups of the keyed self-connection, the knot bins of the radial table and the edge records the packed tensor-product
    # kernels walk -- is about 25 small dependent launches (0.2 ms of a 4.2 ms replayed step at 256 molecules) in front of the first
    # convolution.  In the reference that work is the data loader's (``computeEdgeIndex`` in ``data.preprocess``,
    # ``e3_layers/configs/config_energy.py:49``; ``Batch.from_data_list``'s segment loops): it runs in worker processes beside the
    # previous step.  ``prepare_data(batch)`` is that split: it runs the parameter-free layers whose inputs are batch keys (or
    # outputs of such layers) and the per-batch builds of the layers that declare a ``prepare_batch`` hook, leaves their results
    # in the batch and records what they were computed FROM; ``forward`` skips a recorded layer only while its inputs are still
    # the same tensor objects at the same version (a loss that perturbs ``pos`` before calling the model gets the layer run
    # again).  ``run/graph_step.PipelinedBucketedStep`` replays it as its own HIP graph on a second stream while the previous
    # batch's step runs.
    @staticmethod
    def _data_only_inputs(layer):
        """Global batch keys a parameter-free layer reads, or None when the layer cannot be run ahead."""
        if isinstance(layer, Module):
            if not getattr(layer, "data_only", False) or any(True for _ in layer.parameters()):
                return None
            return tuple(layer.input_key_mapping.keys())
        return getattr(layer, "data_only_inputs", None)

    def prepare_data(self, batch, exclude=()) -> dict:
        """``exclude``: keys that will require grad in the forward (``GradientOutput``'s ``x``): nothing that reads them is run ahead."""
        from ..backend.graph import get_topology

        data, attrs = batch.data, batch.attrs
        avail = {k for k, v in data.items() if torch.is_tensor(v) and not v.requires_grad} - set(exclude)
        done = {}
        with torch.no_grad():
            if "edge_index" in data and "pos" in data and data["edge_index"].is_cuda:
                before = dict(data)
                data.update(get_topology(data, data["pos"].shape[0]).as_dict())
                wrote = tuple(k for k, v in data.items() if before.get(k) is not v)
                done["_topology"] = ((("edge_index", data["edge_index"], data["edge_index"]._version),), wrote)
            for name, layer in self.layers:
                ins = self._data_only_inputs(layer)
                if ins is None or not all(k in avail for k in ins):
                    continue
                before = dict(data)
                self._run_layer(layer, data, attrs)
                wrote = tuple(k for k, v in data.items() if before.get(k) is not v)
                avail |= set(wrote)
                done[name] = (tuple((k, data[k], data[k]._version) for k in ins), wrote)
            for name, layer in self.layers:
                hook = getattr(layer, "prepare_batch", None)
                if hook is not None:
                    hook(self, data, avail)
        batch._e3k_done = done
        return done

    @staticmethod
    def _still_valid(record, data) -> bool:
        """``record``: (((input key, tensor, version), ...), (output keys)) of one layer run ahead."""
        return all(data.get(k) is t and t._version == v for k, t, v in record[0])

    @staticmethod
    def _forget(record, data) -> None:
        """The inputs of a layer run ahead were replaced or written to since: what it left in the batch is stale, and a layer that
        finds its output present keeps it (``computeEdgeVector`` does, as the reference's: ``e3_layers/data/computeEdgeVector``)."""
        for k in record[1]:
            data.pop(k, None)

    def forward(self, batch):
        data, attrs = batch.data, batch.attrs
        done = getattr(batch, "_e3k_done", None)
        if done:
            batch._e3k_done = None      # (one forward: see ``prepare`` below)
        # layer names appear in a torch profile when one is running; otherwise the 2 x 14 record_function ops per forward
        # are 0.3 ms of pure host time
        profiling = torch.autograd._profiler_enabled()
        # the mark applies to exactly ONE forward: a sampler / reverse-SDE loop that updates ``pos`` in place and calls the model
        # again on the same Batch object must get its edge list rebuilt, not the stale one
        start = int(getattr(batch, "_e3k_prepared", 0))
        if start:
            batch._e3k_prepared = 0
        if done and "_topology" in done and not self._still_valid(done["_topology"], data):
            self._forget(done["_topology"], data)
        for key, layer in (self.layers[start:] if start else self.layers):
            if done and key in done:
                if self._still_valid(done[key], data):
                    continue            # run ahead by prepare_data() on exactly these input tensors: its outputs are in ``data``
                self._forget(done[key], data)
            if profiling:
                with record_function(key):
                    self._run_layer(layer, data, attrs)
            else:
                self._run_layer(layer, data, attrs)
        return Batch(attrs, **data)

    @staticmethod
    def _run_layer(layer, data, attrs):
        mapped = isinstance(layer, Module)
        d, a = (layer.inputKeyMap(data), layer.inputKeyMap(attrs)) if mapped else (data, attrs)
        d, a = layer(d, a)
        if mapped:
            d, a = layer.outputKeyMap(d), layer.outputKeyMap(a)
        data.update(d)
        attrs.update(a)
