"""Minimal stand-in for ``ml_collections.ConfigDict`` (not installable offline).

Only what the reference's config files and factory use: attribute and item access, ``update``,
``to_dict``, ``keys/items/values/get/pop``, ``in``, iteration — nested dicts are wrapped on
assignment (``e3_layers/configs/layer_configs.py:21-101,150-166``).
"""
from __future__ import annotations

from collections.abc import Mapping


class ConfigDict(Mapping):
    def __init__(self, initial=None, **kwargs):
        object.__setattr__(self, "_fields", {})
        if initial is not None:
            self.update(initial)
        if kwargs:
            self.update(kwargs)

    @staticmethod
    def _wrap(value):
        if isinstance(value, dict):
            return ConfigDict(value)
        return value

    # mapping protocol
    def __getitem__(self, key):
        return self._fields[key]

    def __setitem__(self, key, value):
        self._fields[key] = self._wrap(value)

    def __delitem__(self, key):
        del self._fields[key]

    def __iter__(self):
        return iter(self._fields)

    def __len__(self):
        return len(self._fields)

    def __contains__(self, key):
        return key in self._fields

    # attribute protocol
    def __getattr__(self, name):
        try:
            return self._fields[name]
        except KeyError:
            raise AttributeError(name) from None

    def __setattr__(self, name, value):
        self[name] = value

    def __delattr__(self, name):
        del self._fields[name]

    def get(self, key, default=None):
        return self._fields.get(key, default)

    def pop(self, key, *default):
        return self._fields.pop(key, *default)

    def update(self, *others, **kwargs):
        for other in others:
            items = other.items() if hasattr(other, "items") else other
            for k, v in items:
                if isinstance(v, (dict, ConfigDict)) and isinstance(self._fields.get(k), ConfigDict):
                    self._fields[k].update(v)
                else:
                    self[k] = v
        for k, v in kwargs.items():
            self[k] = v

    def to_dict(self):
        out = {}
        for k, v in self._fields.items():
            out[k] = v.to_dict() if isinstance(v, ConfigDict) else v
        return out

    def copy_and_resolve_references(self):
        return ConfigDict(self.to_dict())

    def __repr__(self):
        return f"ConfigDict({self._fields!r})"
