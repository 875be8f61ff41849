"""Minimal stand-in for tfpnp/data/batch.py (a 700-line tianshou-style container) covering what the env <-> policy
path uses: keyword construction, attribute / key access, row indexing, `.shape[0]`, `len()`, `.to(device)`."""
import torch


class Batch:
    def __init__(self, _dict=None, **kwargs):
        if _dict is not None:
            self.__dict__.update(_dict)
        self.__dict__.update(kwargs)

    def keys(self):
        return self.__dict__.keys()

    def items(self):
        return self.__dict__.items()

    def __contains__(self, key):
        return key in self.__dict__

    def __getitem__(self, index):
        if isinstance(index, str):
            return self.__dict__[index]
        return Batch({k: v[index] for k, v in self.__dict__.items()})

    def __setitem__(self, key, value):
        self.__dict__[key] = value

    def update(self, other=None, **kwargs):
        if other is not None:
            self.__dict__.update(other.__dict__ if isinstance(other, Batch) else other)
        self.__dict__.update(kwargs)

    def __len__(self):
        for v in self.__dict__.values():
            return len(v)
        return 0

    @property
    def shape(self):
        return [len(self)]

    def to(self, device):
        return Batch({k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in self.__dict__.items()})

    def __repr__(self):
        return "Batch(" + ", ".join(f"{k}: {tuple(v.shape) if hasattr(v, 'shape') else v}" for k, v in self.items()) + ")"
