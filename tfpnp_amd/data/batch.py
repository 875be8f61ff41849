"""Minimal stand-in for tfpnp/data/batch.py (a 700-line tianshou-style container) covering what the env <-> policy
path uses: keyword construction, attribute / key access, row indexing, `.shape[0]`, `len()`, `.to(device)`, and the
two combinators the replay-memory side of a trainer needs (`Batch.stack`, trainer/mddpg/trainer.py:236-241; `Batch.cat`)
for batches that share their keys (the reference's zero-padding of missing keys is not reproduced: a missing key raises)."""
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

    @staticmethod
    def _combine(batches, fn, what):
        batches = [b if isinstance(b, Batch) else Batch(b) for b in batches]
        if not batches:
            raise ValueError(f"Batch.{what} of an empty sequence")
        keys = list(batches[0].keys())
        for b in batches[1:]:
            if set(b.keys()) != set(keys):
                raise KeyError(f"Batch.{what}: batches do not share their keys ({sorted(keys)} vs {sorted(b.keys())})")
        out = {}
        for k in keys:
            vals = [b[k] for b in batches]
            if isinstance(vals[0], Batch):
                out[k] = Batch._combine(vals, fn, what)
            elif isinstance(vals[0], torch.Tensor):
                out[k] = fn(vals)
            else:
                out[k] = fn([torch.as_tensor(v) for v in vals])
        return Batch(out)

    @staticmethod
    def stack(batches, axis=0):
        """New leading (or `axis`) dimension over a list of batches: what convert2batch does with sampled transitions."""
        return Batch._combine(batches, lambda v: torch.stack(v, axis), "stack")

    @staticmethod
    def cat(batches):
        """Concatenation along dimension 0."""
        return Batch._combine(batches, lambda v: torch.cat(v, 0), "cat")

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
