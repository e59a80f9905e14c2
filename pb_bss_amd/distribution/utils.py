"""Parameter-container plumbing shared by the model dataclasses.

Mirrors pb_bss/distribution/utils.py:118-220 (`_ProbabilisticModel`): nested
to_dict / from_dict and an AttributeError that suggests close field names.
Host-side only; no numerics.
"""
import difflib
from dataclasses import fields, is_dataclass

import numpy as np

from .. import _lib


class _ProbabilisticModel:
    def to_dict(self):
        out = {}
        for f in fields(self):
            v = getattr(self, f.name)
            out[f.name] = v.to_dict() if isinstance(v, _ProbabilisticModel) else v
        return out

    @classmethod
    def from_dict(cls, d):
        names = [f.name for f in fields(cls)]
        assert set(names) == set(d.keys()), (names, list(d.keys()))
        return cls(**d)

    def __getattr__(self, name):
        names = [f.name for f in fields(self)] if is_dataclass(self) else []
        close = difflib.get_close_matches(name, names) or names
        raise AttributeError(
            f'{self.__class__.__name__!r} object has no attribute {name!r}.\n'
            f'Close matches: {close}')


def as_result(x, like_torch):
    """Device tensor -> what the caller works with (torch in, torch out;
    NumPy in, NumPy out, as the reference returns)."""
    if x is None:
        return None
    return x if like_torch else _lib.to_host(x)
