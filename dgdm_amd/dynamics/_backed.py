"""Mixin for nn.Modules whose ``forward`` runs on a packed copy of their weights inside libdgdm_hip.so."""
from __future__ import annotations

import torch
import torch.nn as nn


class HipBacked(nn.Module):
    """Keeps the reference's parameter tree (so ``state_dict`` keys match) and a lazily built device handle.

    The handle is rebuilt after ``load_state_dict`` or any ``.to()/.cuda()``; in-place edits of the
    parameters afterwards need an explicit ``invalidate()`` (the sampling path never edits them:
    generator/train.py:91-92 freezes the classifier)."""

    def __init__(self):
        super().__init__()
        object.__setattr__(self, "_hip_handle", None)

    def invalidate(self) -> None:
        object.__setattr__(self, "_hip_handle", None)

    def _build_handle(self):
        raise NotImplementedError

    def handle(self):
        if self._hip_handle is None:
            if not torch.cuda.is_available():
                raise RuntimeError("dgdm_amd runs on an MI355X through libdgdm_hip.so; no GPU is visible and there is no CPU path")
            object.__setattr__(self, "_hip_handle", self._build_handle())
        return self._hip_handle

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.invalidate()
        return out

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self.invalidate()
        return out

    def plain_state_dict(self):
        return {k: v.detach().cpu() for k, v in self.state_dict().items()}
