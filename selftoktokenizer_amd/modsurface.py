"""The read-only part of `torch.nn.Module`'s surface for the objects the pipeline exposes as `pipe.model`, `pipe.model.encoder`,
`pipe.model.model` and `pipe.vae` (reference: ImageTokenizer / QformerEncoder / MMDiT / AutoencoderKL, all nn.Modules -- users of the
reference call `.eval()`, `.to(device)`, `.state_dict()`, `.parameters()` on them: SelftokPipeline.py:163, 200-208).

The GPU objects are not Modules (their forward is a sequence of C-ABI launches over a flat weight dict, precomputed tables and
packed images), so this mixin presents that dict the way a Module would: same key names as the reference checkpoint, tensors on the
device, inference only."""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Iterator, Tuple

import torch


class ModuleSurface:
    training = False
    _sd_prefix = ""              # prefix of this object's keys inside the flat checkpoint (stripped by state_dict(), as a sub-module's would be)

    def _flat_weights(self) -> Dict[str, torch.Tensor]:
        return self.w

    def state_dict(self, prefix: str = "") -> "OrderedDict[str, torch.Tensor]":
        n = len(self._sd_prefix)
        return OrderedDict((prefix + k[n:], v) for k, v in self._flat_weights().items() if k.startswith(self._sd_prefix))

    def named_parameters(self, prefix: str = "", recurse: bool = True) -> Iterator[Tuple[str, torch.Tensor]]:
        for k, v in self.state_dict(prefix + ("." if prefix else "")).items():
            yield k, v

    def parameters(self, recurse: bool = True) -> Iterator[torch.Tensor]:
        for _, v in self.named_parameters():
            yield v

    def named_buffers(self, *a, **k):
        return iter(())

    def buffers(self, *a, **k):
        return iter(())

    def eval(self):
        return self

    def set_eval(self):
        return self

    def train(self, mode: bool = True):
        if mode:
            raise NotImplementedError("the MI355X hot path is inference only (the reference's training forward is out of scope)")
        return self

    def requires_grad_(self, requires_grad: bool = False):
        if requires_grad:
            raise NotImplementedError("the MI355X hot path is inference only")
        return self

    def _own_device(self) -> torch.device:
        return torch.device(self.device)

    def to(self, *args, **kwargs):
        """accepted when it is a no-op (same device, no dtype change), as the reference pipeline's own `.to(device)` calls are"""
        dev = kwargs.get("device", None)
        dt = kwargs.get("dtype", None)
        for a in args:
            if isinstance(a, (str, torch.device)):
                dev = a
            elif isinstance(a, torch.dtype):
                dt = a
        if dev is not None:
            d = torch.device(dev)
            own = self._own_device()
            if d.type != own.type or (d.index is not None and own.index is not None and d.index != own.index):
                raise NotImplementedError(f"weights live on {own}: build a SelftokPipeline(device=...) for {d} instead of moving this one")
        if dt is not None and dt not in {getattr(self, "dtype", dt)}:
            raise NotImplementedError("dtype conversion is not supported: the arithmetic of each stage is fixed (fp32 tokenizer, bf16 VAE)")
        return self

    def cuda(self, device=None):
        return self.to(torch.device("cuda", device) if isinstance(device, int) else "cuda")

    def load_state_dict(self, *a, **k):
        raise NotImplementedError("weights are packed at construction: build a new SelftokPipeline(state_dict=...) instead")
