"""Image pre/post-processing of the reference's user script without torchvision (reference test.py:27-31, 42-43):
Resize(size) [shorter side, bilinear + antialias on PIL images] -> CenterCrop(size) -> NormalizeToTensor, and
`save_image` for the [0,1] outputs of `decoding`."""
from __future__ import annotations

import torch
from PIL import Image

from .pipeline import NormalizeToTensor


def resize_shorter_side(img: Image.Image, size: int) -> Image.Image:
    """torchvision.transforms.Resize(int) on a PIL image: shorter side -> size, aspect kept, bilinear."""
    w, h = img.size
    if (w <= h and w == size) or (h <= w and h == size):
        return img
    if w < h:
        ow, oh = size, int(size * h / w)
    else:
        oh, ow = size, int(size * w / h)
    return img.resize((ow, oh), Image.BILINEAR)


def center_crop(img: Image.Image, size: int) -> Image.Image:
    w, h = img.size
    left, top = int(round((w - size) / 2.0)), int(round((h - size) / 2.0))
    return img.crop((left, top, left + size, top + size))


def load_image(path: str, size: int = 256) -> torch.Tensor:
    """-> float tensor [3,size,size] in [-1,1] (what `SelftokPipeline.encoding` takes, stacked along dim 0)."""
    img = Image.open(path).convert("RGB")
    return NormalizeToTensor()(center_crop(resize_shorter_side(img, size), size))


def save_image(img: torch.Tensor, path: str) -> None:
    """[3,H,W] in [0,1] (any float dtype/device) -> 8-bit file with torchvision.utils.save_image's arithmetic:
    `mul(255).add_(0.5).clamp_(0, 255)` evaluated IN THE TENSOR'S OWN DTYPE (the reference passes the bf16 output of `decoding`
    straight to torchvision, test.py:42-43, so the x*255+0.5 rounding happens in bf16), then truncation to uint8."""
    a = img.detach().cpu().clone().mul_(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()
    Image.fromarray(a).save(path)
