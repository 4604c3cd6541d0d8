"""Input transform of the ViBERTgrid model — same class names, constructor and return types as the
reference (pipeline/transform.py: `ImageList` :9-33, `GeneralizedViBERTgridTransform` :36-312), with the
per-image normalise / bilinear resize / zero-pad-to-32 and the box rescale running as HIP kernels
that write straight into the padded batch (no intermediate per-image tensors).
"""
import math
from typing import List, Tuple

import torch
import torch.nn as nn

from vbg import ops


class ImageList(object):
    """padded batch tensor + the per-image sizes before padding"""

    def __init__(self, tensors: torch.Tensor, image_sizes: List[Tuple[int, int]]) -> None:
        self.tensors = tensors
        self.image_sizes = image_sizes

    def to(self, device):
        return ImageList(self.tensors.to(device), self.image_sizes)


def _triple(v, what):
    assert isinstance(v, (float, List)), f"{what} must be float or list of float, {type(v)} given"
    if isinstance(v, float):
        return [v] * 3
    if len(v) != 3:
        raise ValueError(f"{what} must contain 3 three values, {len(v)} given")
    return v


class GeneralizedViBERTgridTransform(nn.Module):
    def __init__(self, image_mean: List[float], image_std: List[float], train_min_size: List, test_min_size: int = 512,
                 max_size: int = 800):
        super().__init__()
        if not isinstance(train_min_size, (list, tuple)):
            train_min_size = list(train_min_size)
        self.train_min_size_list = train_min_size
        self.test_min_size = test_min_size
        self.max_size = max_size
        self.image_mean = _triple(image_mean, "image_mean")
        self.image_std = _triple(image_std, "image_std")

    def torch_choice(self, k: List[int]) -> int:
        # same draw as the reference (:124-131): torch's global CPU generator
        return k[int(torch.empty(1).uniform_(0.0, float(len(k))).item())]

    @staticmethod
    def _scale(h: int, w: int, self_min_size: float, self_max_size: float) -> float:
        mn, mx = float(min(h, w)), float(max(h, w))
        s = self_min_size / mn
        if mx * s > self_max_size:
            s = self_max_size / mx
        return s

    def plan(self, shapes):
        """host-side geometry: per-image resized (h, w) and the padded batch (H, W)"""
        sizes = []
        for (h, w) in shapes:
            size = float(self.torch_choice(self.train_min_size_list)) if self.training else float(self.test_min_size)
            s = self._scale(h, w, size, float(self.max_size))
            sizes.append((int(math.floor(float(h) * s)), int(math.floor(float(w) * s))))     # recompute_scale_factor=True
        H = int(math.ceil(float(max(s[0] for s in sizes)) / 32.0) * 32)
        W = int(math.ceil(float(max(s[1] for s in sizes)) / 32.0) * 32)
        return sizes, H, W

    def forward_nhwc(self, images, ocr_coors):
        """-> (batch NHWC fp32 [B,H,W,3], list of int32 [S,4] boxes, sizes)"""
        images, ocr_coors = list(images), list(ocr_coors)
        for image in images:
            if image.dim() != 3:
                raise ValueError("images is expected to be a list of 3d tensors of shape [C, H, W], got {}".format(image.shape))
        shapes = [tuple(im.shape[-2:]) for im in images]
        sizes, H, W = self.plan(shapes)
        dev = images[0].device
        batch = torch.zeros((len(images), H, W, 3), device=dev, dtype=torch.float32)
        out_coors = []
        for b, (im, (oh, ow)) in enumerate(zip(images, sizes)):
            im = im.to(torch.float32)
            ops.normalize_resize(im if im.is_contiguous() else im.contiguous(), oh, ow, self.image_mean, self.image_std, batch, b)
            h, w = shapes[b]
            c = ocr_coors[b]
            if c is not None:
                c = ops.rescale_boxes(c.long().contiguous(), oh / h, ow / w)     # cols 0,2 <- HEIGHT ratio (reference :167-168)
            out_coors.append(c)
        return batch, out_coors, sizes

    def forward(self, images: Tuple[torch.Tensor], ocr_coors: Tuple[torch.Tensor]):
        batch, out_coors, sizes = self.forward_nhwc(images, ocr_coors)
        return ImageList(ops.nhwc_to_nchw(batch), [(s[0], s[1]) for s in sizes]), out_coors
