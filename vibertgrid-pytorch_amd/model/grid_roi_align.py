"""GridROIAlign — mirror of the reference module (model/grid_roi_align.py:8-83): RoIAlign(output_size,
spatial_scale=1/step, sampling_ratio=-1, aligned=False) over P_fuse, as a channel-coalesced NHWC HIP
kernel (forward gather, backward wavefront atomics).  Boxes are the integer image-space boxes of the
transform; the reference casts them to float (:73) and lets torchvision scale them."""
from typing import Any, Tuple

import torch
import torch.nn as nn

from vbg import functions as Fn


class GridROIAlign(nn.Module):
    def __init__(self, output_size: Any = 7, step: int = 4) -> None:
        super().__init__()
        if isinstance(output_size, int) or isinstance(output_size, Tuple):
            self.output_size = output_size
        else:
            raise TypeError(f"parameter 'output_size' requires int or tuple, {type(output_size)} were given")
        if isinstance(output_size, tuple):
            assert output_size[0] == output_size[1], "square ROI outputs only"
        self.spatial_scale = 1 / float(step)

    def forward(self, feature_map: torch.Tensor, coords: Tuple[torch.Tensor], mask: torch.Tensor = None,
                packed=None) -> torch.Tensor:
        """feature_map NHWC [B,H,W,C]; coords: per-image [S_b,4] boxes -> [sum S_b, out, out, C] (NHWC)."""
        if packed is None:
            from model.BERTgrid_generator import BERTgridGenerator
            if mask is not None:
                coords = tuple(c[mask[b] == 1] for b, c in enumerate(coords))
            packed = BERTgridGenerator.pack_boxes(tuple(coords))
        boxes, _, box_doc = packed
        out = self.output_size if isinstance(self.output_size, int) else self.output_size[0]
        return Fn.RoiAlignFn.apply(feature_map, boxes, box_doc, out, self.spatial_scale)
