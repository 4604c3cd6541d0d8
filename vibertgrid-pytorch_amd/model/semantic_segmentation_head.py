"""Auxiliary semantic segmentation heads (simplified and two-stage variants) on MI355X kernels.

Mirrors reference model/semantic_segmentation_head.py: `SemanticSegmentationEncoder` :23-78,
`SimplifiedSemanticSegmentationClassifier` :236-352 (classifier_mode simp) and `SemanticSegmentationClassifier` :100-233 with its
per-class `SemanticSegmentationBinaryClassifier` :81-97 (classifier_mode full / crf), same parameter names.  Differences in HOW:
* the two 1x1 classifiers run at P_fuse resolution; nearest x4 upsampling commutes exactly with a
  1x1 conv + bias, so the reference's [B,256,H,W] activation (268 MB/doc at 512x512) never exists;
* labels are rasterised by one owner-map kernel at stride 1 (bit-exact last-writer-wins) instead of
  2*S syncing slice assignments (:326-341);
* the sampled / OHEM losses evaluate CE only where needed, indexing low-res logits by full-res pixel.
"""
from typing import List, Tuple

import torch
import torch.nn as nn

from model.ResNetFPN_ViBERTgrid import _cl, conv, conv_bn
from pipeline.custom_loss import BCELossOHEM, CrossEntropyLossOHEM, CrossEntropyLossRandomSample, resolve_plans
from vbg import ops


class SemanticSegmentationEncoder(nn.Module):
    def __init__(self, fuse_channel: int, num_classes: int) -> None:
        super().__init__()
        self.conv_1 = nn.Conv2d(fuse_channel, fuse_channel, 3, 1, 1, bias=False)
        self.bn_1 = nn.BatchNorm2d(fuse_channel)
        self.conv_2 = nn.Conv2d(fuse_channel, fuse_channel, 3, 1, 1, bias=False)
        self.bn_2 = nn.BatchNorm2d(fuse_channel)
        self.upsampling = nn.UpsamplingNearest2d(scale_factor=4)
        self.conv_3_1 = nn.Conv2d(fuse_channel, 3, kernel_size=1)
        self.conv_3_2 = nn.Conv2d(fuse_channel, num_classes, kernel_size=1)
        _cl(self)

    def forward(self, x):
        """x NHWC P_fuse -> LOW-RES logits NHWC ([B,h,w,3], [B,h,w,ncls]); the x4 upsampling is implicit"""
        x = conv_bn(x, self.conv_1, self.bn_1, None, True)
        x = conv_bn(x, self.conv_2, self.bn_2, None, True)
        return conv(x, self.conv_3_1), conv(x, self.conv_3_2)


class SimplifiedSemanticSegmentationClassifier(nn.Module):
    def __init__(self, p_fuse_channel: int, num_classes: int, loss_weights: torch.Tensor = None, loss_1_sample_list: List = None,
                 num_hard_positive: int = -1, num_hard_negative: int = -1) -> None:
        super().__init__()
        self.semantic_segmentation_encoder = SemanticSegmentationEncoder(p_fuse_channel, num_classes)
        self.aux_loss_1 = CrossEntropyLossRandomSample(sample_list=loss_1_sample_list)          # never weighted (:279-283)
        self.aux_loss_2 = CrossEntropyLossOHEM(num_hard_positive=num_hard_positive, num_hard_negative=num_hard_negative,
                                               weight=loss_weights)

    def make_labels(self, packed_boxes, classes_i32, B, H, W):
        """-> (pos_neg int32 [B*H*W], class int32 [B*H*W]) flat full-resolution labels"""
        boxes, box_off, _ = packed_boxes
        owner = ops.owner_map(boxes, box_off, B, H, W, 1)
        pn, cl = ops.label_raster(owner, classes_i32)
        return pn.view(-1), cl.view(-1)

    def plans(self, pos_neg, cls):
        return [self.aux_loss_1.plan(pos_neg, 3), self.aux_loss_2.plan(cls)]

    def forward(self, fuse_feature: torch.Tensor, seg_classes: Tuple[torch.Tensor], coors: Tuple[torch.Tensor], prepared=None,
                materialize: bool = True):
        """fuse_feature NHWC.  Returns (aux_loss, pred_mask [B,3,H,W], pred_ss [B,ncls,H,W]) like the reference; the
        two full-resolution maps are only materialised when `materialize` (eval / API parity)."""
        x1, x2 = self.semantic_segmentation_encoder(fuse_feature)
        B, h, w, _ = x1.shape
        H, W = 4 * h, 4 * w
        if prepared is None:
            from model.BERTgrid_generator import BERTgridGenerator
            packed = BERTgridGenerator.pack_boxes(tuple(c.int() for c in coors))
            classes = torch.cat([c.reshape(-1) for c in seg_classes]).int()
            pos_neg, cls = self.make_labels(packed, classes, B, H, W)
            plans = self.plans(pos_neg, cls)
            resolve_plans(plans)
        else:
            pos_neg, cls, plans = prepared
        l1 = self.aux_loss_1(x1.reshape(-1, 3), pos_neg, plans[0], 2, H, W)
        l2 = self.aux_loss_2(x2.reshape(-1, x2.shape[-1]), cls, plans[1], 2, H, W)
        if materialize:
            return l1 + l2, ops.upsample_nhwc_to_nchw(x1.detach(), 4), ops.upsample_nhwc_to_nchw(x2.detach(), 4)
        return l1 + l2, None, None


class SemanticSegmentationBinaryClassifier(nn.Module):
    def __init__(self, in_channels: int) -> None:
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, 1, kernel_size=1)
        _cl(self)

    def forward(self, x):
        return conv(x, self.conv1)


class _Proxy(object):
    def __init__(self, module, prefix):
        self.module, self.prefix = module, prefix

    def __getitem__(self, i):
        return getattr(self.module, self.prefix + str(i))


class SemanticSegmentationClassifier(nn.Module):
    """the paper's two-stage auxiliary head used by classifier_mode full / crf (reference :100-233): aux_loss_1 as in the simplified
    head; on the pixels PREDICTED positive (`softmax(x_out_1).argmax(1) == 1`) one 1x1 conv (ncls -> 1) per foreground class with a
    BCE-OHEM loss against `class_label == idx + 1`.  Everything stays at P_fuse resolution: the per-class 1x1 convs commute with
    the nearest x4 upsampling like conv_3_1 / conv_3_2 do, the predicted-positive set is the x4 replication of the low-resolution
    argmax, and the losses index low-resolution logits by full-resolution pixel."""

    def __init__(self, p_fuse_channel: int, num_classes: int, loss_weights: torch.Tensor = None, loss_1_sample_list: List = None,
                 num_hard_positive: int = -1, num_hard_negative: int = -1) -> None:
        super().__init__()
        self.num_classes = num_classes
        self.ss_encoder = SemanticSegmentationEncoder(p_fuse_channel, num_classes)
        self.aux_loss_1 = CrossEntropyLossRandomSample(sample_list=loss_1_sample_list)
        for idx in range(num_classes - 1):
            self.add_module(f"ss_binary_classifier_{idx}", SemanticSegmentationBinaryClassifier(in_channels=num_classes))
            self.add_module(f"aux_loss_2_{idx}", BCELossOHEM(num_hard_positive=num_hard_positive, num_hard_negative=num_hard_negative,
                                                             weight=loss_weights))
        self.ss_binary_classifier = _Proxy(self, "ss_binary_classifier_")
        self.aux_loss_2 = _Proxy(self, "aux_loss_2_")

    def forward(self, fuse_feature: torch.Tensor, seg_classes: Tuple[torch.Tensor], coors: Tuple[torch.Tensor], prepared=None,
                materialize: bool = True):
        from model.BERTgrid_generator import BERTgridGenerator
        x1, x2 = self.ss_encoder(fuse_feature)
        B, h, w, _ = x1.shape
        H, W = 4 * h, 4 * w
        packed = BERTgridGenerator.pack_boxes(tuple(c.int() for c in coors))
        classes = torch.cat([c.reshape(-1) for c in seg_classes]).int()
        boxes, box_off, _ = packed
        owner = ops.owner_map(boxes, box_off, B, H, W, 1)
        pos_neg, cls = ops.label_raster(owner, classes)
        pos_neg, cls = pos_neg.view(-1), cls.view(-1)
        plan = self.aux_loss_1.plan(pos_neg, 3)
        resolve_plans([plan])
        loss = self.aux_loss_1(x1.reshape(-1, 3), pos_neg, plan, 2, H, W)
        # predicted-positive pixels: low-resolution argmax == 1, replicated x4 (torch.argmax keeps the first maximum, like softmax().argmax())
        pm = (x1.detach().argmax(dim=-1) == 1)
        l2 = torch.zeros((1,), device=x1.device)
        if bool(pm.any()):
            pmf = pm.repeat_interleave(4, dim=1).repeat_interleave(4, dim=2).reshape(-1)
            for ci in range(self.num_classes - 1):
                logit = self.ss_binary_classifier[ci](x2).reshape(-1, 1)                     # low-resolution rows
                key = torch.where(pmf, (cls == ci + 1).to(torch.int32), torch.full_like(cls, 2))
                l2 = l2 + self.aux_loss_2[ci](logit, keyed_labels=key, up_shift=2, H=H, W=W)
        loss = loss + l2
        if materialize:
            return loss, ops.upsample_nhwc_to_nchw(x1.detach(), 4), ops.upsample_nhwc_to_nchw(x2.detach(), 4)
        return loss, None, None
