"""Word-level field-type classification: ROI embedding, late fusion and the simplified classifier.

Mirrors reference model/field_type_classification_head.py — `ROIEmbedding` :26-75, `SingleLayer` /
`MultipleLayer` :78-110, `LateFusion` :130-190, `SimplifiedFieldTypeClassification` :410-588 — with the
same parameter names.  The ROI convolutions are one implicit GEMM over all ROIs (M = N_roi*49), the
concat of ROI and BERT embeddings is a two-operand GEMM (no torch.cat), the MLPs fuse bias+ReLU into
the GEMM epilogue, the OHEM losses are device-driven (pipeline/custom_loss.py).
"""
from typing import Any, List, Optional, Tuple

import torch
import torch.nn as nn

from model.ResNetFPN_ViBERTgrid import _cl, conv_bn
from pipeline.custom_loss import CrossEntropyLossOHEM, resolve_plans
from vbg import functions as Fn
from vbg import ops


class ROIEmbedding(nn.Module):
    def __init__(self, num_channels: int, roi_shape: Any) -> None:
        super().__init__()
        if isinstance(roi_shape, Tuple):
            assert len(roi_shape) == 2, f"roi_shape must be int or two-element tuple, {len(roi_shape)} elements were given"
            num_flatten = num_channels * roi_shape[0] * roi_shape[1]
        elif isinstance(roi_shape, int):
            num_flatten = num_channels * roi_shape * roi_shape
        else:
            raise ValueError("roi_shape must be int or two-element tuple")
        self.conv_1 = nn.Conv2d(num_channels, num_channels, 3, 1, 1, bias=False)
        self.bn_1 = nn.BatchNorm2d(num_channels)
        self.conv_2 = nn.Conv2d(num_channels, num_channels, 3, 1, 1, bias=False)
        self.bn_2 = nn.BatchNorm2d(num_channels)
        self.linear = nn.Linear(num_flatten, 1024)
        _cl(self)

    def forward(self, ROI: torch.Tensor) -> torch.Tensor:
        """ROI NHWC [N,h,w,C] -> [N,1024]"""
        x = conv_bn(ROI, self.conv_1, self.bn_1, None, True)
        x = conv_bn(x, self.conv_2, self.bn_2, None, True)
        x = Fn.NhwcToNchwFlatFn.apply(x)            # the reference's (c, y, x) flatten order
        return Fn.LinearFn.apply(x, self.linear.weight, self.linear.bias, False)


class SingleLayer(nn.Module):
    def __init__(self, in_features, out_features, bias: bool = True) -> None:
        super().__init__()
        self.linear = nn.Linear(in_features, out_features, bias=bias)

    def forward(self, x):
        return Fn.LinearFn.apply(x, self.linear.weight, self.linear.bias, False)


class MultipleLayer(nn.Module):
    def __init__(self, in_features, out_features, bias: bool = True) -> None:
        super().__init__()
        self.linear_1 = nn.Linear(in_features, in_features // 2, bias=bias)
        self.linear_2 = nn.Linear(in_features // 2, out_features, bias=bias)

    def forward(self, x):
        h = Fn.LinearFn.apply(x, self.linear_1.weight, self.linear_1.bias, True)
        return Fn.LinearFn.apply(h, self.linear_2.weight, self.linear_2.bias, False)


class LateFusion(nn.Module):
    def __init__(self, bert_hidden_size: int, roi_channel: int, roi_shape: Any) -> None:
        super().__init__()
        self.BERT_dimension = bert_hidden_size
        if isinstance(roi_shape, int):
            ROI_output = (roi_shape, roi_shape)
        elif isinstance(roi_shape, Tuple):
            ROI_output = roi_shape
        else:
            raise TypeError(f"roi_shape must be int or Tuple, {type(roi_shape)} given")
        self.ROI_embedding_net = ROIEmbedding(num_channels=roi_channel, roi_shape=(ROI_output[0], ROI_output[1]))
        self.fuse_embedding_net = SingleLayer(in_features=self.BERT_dimension + 1024, out_features=1024, bias=True)

    def forward(self, ROI_output: torch.Tensor, BERT_embeddings):
        """ROI_output NHWC [N,h,w,C]; BERT_embeddings: tuple of [S_b,768] or the concatenated [N,768]"""
        roi_emb = self.ROI_embedding_net(ROI_output)
        bert = BERT_embeddings if isinstance(BERT_embeddings, torch.Tensor) else torch.cat(list(BERT_embeddings), dim=0)
        assert roi_emb.shape[0] == bert.shape[0]
        lin = self.fuse_embedding_net.linear
        return Fn.SegLinearFn.apply(lin.weight, lin.bias, (0, 0), (0, 0), roi_emb, bert)


class SimplifiedFieldTypeClassification(nn.Module):
    def __init__(self, num_classes: int, fuse_embedding_channel: int, loss_weights: Optional[List] = None,
                 num_hard_positive_1: int = -1, num_hard_negative_1: int = -1, num_hard_positive_2: int = -1,
                 num_hard_negative_2: int = -1, random: bool = False, layer_mode: str = "multi", work_mode: str = "train",
                 add_pos_neg: bool = True) -> None:
        super().__init__()
        assert work_mode in ["train", "eval", "inference"], f"mode must be 'train' 'eval' or 'inference', {work_mode} given"
        assert layer_mode in ["single", "multi"], f"layer_mode must be single or multi, {layer_mode} given"
        self.work_mode, self.num_classes, self.fuse_embedding_channel = work_mode, num_classes, fuse_embedding_channel
        # The reference compares layer_mode with the misspelt "sigle" (:474), so BOTH modes build the 2-layer MLP.
        self.pos_neg_classification_net = None if work_mode == "inference" else MultipleLayer(fuse_embedding_channel, 2, bias=True)
        self.category_classification_net = MultipleLayer(fuse_embedding_channel, num_classes)
        if work_mode == "inference":
            self.pos_neg_classification_loss = None
            self.field_type_classification_loss = None
        else:
            self.pos_neg_classification_loss = CrossEntropyLossOHEM(num_hard_positive_1, num_hard_negative_1, random=random)
            self.field_type_classification_loss = CrossEntropyLossOHEM(num_hard_positive_2, num_hard_negative_2, weight=loss_weights,
                                                                       random=random)
        self.add_pos_neg = add_pos_neg

    def inference(self, fuse_embeddings: torch.Tensor):
        fuse_embeddings = fuse_embeddings.reshape((-1, self.fuse_embedding_channel))
        return ops.row_softmax(self.category_classification_net(fuse_embeddings).detach())

    def make_labels(self, segment_classes):
        label_class = torch.cat([c.reshape(-1) for c in segment_classes], dim=0).int()
        return label_class, (label_class > 0).int()

    def plans(self, label_class, label_pos_neg):
        return [self.pos_neg_classification_loss.plan(label_pos_neg), self.field_type_classification_loss.plan(label_class)]

    def forward(self, fuse_embeddings: torch.Tensor, segment_classes: Tuple[torch.Tensor], prepared=None):
        fuse_embeddings = fuse_embeddings.reshape((-1, self.fuse_embedding_channel))
        if prepared is None:
            label_class, label_pos_neg = self.make_labels(segment_classes)
            plans = self.plans(label_class, label_pos_neg)
            resolve_plans(plans)
        else:
            label_class, label_pos_neg, plans = prepared
        assert fuse_embeddings.shape[0] == label_class.shape[0]
        pred_pos_neg = self.pos_neg_classification_net(fuse_embeddings)
        loss_pn = self.pos_neg_classification_loss(pred_pos_neg, label_pos_neg, plans[0])
        pred_class = self.category_classification_net(fuse_embeddings)
        loss_c = self.field_type_classification_loss(pred_class, label_class, plans[1])
        loss = loss_pn + loss_c if self.add_pos_neg else loss_c
        return loss, label_class.int(), ops.row_softmax(pred_class.detach())


class FieldTypeClassification(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError("classifier_mode 'full' (two-stage binary classifiers) is not built yet; use 'simp'")


class CRFFieldTypeClassification(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError("classifier_mode 'crf' is not built yet; use 'simp'")
