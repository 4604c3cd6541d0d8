"""Word-level field-type classification: ROI embedding, late fusion and the three classifier heads.

Mirrors reference model/field_type_classification_head.py — `ROIEmbedding` :26-75, `SingleLayer` /
`MultipleLayer` :78-110, `BinaryClassifier` :111-127, `LateFusion` :130-190, `FieldTypeClassification` :193-407 (full),
`SimplifiedFieldTypeClassification` :410-588 (simp), `CRFFieldTypeClassification` :591-718 (crf) — with the
same parameter names.  The ROI convolutions are one implicit GEMM over all ROIs (M = N_roi*49), the
concat of ROI and BERT embeddings is a two-operand GEMM (no torch.cat), the MLPs fuse bias+ReLU into
the GEMM epilogue, the OHEM losses are device-driven (pipeline/custom_loss.py).
"""
from typing import Any, List, Optional, Tuple

import torch
import torch.nn as nn

from model.ResNetFPN_ViBERTgrid import _cl, conv_bn
from pipeline.custom_loss import BCELossOHEM, BCELossRandomSample, CrossEntropyLossOHEM, resolve_plans
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


class BinaryClassifier(nn.Module):
    """reference :111-127: `layer` is a SingleLayer or a MultipleLayer with one output"""

    def __init__(self, in_channels, bias: bool = True, layer_mode: str = "multi") -> None:
        super().__init__()
        assert layer_mode in ["single", "multi"], f"layer_mode must be single or multi, {layer_mode} given"
        self.layer = SingleLayer(in_channels, 1, bias=bias) if layer_mode == "single" else MultipleLayer(in_channels, 1)

    def forward(self, x):
        return self.layer(x)


class AttrProxy(object):
    """indexable view of numbered sub-modules (reference utils: `category_classification_net[i]`)"""

    def __init__(self, module, prefix):
        self.module, self.prefix = module, prefix

    def __getitem__(self, i):
        return getattr(self.module, self.prefix + str(i))


class FieldTypeClassification(nn.Module):
    """the paper's two-stage classifier (reference :193-407): a binary key / non-key net over all ROIs with
    BCELossRandomSample([num_hard_negative_1, num_hard_positive_1]), then one binary net per foreground class evaluated only on the
    rows PREDICTED positive (sigmoid >= 0.5), each with a BCELossOHEM.  The row subset is a device compaction + one row
    gather (its size is the only host read); parameter names as in the reference (`category_classification_net_{i}`)."""

    def __init__(self, num_classes: int, fuse_embedding_channel: int, loss_weights: Optional[List] = None,
                 num_hard_positive_1: int = -1, num_hard_negative_1: int = -1, num_hard_positive_2: int = -1,
                 num_hard_negative_2: int = -1, random: bool = False, layer_mode: str = "multi", work_mode: str = "train") -> None:
        super().__init__()
        assert work_mode in ["train", "eval", "inference"], f"mode must be 'train' 'eval' or 'inference', {work_mode} given"
        self.work_mode, self.num_classes, self.fuse_embedding_channel = work_mode, num_classes, fuse_embedding_channel
        self.pos_neg_classification_net = BinaryClassifier(fuse_embedding_channel, bias=True, layer_mode=layer_mode)
        self.pos_neg_classification_loss = None if work_mode == "inference" else BCELossRandomSample(
            sample_list=[num_hard_negative_1, num_hard_positive_1])
        for idx in range(num_classes - 1):
            self.add_module(f"category_classification_net_{idx}", BinaryClassifier(fuse_embedding_channel, bias=True, layer_mode=layer_mode))
            if work_mode != "inference":
                self.add_module(f"field_type_classification_loss_{idx}",
                                BCELossOHEM(num_hard_positive=num_hard_positive_2, num_hard_negative=num_hard_negative_2,
                                            weight=loss_weights, random=random))
        self.category_classification_net = AttrProxy(self, "category_classification_net_")
        self.field_type_classification_loss = None if work_mode == "inference" else AttrProxy(self, "field_type_classification_loss_")

    def _stage_two(self, fuse, pred_pos_neg, label=None):
        """-> (sum of per-class losses or None, class_pred [N, ncls])"""
        prob = torch.sigmoid(pred_pos_neg.detach())
        mask = prob.ge(0.5).to(torch.int32)
        idx, cnt = ops.compact(mask, 1, True)
        n = int(cnt.item())
        class_pred = torch.zeros((fuse.shape[0], self.num_classes), dtype=pred_pos_neg.dtype, device=fuse.device)
        class_pred[:, 0] = prob
        loss = torch.zeros((1,), device=fuse.device)
        if n:
            idx = idx[:n].contiguous()
            pos = Fn.GatherRowsFn.apply(fuse, idx)
            lab = ops.gather_i32(label, idx) if label is not None else None
            for ci in range(self.num_classes - 1):
                cp = self.category_classification_net[ci](pos).reshape(-1)
                if lab is not None:
                    loss = loss + self.field_type_classification_loss[ci](cp, (lab == ci + 1).to(torch.float32))
                class_pred[idx.long(), ci + 1] = torch.sigmoid(cp.detach())
        return loss, class_pred

    def inference(self, fuse_embeddings: torch.Tensor):
        fuse = fuse_embeddings.reshape((-1, self.fuse_embedding_channel))
        pred_pos_neg = self.pos_neg_classification_net(fuse).reshape(-1)
        return self._stage_two(fuse, pred_pos_neg)[1]

    def forward(self, fuse_embeddings: torch.Tensor, segment_classes: Tuple[torch.Tensor], prepared=None):
        label = torch.cat([c.reshape(-1) for c in segment_classes], dim=0).to(fuse_embeddings.device).int()
        fuse = fuse_embeddings.reshape((-1, self.fuse_embedding_channel))
        assert fuse.shape[0] == label.shape[0]
        pred_pos_neg = self.pos_neg_classification_net(fuse).reshape(-1)
        loss_pn = self.pos_neg_classification_loss(pred_pos_neg, (label > 0).to(torch.float32))
        loss_c, class_pred = self._stage_two(fuse, pred_pos_neg, label)
        return loss_pn + loss_c, label, class_pred


class CRFFieldTypeClassification(nn.Module):
    """emission net over ncls + 2 tags + linear-chain CRF (reference :591-718); all documents of the batch go through one
    forward-algorithm / Viterbi launch (model/crf.py in this package)."""

    def __init__(self, tag_to_idx, fuse_embedding_channel: int, layer_mode: str = "multi", work_mode: str = "train") -> None:
        super().__init__()
        from model.crf import CRF, START_TAG, STOP_TAG
        assert work_mode in ["train", "eval", "inference"], f"mode must be 'train' 'eval' or 'inference', {work_mode} given"
        assert layer_mode in ["single", "multi"], f"layer_mode must be single or multi, {layer_mode} given"
        self.work_mode = work_mode
        self.num_classes = len(tag_to_idx)
        self.num_tags = self.num_classes + 2
        assert max(tag_to_idx.values()) == self.num_classes - 1, "invalid tag_to_idx format"
        self.tag_to_idx = tag_to_idx
        self.tag_to_idx[START_TAG] = self.num_classes
        self.tag_to_idx[STOP_TAG] = self.num_classes + 1
        self.fuse_embedding_channel = fuse_embedding_channel
        net = SingleLayer if layer_mode == "single" else MultipleLayer
        self.category_classification_net = net(fuse_embedding_channel, self.num_tags, bias=True)
        self.crf_layer = CRF(self.tag_to_idx)

    @staticmethod
    def _doc_off(lens, device):
        off = [0]
        for n in lens:
            off.append(off[-1] + int(n))
        return torch.tensor(off, dtype=torch.int32).to(device)

    def inference(self, fuse_embeddings: torch.Tensor):
        fuse = fuse_embeddings.reshape((-1, self.fuse_embedding_channel))
        em = self.category_classification_net(fuse).detach()
        path, _ = self.crf_layer.decode(em, self._doc_off([em.shape[0]], em.device))
        return path.unsqueeze(1).float()

    def forward(self, fuse_embeddings: torch.Tensor, segment_classes: Tuple[torch.Tensor], prepared=None):
        device = fuse_embeddings.device
        lens = [int(b.shape[0]) for b in segment_classes]
        label = torch.cat([c.reshape(-1) for c in segment_classes], dim=0).to(device)
        fuse = fuse_embeddings.reshape((-1, self.fuse_embedding_channel))
        assert fuse.shape[0] == label.shape[0]
        em = self.category_classification_net(fuse)
        doc_off = self._doc_off(lens, device)
        if self.training:
            nll = self.crf_layer.nll(em, label.int(), doc_off)
            return nll.sum().reshape(1) / len(lens), label.int(), em.detach().float()
        path, score = self.crf_layer.decode(em.detach(), doc_off)
        return score.sum().reshape(1) / len(lens), label.int(), path.unsqueeze(1).float()
