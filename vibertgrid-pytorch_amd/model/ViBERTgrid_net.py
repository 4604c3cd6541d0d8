"""ViBERTgridNet — the drop-in boundary (reference model/ViBERTgrid_net.py:37-544).

Same constructor keywords / defaults / validation errors (:128-460), same `forward` (:501-544),
`inference` (:470-499), `train()` / `eval()` work_mode flips (:462-468), same public attributes
(`tokenizer`, `bert_model`, `backbone`, `transform`, `work_mode`, `bert_model_list`, `backbone_list`)
and the same `state_dict()` key layout (BERT registered under `bert_model.` and again under
`BERTgrid_generator.model.`).  Everything numeric below this class runs in libvbg (HIP, gfx950);
importing this module without the built library raises ImportError — there is no fallback.

What differs from the reference is only HOW the step is scheduled: label rasterisation, category
compaction and the host random draws of all four losses are planned up-front with a single
device->host copy, BERT runs on packed (un-padded) tokens, the grid / P_fuse / late-fusion concats
are never materialised, and activations are NHWC.
"""
from typing import Any, Dict, List, Tuple

import contextlib

import torch
import torch.nn as nn
from transformers import AutoConfig, BertModel, BertTokenizer, RobertaModel, RobertaTokenizer

from model.BERTgrid_generator import BERTgridGenerator
from model.field_type_classification_head import (CRFFieldTypeClassification, FieldTypeClassification, LateFusion,
                                                   SimplifiedFieldTypeClassification)
from model.grid_roi_align import GridROIAlign
from model.ResNetFPN_ViBERTgrid import bn_tick_scope, resnet_18_D_fpn, resnet_18_fpn, resnet_34_D_fpn, resnet_34_fpn
from model.semantic_segmentation_head import SemanticSegmentationClassifier, SimplifiedSemanticSegmentationClassifier
from vbg import functions as Fn
from vbg import ops
from pipeline.custom_loss import PendingCounts, resolve_plans  # noqa: F401
from pipeline.transform import GeneralizedViBERTgridTransform, ImageList  # noqa: F401  (ImageList re-exported like the reference)

_BERT_HIDDEN = {
    "private_bert-base-uncased": 768, "bert-base-uncased": 768, "bert-base-cased": 768, "roberta-base": 768,
    "bert-base-chinese": 768, "hfl/chinese-bert-wwm-ext": 768, "hfl/chinese-bert-wwm": 768,
}
_BACKBONES = {
    "resnet_18_fpn": lambda c: resnet_18_fpn(grid_channel=c),
    "resnet_34_fpn": lambda c: resnet_34_fpn(grid_channel=c),
    "resnet_18_fpn_pretrained": lambda c: resnet_18_fpn(grid_channel=c, pretrained=True),
    "resnet_34_fpn_pretrained": lambda c: resnet_34_fpn(grid_channel=c, pretrained=True),
    "resnet_18_D_fpn": lambda c: resnet_18_D_fpn(grid_channel=c),
    "resnet_34_D_fpn": lambda c: resnet_34_D_fpn(grid_channel=c),
}


def _three(v, name):
    assert isinstance(v, (float, List)), f"{name} must be float or list of float, {type(v)} given"
    if isinstance(v, float):
        return [v] * 3
    if len(v) != 3:
        raise ValueError(f"{name} must contain 3 three values, {len(v)} given")
    return v


class ViBERTgridNet(nn.Module):
    def __init__(self, num_classes, image_mean: Any, image_std: Any, image_min_size: Any, image_max_size, test_image_min_size=512,
                 bert_model: str = "bert-base-uncased", tokenizer: Any = None, backbone: str = "resnet_18_fpn", grid_mode: str = "mean",
                 early_fusion_downsampling_ratio=8, roi_shape=7, p_fuse_downsampling_ratio=4, late_fusion_fuse_embedding_channel=1024,
                 loss_weights: Any = None, num_hard_positive_main_1=-1, num_hard_negative_main_1=-1, num_hard_positive_main_2=-1,
                 num_hard_negative_main_2=-1, loss_aux_sample_list: List = None, num_hard_positive_aux=-1, num_hard_negative_aux=-1,
                 loss_control_lambda: float = 1, add_pos_neg: bool = True, classifier_mode: str = "full", tag_to_idx: Dict = None,
                 ohem_random: bool = False, layer_mode: str = "single", work_mode: str = "train") -> None:
        super().__init__()
        assert work_mode in ["train", "eval", "inference"], f"mode must be 'train' 'eval' or 'inference', {work_mode} given"
        self.work_mode = work_mode
        self.num_classes = num_classes
        self.num_tokens = len(tag_to_idx) if tag_to_idx is not None else num_classes

        # ---- pre-processing -----------------------------------------------------------------
        self.image_mean = _three(image_mean, "image_mean")
        self.image_std = _three(image_std, "image_std")
        self.test_image_min_size = test_image_min_size
        assert isinstance(image_min_size, (int, Tuple, List)), f"image_min_size must be int, Tuple or List, {type(image_min_size)} given"
        image_min_size = list(image_min_size)
        assert isinstance(image_max_size, int), f"image_max_size must be int, {type(image_max_size)} given"
        self.image_min_size, self.image_max_size = image_min_size, image_max_size
        self.transform = GeneralizedViBERTgridTransform(image_mean=self.image_mean, image_std=self.image_std,
                                                        train_min_size=self.image_min_size, test_min_size=self.test_image_min_size,
                                                        max_size=self.image_max_size)

        # ---- language model (parameter owner; arithmetic is libvbg) ----------------------------
        self.bert_model_list = dict(_BERT_HIDDEN)
        assert bert_model in self.bert_model_list.keys(), \
            f"the given bert model {bert_model} does not exists, see attribute bert_model_list for all bert_models"
        self.bert_hidden_size = self.bert_model_list[bert_model]
        is_roberta = "roberta-" in bert_model
        if not is_roberta and "bert-" not in bert_model:
            raise ValueError("no tokenizer and bert model loaded")
        tok_cls, model_cls, tok_name = ((RobertaTokenizer, RobertaModel, "RobertaTokenizer") if is_roberta
                                        else (BertTokenizer, BertModel, "BertTokenizer"))
        if tokenizer is None:
            self.tokenizer = tok_cls.from_pretrained(bert_model)
        elif isinstance(tokenizer, tok_cls):
            self.tokenizer = tokenizer
        else:
            raise ValueError(f"invalid value of parameter tokenizer, must be None or callable {tok_name}")
        if self.work_mode in ("train", "inference"):
            print("loading pretrained")
            self.bert_model = model_cls.from_pretrained(bert_model)
        else:
            print("in evaluation mode, no pretrained will be loaded")
            self.bert_config = AutoConfig.from_pretrained(bert_model)
            self.bert_model = model_cls(self.bert_config)

        # ---- backbone ---------------------------------------------------------------------------
        self.backbone_list = list(_BACKBONES.keys())
        assert backbone in self.backbone_list, \
            f"the given backbone {backbone} does not exists, see attribute backbone_list for all backbones"
        self.backbone = _BACKBONES[backbone](self.bert_hidden_size)
        self.p_fuse_channel = 256

        assert grid_mode in ["mean", "first"], f"grid_mode should be 'mean' or 'first', {grid_mode} were given"
        self.grid_mode = grid_mode
        self.early_fusion_downsampling_ratio = early_fusion_downsampling_ratio
        self.roi_shape = roi_shape
        self.p_fuse_downsampling_ratio = p_fuse_downsampling_ratio
        self.late_fusion_fuse_embedding_channel = late_fusion_fuse_embedding_channel

        # ---- losses -----------------------------------------------------------------------------
        self.loss_control_lambda = None if self.work_mode == "inference" else loss_control_lambda
        if loss_weights is None or self.work_mode == "inference":
            self.loss_weights = None
        elif isinstance(loss_weights, List):
            self.loss_weights = torch.tensor(loss_weights)
        elif isinstance(loss_weights, torch.Tensor):
            pass          # (sic) the reference leaves self.loss_weights unset for a Tensor argument (:344-345)
        else:
            raise TypeError(f"loss_weights must be None, List or torch.Tensor, {type(loss_weights)} given")
        assert classifier_mode in ["full", "simp", "crf"], "invalid classifier mode, must be 'full', 'simp' or 'crf'"
        self.classifier_mode = classifier_mode

        self.BERTgrid_generator = BERTgridGenerator(bert_model=self.bert_model, grid_mode=self.grid_mode,
                                                    stride=self.early_fusion_downsampling_ratio)
        self.grid_roi_align_net = GridROIAlign(output_size=self.roi_shape, step=self.p_fuse_downsampling_ratio)
        self.late_fusion_net = LateFusion(bert_hidden_size=self.bert_hidden_size, roi_channel=self.p_fuse_channel, roi_shape=self.roi_shape)

        inference = self.work_mode == "inference"
        if self.classifier_mode == "simp":
            if inference:
                self.field_type_classification_head = SimplifiedFieldTypeClassification(
                    num_classes=self.num_tokens, fuse_embedding_channel=self.late_fusion_fuse_embedding_channel, layer_mode=layer_mode,
                    work_mode=self.work_mode, add_pos_neg=add_pos_neg)
                self.semantic_segmentation_head = None
            else:
                lw = getattr(self, "loss_weights", None)
                lw = None if lw is None else lw.to(torch.float32)
                # NOTE the reference does not forward add_pos_neg here (:417-428): the head default (True) applies
                self.field_type_classification_head = SimplifiedFieldTypeClassification(
                    num_classes=self.num_tokens, fuse_embedding_channel=self.late_fusion_fuse_embedding_channel, loss_weights=lw,
                    num_hard_positive_1=num_hard_positive_main_1, num_hard_negative_1=num_hard_negative_main_1,
                    num_hard_positive_2=num_hard_positive_main_2, num_hard_negative_2=num_hard_negative_main_2, random=ohem_random,
                    layer_mode=layer_mode, work_mode=self.work_mode)
                self.semantic_segmentation_head = SimplifiedSemanticSegmentationClassifier(
                    p_fuse_channel=self.p_fuse_channel, num_classes=self.num_tokens, loss_weights=lw,
                    loss_1_sample_list=loss_aux_sample_list, num_hard_positive=num_hard_positive_aux,
                    num_hard_negative=num_hard_negative_aux)
        else:
            lw = None if inference else getattr(self, "loss_weights", None)
            lw = None if lw is None else lw.to(torch.float32)
            if self.classifier_mode == "full":
                if inference:
                    self.field_type_classification_head = FieldTypeClassification(
                        num_classes=self.num_tokens, fuse_embedding_channel=self.late_fusion_fuse_embedding_channel, layer_mode=layer_mode,
                        work_mode=self.work_mode)
                else:
                    self.field_type_classification_head = FieldTypeClassification(
                        num_classes=self.num_tokens, fuse_embedding_channel=self.late_fusion_fuse_embedding_channel, loss_weights=lw,
                        num_hard_positive_1=num_hard_positive_main_1, num_hard_negative_1=num_hard_negative_main_1,
                        num_hard_positive_2=num_hard_positive_main_2, num_hard_negative_2=num_hard_negative_main_2, random=ohem_random,
                        layer_mode=layer_mode, work_mode=self.work_mode)
            else:
                assert tag_to_idx is not None, "tag_to_idx cannot be None in crf mode"
                self.field_type_classification_head = CRFFieldTypeClassification(
                    tag_to_idx=tag_to_idx, fuse_embedding_channel=self.late_fusion_fuse_embedding_channel, layer_mode=layer_mode)
            # both modes use the paper's two-stage segmentation head (reference :384-396, :443-456)
            self.semantic_segmentation_head = None if inference else SemanticSegmentationClassifier(
                p_fuse_channel=self.p_fuse_channel, num_classes=self.num_tokens, loss_weights=lw, loss_1_sample_list=loss_aux_sample_list,
                num_hard_positive=num_hard_positive_aux, num_hard_negative=num_hard_negative_aux)

    # the reference flips work_mode on train()/eval() (:462-468), even for a model built in another mode
    def train(self, mode: bool = True):
        self.work_mode = "train"
        return super().train(mode)

    def eval(self):
        self.work_mode = "eval"
        return super().eval()

    # ---------------------------------------------------------------------------------------------
    def _trunk(self, image, seg_indices, coors, corpus, mask):
        batch, icoors, _ = self.transform.forward_nhwc(image, coors)
        B, H, W, _ = batch.shape
        packed = BERTgridGenerator.pack_boxes(tuple(icoors))
        return batch, icoors, packed, B, H, W

    def _features(self, batch, packed, B, H, W, seg_indices, corpus, mask):
        gen = self.BERTgrid_generator
        # the part of the CNN in front of the early fusion does not need the grid: enqueue it first, so the host-side packing of
        # the token windows / index tables of the encoder runs behind ~3 ms of device work instead of an idle device
        if ops.overlap_enabled() and batch.is_cuda and self._overlap_safe():
            # ... and the encoder goes on the side stream (vbg/ops.py): its workgroups and the CNN's fill each other's idle CUs.
            # autograd runs every backward node on the stream of its forward, so the encoder's backward overlaps the backward of
            # the CNN in front of the early fusion the same way; JoinSideFn brings the two streams together at the end of backward
            main, side = torch.cuda.current_stream(batch.device), ops.side_stream(batch.device)
            Fn.SIDE_OK[0] = True                # (until forward() returns: the graph will hold a JoinSideFn node)
            side.wait_stream(main)              # inputs / parameters written on the caller's stream so far
            seq_lo = torch.autograd._get_sequence_nr()
            pre = self.backbone.stage1(batch)
            seq_hi = torch.autograd._get_sequence_nr()
            with torch.cuda.stream(side):
                emb_cat, counts = gen._segment_embeddings(corpus, mask, seg_indices)
            self._stage1_backward_priority(pre, seq_lo, seq_hi, getattr(gen, "_layer_seq", None))
            main.wait_stream(side)
            emb_cat.record_stream(main)
            emb_cat = Fn.JoinSideFn.apply(emb_cat)
        else:
            pre = self.backbone.stage1(batch)
            emb_cat, counts = gen._segment_embeddings(corpus, mask, seg_indices)
        boxes, box_off, box_doc = packed
        assert emb_cat.shape[0] == boxes.shape[0], "number of segment embeddings and boxes mismatch"
        grid = gen._scatter((H, W), emb_cat, boxes, box_off, box_doc, B, 0)
        p_fuse = self.backbone.stage2(pre, grid)
        return emb_cat, p_fuse

    @staticmethod
    def _stage1_backward_priority(pre, seq_lo, seq_hi, layer_seq):
        """WHEN the host enqueues the backward of the CNN's first stage.  The autograd engine runs the ready node with the highest
        sequence number first; the encoder's nodes are created behind the first stage's (the forward enqueues the CNN first, so that
        the host-side packing of the token windows hides behind device work), so in backward the engine walks the WHOLE encoder -- twelve
        layers, ~5 ms of host enqueue time -- before it touches the first stage, whose 2.5 ms of device work then starts when the
        encoder's backward is nearly over instead of beside it (kernel trace of round 6: the caller's stream idle for 7 ms).  Here the
        first stage's nodes get the sequence number that sits behind the encoder's top `n` layers: those are enqueued first (they keep
        the encoder's stream fed while the host is busy elsewhere), then the first stage, then the rest of the encoder."""
        n = ops.stage1_bwd_after()
        if n < 0 or not layer_seq or seq_hi <= seq_lo:
            return
        L = len(layer_seq)
        k = max(0, L - n)                          # layers k .. L-1 first, then the first stage, then layers k-1 .. 0
        target = layer_seq[k - 1] if k > 0 else max(seq_lo - 1, 0)
        if k == L:
            return
        if n == 0:
            target = layer_seq[-1] + 1024          # ahead of the whole encoder
        seen, stack = set(), [t.grad_fn for t in pre if t is not None and t.grad_fn is not None]
        while stack:
            node = stack.pop()
            if node is None or id(node) in seen:
                continue
            seen.add(id(node))
            sq = node._sequence_nr()
            if seq_lo <= sq < seq_hi:
                if not hasattr(node, "_set_sequence_nr"):
                    return          # (a scheduling hint only: without the setter the engine's own order stands)
                node._set_sequence_nr(target)
            for nxt, _ in node.next_functions:
                if nxt is not None and id(nxt) not in seen:
                    stack.append(nxt)

    def _overlap_safe(self) -> bool:
        """May the encoder run on the side stream in this call?  Only where it pays and where nobody reads gradients behind the library's
        back.  (i) Training steps: forward-only calls lose on the second stream (round 5: 5.06 -> 5.21 ms for one document; round 6, with
        the call no longer host-bound: 3.9 -> 4.1 ms, eight documents 8.95 -> 9.18 ms, profiles/r06_infer_overlap_ab.txt).  (ii) The encoder's weight gradients are written by kernels on the side stream straight into the flat gradient views
        (autograd sees `None`): vbg.optim.FlatReducer is told per parameter and its staging stream waits for the side stream, the end of
        backward() joins the streams for whatever the caller enqueues next -- but torch's DistributedDataParallel copies `.grad` into its
        buckets from a hook on the AccumulateGrad node, which the engine runs on the caller's stream WITHOUT an event for an undefined
        gradient: it would read gradients the side stream is still writing (measured: the second step's loss of the stock DDP route off
        by 2.5e-4 instead of 1e-6, tests/test_gpu_ddp.py).  So: inside a process group, one stream unless every flat group this model
        is homed in is collected by a live FlatReducer."""
        if not torch.is_grad_enabled():
            return False
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            home = self.__dict__.get("_vbg_home_state")
            if home is None or Fn.GRAD_READY[0] is None:
                return False
            for g in home.groups:
                r = getattr(g, "_vbg_reducer", None)
                if r is None or r() is None or Fn.GRAD_READY[0] != r()._param_ready:
                    return False
        return True

    def inference(self, image: Tuple[torch.Tensor], seg_indices: Tuple[torch.Tensor], coors: torch.Tensor, corpus: torch.Tensor,
                  mask: torch.Tensor):
        ops.set_amp(torch.is_autocast_enabled("cuda"))
        self.BERTgrid_generator.prefetch_host(corpus, mask, seg_indices)
        try:
            batch, icoors, packed, B, H, W = self._trunk(image, seg_indices, coors, corpus, mask)
            emb_cat, p_fuse = self._features(batch, packed, B, H, W, seg_indices, corpus, mask)
        finally:
            self.BERTgrid_generator.release_host()
            Fn.SIDE_OK[0] = False
        roi = self.grid_roi_align_net(p_fuse, icoors, None, packed=packed)
        fuse = self.late_fusion_net(roi, emb_cat)
        return self.field_type_classification_head.inference(fuse)

    def forward(self, image: Tuple[torch.Tensor], seg_indices: Tuple[torch.Tensor], segment_classes: Tuple[torch.Tensor],
                coors: torch.Tensor, corpus: torch.Tensor, mask: torch.Tensor):
        # (the host copies of the integer inputs first: one device->host copy while the stream holds nothing of this step)
        self.BERTgrid_generator.prefetch_host(corpus, mask, seg_indices)
        try:
            with bn_tick_scope():          # the BatchNorm step counters of the whole forward advance in one launch
                return self._forward(image, seg_indices, segment_classes, coors, corpus, mask)
        finally:
            self.BERTgrid_generator.release_host()
            Fn.SIDE_OK[0] = False

    def _home(self):
        """Flat parameter / gradient storage, owned by the model (vbg.optim.home_parameters): built at the first training forward on
        the device the parameters are on by then (after `.to(device)`, `convert_sync_batchnorm`, the DDP wrapper: train_SROIE.py:202-210),
        rebuilt when something moved them (`.to()`, a fresh `.data`).  The all-pair encoder backward, the once-per-step plane / filter
        images and the in-place weight-gradient accumulation hang on this storage and on nothing the caller constructs: torch.optim.SGD /
        AdamW step the views like any other parameter."""
        home = self.__dict__.get("_vbg_home_state")
        if home is None or not home.valid():
            if home is not None:
                # something moved the parameters (`.to()`, `.half()`, a fresh `p.data`) after an optimizer / reducer of vbg.optim was built
                # over the old flat buffers: it would go on stepping buffers the model no longer reads -- training would silently stop
                # updating the weights.  Rebuild the optimizer after moving the model.
                for g in home.groups:
                    for attr in ("_vbg_optimizer", "_vbg_reducer"):
                        ref = getattr(g, attr, None)
                        if ref is not None and ref() is not None:
                            raise RuntimeError("ViBERTgridNet: the parameters left their flat storage (Module.to / .half / a new param.data) while a "
                                               f"live {type(ref()).__name__} still steps the old buffers; move the model first, then build the optimizers")
            from vbg.optim import ModelHome
            home = ModelHome(self, track_unused=self.classifier_mode == "full")
            self.__dict__["_vbg_home_state"] = home
        return home

    def _rooted(self, loss, home):
        return loss if home is None else Fn.StepRootFn.apply(loss, home)

    def _forward(self, image, seg_indices, segment_classes, coors, corpus, mask):
        # (training forward on the device: the parameters live in flat storage from here on; validation under no_grad does not need it)
        home = self._home() if (self.training and torch.is_grad_enabled() and ops.home_enabled()
                                and next(self.late_fusion_net.parameters()).is_cuda) else None
        # `amp: True`: the caller wraps this call in torch.cuda.amp.autocast (reference pipeline/train_val_utils.py:264); the
        # matrix products of this forward AND of its backward then run on the bf16 matrix cores (see vbg.ops.set_amp)
        ops.set_amp(torch.is_autocast_enabled("cuda"))
        batch, icoors, packed, B, H, W = self._trunk(image, seg_indices, coors, corpus, mask)
        seg_head, cls_head = self.semantic_segmentation_head, self.field_type_classification_head
        if self.classifier_mode != "simp":
            # two-stage heads: their selections depend on predictions, so they resolve their own plans, in the reference's order
            # (segmentation head first, then the classifier: the host RNG draws line up)
            emb_cat, p_fuse = self._features(batch, packed, B, H, W, seg_indices, corpus, mask)
            train_only = self.work_mode == "train" and self.training
            loss_aux, pred_mask, pred_ss = seg_head(p_fuse, segment_classes, icoors, materialize=not train_only)
            roi = self.grid_roi_align_net(p_fuse, icoors, None, packed=packed)
            fuse = self.late_fusion_net(roi, emb_cat)
            loss_c, gt_label, pred_label = cls_head(fuse, segment_classes)
            total_loss = self._rooted(loss_c + self.loss_control_lambda * loss_aux, home)
            if train_only:
                return total_loss
            return total_loss, pred_mask, pred_ss, gt_label, pred_label
        # label-only work of all four losses first: ONE device->host copy, host RNG draws in the reference's order
        # Round 6: with the heads' stream on (training steps, same conditions as the encoder's stream) this prologue -- owner map, label
        # raster, nine rocPRIM selections: ~0.5 ms of small launches -- no longer sits in front of the trunk on the caller's stream (where
        # the encoder's stream waited for it as well) but on the heads' stream, which has nothing else to do until P_fuse exists.
        dev = batch.device
        hs_on = (ops.heads_stream_enabled() and ops.overlap_enabled() and batch.is_cuda and self._overlap_safe()
                 # (a SyncBatchNorm communicator driven on the COMPUTE stream -- the opt-in direct RCCL one -- must see its collectives
                 #  from one stream: the RoI branch's BatchNorms would issue theirs from the heads' stream)
                 and not (Fn.SyncCtx.direct is not None and Fn.SyncCtx.active()))
        main = hs = None
        if hs_on:
            main, hs = torch.cuda.current_stream(dev), ops.side_stream(dev, "heads")
            hs.wait_stream(main)            # boxes / classes written on the caller's stream so far
        with (torch.cuda.stream(hs) if hs_on else contextlib.nullcontext()):
            classes = torch.cat([c.reshape(-1) for c in segment_classes]).int()
            pos_neg, cls_map = seg_head.make_labels(packed, classes, B, H, W)
            label_class, label_pn = cls_head.make_labels(segment_classes)
            plans = seg_head.plans(pos_neg, cls_map) + cls_head.plans(label_class, label_pn)
            pending = PendingCounts(plans)          # counts travel to the host while the trunk below is being enqueued / running
        if hs_on:
            ops.reserve_for(hs, *packed, *[t for t in segment_classes if torch.is_tensor(t)])

        emb_cat, p_fuse = self._features(batch, packed, B, H, W, seg_indices, corpus, mask)
        labels_ready = None
        with (torch.cuda.stream(hs) if hs_on else contextlib.nullcontext()):
            pending.finish()
            if hs_on:
                labels_ready = torch.cuda.Event()
                labels_ready.record(hs)
        train_only = self.work_mode == "train" and self.training
        if hs_on and Fn.SIDE_OK[0]:
            # the two heads are independent between P_fuse and the loss sum, forward and backward: the RoI / field-type branch (small
            # launches: RoIAlign, region-map convolutions, two linears, the classifier) stays on the heads' stream beside the
            # segmentation head's chip-wide convolutions.  autograd runs every node's backward on the stream of its forward and
            # synchronises the gradients that cross; what autograd does NOT know is written here: every tensor allocated under one
            # stream and read (also from saved-for-backward slots) under the other is reserved for it.
            hs.wait_stream(main)            # P_fuse, the segment embeddings
            with torch.cuda.stream(hs):
                roi = self.grid_roi_align_net(p_fuse, icoors, None, packed=packed)
                fuse = self.late_fusion_net(roi, emb_cat)
                loss_c, gt_label, pred_label = cls_head(fuse, segment_classes, prepared=(label_class, label_pn, plans[2:]))
            ops.reserve_for(hs, p_fuse, emb_cat)
            main.wait_event(labels_ready)   # (the label work only -- not the RoI branch enqueued behind it)
            seg_t = [pos_neg, cls_map] + [t for pl in plans[:2] for t in [pl.labels] + [x for c in pl.cats for x in (c.idx, c.cnt_dev, getattr(c, "elem", None))]]
            ops.reserve_for(main, *[t for t in seg_t if torch.is_tensor(t)])
            loss_aux, pred_mask, pred_ss = seg_head(p_fuse, segment_classes, icoors, prepared=(pos_neg, cls_map, plans[:2]),
                                                    materialize=not train_only)
            main.wait_stream(hs)
            ops.reserve_for(main, loss_c, gt_label, pred_label)
        else:
            if hs_on:                       # (labels on the heads' stream, but no joining node in this graph: everything else on one stream)
                main.wait_stream(hs)
                every = [pos_neg, cls_map, label_class, label_pn] + [t for pl in plans for t in [pl.labels] + [x for c in pl.cats for x in (c.idx, c.cnt_dev, getattr(c, "elem", None))]]
                ops.reserve_for(main, *[t for t in every if torch.is_tensor(t)])
            loss_aux, pred_mask, pred_ss = seg_head(p_fuse, segment_classes, icoors, prepared=(pos_neg, cls_map, plans[:2]),
                                                    materialize=not train_only)
            roi = self.grid_roi_align_net(p_fuse, icoors, None, packed=packed)
            fuse = self.late_fusion_net(roi, emb_cat)
            loss_c, gt_label, pred_label = cls_head(fuse, segment_classes, prepared=(label_class, label_pn, plans[2:]))
        total_loss = self._rooted(loss_c + self.loss_control_lambda * loss_aux, home)
        if train_only:
            return total_loss
        return total_loss, pred_mask, pred_ss, gt_label, pred_label
