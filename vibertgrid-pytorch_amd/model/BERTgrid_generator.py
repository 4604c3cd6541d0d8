"""BERTgridGenerator — MI355X-native mirror of the reference module of the same name
(reference model/BERTgrid_generator.py:7-287): same constructor, same `BERT_embedding`,
`BERTgrid_embedding` and `forward` contracts, but

* the 510-token sliding windows (:81-146) are PACKED: only mask==1 tokens plus the [CLS]/[SEP] of each
  (document, window) are encoded, with the position ids they have inside the reference's 512-wide
  window.  Masked keys contribute exactly 0 to the softmax and masked queries are discarded by the
  reference (:137-151), so this is exact while skipping every PAD FLOP;
* the HF model object only OWNS the parameters (state_dict keys unchanged); the arithmetic runs in
  libvbg (fp32 MFMA GEMMs + fused row kernels);
* the ~2T `.item()` syncs of the aggregation loop (:148-189) and the 4S syncing slice assignments
  (:230-243) become one segmented-reduce kernel and one owner-map + gather-write pair.
"""
from typing import Callable, Tuple

import numpy as np
import torch
import torch.nn as nn

from vbg import functions as Fn
from vbg.batch import host_mirror
from vbg import ops

CLS_ID, SEP_ID = 101, 102      # hard-coded in the reference for every model family (:88-93)


class _Packed:
    """Host-built packing of one batch (numpy), uploaded once per forward."""
    pass


def pack_windows(corpus: np.ndarray, mask: np.ndarray, roberta: bool):
    """corpus int64 [B,T], mask [B,T] -> packed ids/positions + sequence table + kept-token rows.

    Sequence = one (document, window) pair: [CLS] + tokens with mask==1 + [SEP].  Position ids are the
    column indices inside the reference's window ([CLS] 0, token at window column j -> 1+j, [SEP] ->
    1+curr_len where curr_len is the BATCH-level width of the window); RoBERTa adds its padding_idx
    offset (ids are never equal to RoBERTa's pad id 1 unless the corpus contains it, handled exactly).
    Windows with curr_len == 0 (T % 510 == 0) produce no kept token and are skipped."""
    B, T = corpus.shape
    nwin = T // 510 + 1
    ids, pos, seq_len, seq_doc = [], [], [], []
    kept_rows = [[] for _ in range(B)]
    row = 0
    for c in range(nwin):
        start = c * 510
        cur = min(510, T - start)
        if cur <= 0:
            continue
        for b in range(B):
            cols = np.flatnonzero(mask[b, start:start + cur] == 1)
            wid = np.concatenate([[CLS_ID], corpus[b, start + cols], [SEP_ID]]).astype(np.int64)
            wpos = np.concatenate([[0], 1 + cols, [1 + cur]]).astype(np.int64)
            if roberta:
                # RobertaEmbeddings.create_position_ids_from_input_ids over the full 512-wide window
                full = np.zeros(512, np.int64)
                full[0] = CLS_ID
                full[1:1 + cur] = corpus[b, start:start + cur]
                full[1 + cur] = SEP_ID
                m = (full != 1).astype(np.int64)
                full_pos = np.cumsum(m) * m + 1
                wpos = full_pos[wpos]
            L = wid.shape[0]
            ids.append(wid)
            pos.append(wpos)
            seq_len.append(L)
            seq_doc.append(b)
            kept_rows[b].append(row + 1 + np.arange(cols.shape[0]))
            row += L
    pk = _Packed()
    pk.ids = np.concatenate(ids).astype(np.int32) if ids else np.zeros(0, np.int32)
    pk.pos = np.concatenate(pos).astype(np.int32) if pos else np.zeros(0, np.int32)
    pk.seq_len = np.asarray(seq_len, np.int64)
    pk.ntok = int(row)
    pk.kept_rows = [np.concatenate(r) if r else np.zeros(0, np.int64) for r in kept_rows]
    return pk


def attention_tables(seq_len: np.ndarray, heads: int, dh: int, hidden: int):
    """Group tables (int64 [G,8]) of the six grouped attention GEMMs; G = nseq*heads.
    Score blocks live in one flat buffer: block (seq, head) is [L, ld] at soff, ld = roundup(maxlen, 4)."""
    nseq = seq_len.shape[0]
    maxlen = int(seq_len.max()) if nseq else 0
    ld = (maxlen + 3) // 4 * 4
    row0 = np.concatenate([[0], np.cumsum(seq_len)[:-1]]) if nseq else np.zeros(0, np.int64)
    G = nseq * heads
    L = np.repeat(seq_len, heads)
    r0 = np.repeat(row0, heads)
    h = np.tile(np.arange(heads, dtype=np.int64), nseq)
    blk = L * ld
    soff = np.concatenate([[0], np.cumsum(blk)[:-1]]) if G else np.zeros(0, np.int64)
    off_qkv = r0 * 3 * hidden + h * dh
    off_ctx = r0 * hidden + h * dh
    dhv = np.full(G, dh, np.int64)
    z = np.zeros(G, np.int64)

    def tab(M, N, K, a, b, c):
        return np.stack([M, N, K, a, b, c, z, z], 1).astype(np.int64)

    t = dict(qk=tab(L, L, dhv, off_qkv, off_qkv, soff), pv=tab(L, dhv, L, soff, off_qkv, off_ctx),
             dp=tab(L, L, dhv, off_ctx, off_qkv, soff), dv=tab(L, dhv, L, soff, off_ctx, off_qkv),
             dq=tab(L, dhv, L, soff, off_qkv, off_qkv))
    return t, soff, int(blk.sum()), maxlen, ld


def flash_tables(seq_len: np.ndarray, heads: int):
    """Host tables of the fused attention kernels (csrc/attn.hip; layouts in include/vbg.h `vbg_attn_desc`): first token row of
    each sequence, its offset in the 32-row padded statistics buffers, the padded index of every token, the first dropout-mask word
    of each sequence, and the (sequence, 128-row block) task list, longest sequences first."""
    nseq = seq_len.shape[0]
    row0 = np.concatenate([[0], np.cumsum(seq_len)[:-1]]).astype(np.int64) if nseq else np.zeros(0, np.int64)
    lpad = (seq_len + 31) // 32 * 32
    pad_off = np.concatenate([[0], np.cumsum(lpad)[:-1]]).astype(np.int64) if nseq else np.zeros(0, np.int64)
    ntok_pad = int(lpad.sum())
    tok_pad = np.concatenate([pad_off[s] + np.arange(seq_len[s]) for s in range(nseq)]).astype(np.int64) if nseq else np.zeros(0, np.int64)
    words = heads * lpad * (lpad // 32)
    mask_off = np.concatenate([[0], np.cumsum(words)[:-1]]).astype(np.int64) if nseq else np.zeros(0, np.int64)
    order = np.argsort(-seq_len, kind="stable")
    tasks = np.asarray([(s, b) for s in order for b in range((int(seq_len[s]) + 127) // 128)], np.int64).reshape(-1, 2)
    return row0, pad_off, ntok_pad, tok_pad, mask_off, int(words.sum()), tasks


class BERTgridGenerator(nn.Module):
    """generate BERTgrid with the given OCR results (same API as the reference class)."""

    def __init__(self, bert_model: Callable = None, grid_mode: str = "mean", stride: int = 8) -> None:
        super().__init__()
        assert bert_model is not None, "no bert model given"
        assert grid_mode in ["mean", "first"], f"grid_mode should be 'mean' or 'first', {grid_mode} were given"
        self.model = bert_model
        self.grid_mode = grid_mode
        self.stride = stride
        self._step_seed = 0x5EED

    # ------------------------------------------------------------------------------------------
    def prefetch_host(self, corpus: torch.Tensor, mask: torch.Tensor, seg_indices):
        """The integer inputs the index tables are built from on the host (token ids, mask, segment indices), fetched in ONE device->host
        copy BEFORE anything of this forward is enqueued.  The reference's loop hands over device tensors (`.to(device)` per tensor,
        pipeline/train_val_utils.py:257-262), and a copy back is a stream drain wherever it stands: behind the CNN's first stage and
        behind the whole encoder, where `_encode` / `_segment_embeddings` would issue theirs, the host waits ~1.5 ms per step and then
        enqueues the rest of the forward in front of an idle device.  At the top of the forward the stream holds nothing of this step yet.
        Tensors uploaded through vbg.batch.PackedBatch carry their host copy along and need nothing.  The arrays are held for this
        forward only (keyed by tensor identity; `release_host` drops them): nothing is attached to the caller's tensors."""
        self._host = {}
        if not corpus.is_cuda:
            return
        want = [t for t in (corpus, mask) + tuple(seg_indices) if host_mirror(t) is None and t.numel()]
        if not want:
            return
        host = torch.cat([t.reshape(-1).long() for t in want]).cpu().numpy()
        o = 0
        for t in want:
            self._host[id(t)] = host[o:o + t.numel()].reshape(tuple(t.shape))
            o += t.numel()

    def release_host(self):
        self._host = {}

    def _mirror(self, t):
        m = host_mirror(t)
        return m if m is not None else self.__dict__.get("_host", {}).get(id(t))

    def _encode(self, corpus: torch.Tensor, mask: torch.Tensor):
        """-> (token states [ntok, hidden] on device, packing)"""
        cfg = self.model.config
        dev = corpus.device
        hc, hm = self._mirror(corpus), self._mirror(mask)
        if hc is None or hm is None:
            host = torch.cat([corpus.reshape(-1).long(), mask.reshape(-1).long()]).cpu().numpy()      # one D2H sync
            n = corpus.numel()
            hc, hm = host[:n].reshape(corpus.shape), host[n:].reshape(corpus.shape)
        roberta = getattr(cfg, "model_type", "bert") == "roberta"
        pk = pack_windows(np.asarray(hc, dtype=np.int64), np.asarray(hm, dtype=np.int64), roberta)
        heads, hidden = cfg.num_attention_heads, cfg.hidden_size
        dh = hidden // heads
        tabs, soff, s_elems, maxlen, ld = attention_tables(pk.seq_len, heads, dh, hidden)
        meta = Fn.AttnMeta()
        meta.ntok, meta.nseq, meta.heads, meta.dh, meta.maxlen, meta.ld, meta.s_elems = pk.ntok, len(pk.seq_len), heads, dh, maxlen, ld, s_elems
        meta.ngroups = len(pk.seq_len) * heads
        # one H2D for all index tables
        row0, pad_off, ntok_pad, tok_pad, mask_off, mask_words, tasks = flash_tables(pk.seq_len, heads)
        blob = np.concatenate([pk.ids.astype(np.int64), pk.pos.astype(np.int64), soff.astype(np.int64), pk.seq_len,
                               np.full(len(pk.seq_len), ld, np.int64)] + [tabs[k].reshape(-1) for k in ("qk", "pv", "dp", "dv", "dq")]
                              + [mask_off, row0, pad_off, tok_pad, tasks.reshape(-1)])
        d = ops.h2d(blob, dev)
        o = 0

        def take(cnt):
            nonlocal o
            t = d[o:o + cnt]
            o += cnt
            return t

        ids = take(pk.ntok).int()
        pos = take(pk.ntok).int()
        meta.soff = take(meta.ngroups).contiguous()
        meta.lens = take(meta.nseq).int()
        meta.ldp = take(meta.nseq).int()
        G8 = meta.ngroups * 8
        meta.t_qk, meta.t_pv, meta.t_dp, meta.t_dv, meta.t_dq = (take(G8).contiguous() for _ in range(5))
        meta.t_dk = meta.t_dq
        meta.mask_off = take(meta.nseq).contiguous()
        i32 = take(2 * meta.nseq + pk.ntok + tasks.size).int()          # (one conversion launch for all int32 tables)
        meta.seq_row0, meta.pad_off = i32[:meta.nseq], i32[meta.nseq:2 * meta.nseq]
        meta.tok_pad, meta.tasks = i32[2 * meta.nseq:2 * meta.nseq + pk.ntok], i32[2 * meta.nseq + pk.ntok:]
        meta.ntok_pad, meta.mask_words, meta.ntasks = ntok_pad, mask_words, int(tasks.shape[0])
        # softmax statistics / delta of every layer's fused attention (padding rows must read zero): one zero fill per step
        meta.stat_pool = torch.zeros((len(self.model.encoder.layer), 3, heads, ntok_pad), device=dev, dtype=torch.float32)

        m = self.model
        emb = m.embeddings
        p = float(cfg.hidden_dropout_prob) if self.training else 0.0
        pa = float(cfg.attention_probs_dropout_prob) if self.training else 0.0
        assert abs(p - pa) < 1e-12 or not self.training, "hidden and attention dropout rates must match"
        self._step_seed += 1
        # per step, per rank (data-parallel ranks seeded alike must not share dropout masks) and per torch seed
        rank = torch.distributed.get_rank() if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 0
        seed = self._step_seed * 0x9E3779B1 + (torch.initial_seed() & 0xFFFFFFFF) + rank * 0x85EBCA6B
        eps = float(cfg.layer_norm_eps)
        # the attention-dropout keeps of all layers in one launch (stream ids as BertLayerFn numbers them: layer * 8)
        flash_ok = maxlen > 0 and ops.flash_ok(hidden, int(cfg.intermediate_size), dh, maxlen)      # (BertLayerFn's own test)
        meta.mask_pool = (ops.attn_mask_layers(meta, pa, seed, 0, 8, len(m.encoder.layer))
                          if (pa > 0 and flash_ok and ops.mask_pool_enabled() and torch.is_grad_enabled()) else None)
        x = Fn.BertEmbedFn.apply(emb.word_embeddings.weight, emb.position_embeddings.weight, emb.token_type_embeddings.weight,
                                 emb.LayerNorm.weight, emb.LayerNorm.bias, ids, pos, eps, p, seed, 1000)
        xpl = None                       # bf16 planes of x, handed from each layer's closing LayerNorm to the next layer's first product
        self._layer_seq = []             # autograd sequence counter behind every layer's node(s): ViBERTgridNet._stage1_backward_priority
        for li, layer in enumerate(m.encoder.layer):
            a, o_, it, ou = layer.attention.self, layer.attention.output, layer.intermediate, layer.output
            x, xpl = Fn.BertLayerFn.apply(x, xpl, a.query.weight, a.query.bias, a.key.weight, a.key.bias, a.value.weight, a.value.bias,
                                     o_.dense.weight, o_.dense.bias, o_.LayerNorm.weight, o_.LayerNorm.bias,
                                     it.dense.weight, it.dense.bias, ou.dense.weight, ou.dense.bias, ou.LayerNorm.weight,
                                     ou.LayerNorm.bias, meta, eps, p, seed, li)
            self._layer_seq.append(torch.autograd._get_sequence_nr())
        return x, pk

    def BERT_embedding(self, corpus: torch.Tensor, mask: torch.Tensor, seg_indices: Tuple[torch.Tensor]):
        """-> tuple of per-document segment embeddings [S_b, hidden] (reference :55-191)."""
        cat, counts = self._segment_embeddings(corpus, mask, seg_indices)
        return tuple(torch.split(cat, counts, 0))

    def _segment_embeddings(self, corpus, mask, seg_indices):
        dev = corpus.device
        x, pk = self._encode(corpus, mask)
        B = corpus.shape[0]
        mirrors = [self._mirror(s) if s.numel() else np.zeros(0, np.int64) for s in seg_indices]
        if B and all(m is not None for m in mirrors):          # uploaded through vbg.batch.PackedBatch, or fetched by prefetch_host
            seg_host = np.concatenate([np.asarray(m, dtype=np.int64).reshape(-1) for m in mirrors])
        else:
            seg_host = torch.cat([s.reshape(-1).long() for s in seg_indices]).cpu().numpy() if B else np.zeros(0, np.int64)
        tok_rows, starts, lens, counts = [], [], [], []
        o = base = 0
        for b in range(B):
            nb = int(seg_indices[b].numel())
            s = seg_host[o:o + nb]
            o += nb
            assert pk.kept_rows[b].shape[0] == nb, "number of valid tokens and seg_indices mismatch"
            if nb:
                brk = np.flatnonzero(np.diff(s) != 0) + 1
                st = np.concatenate([[0], brk])
                ln = np.diff(np.concatenate([st, [nb]]))
            else:
                st = ln = np.zeros(0, np.int64)
            starts.append(st + base)
            lens.append(ln)
            counts.append(int(st.shape[0]))
            tok_rows.append(pk.kept_rows[b])
            base += nb
        blob = np.concatenate(tok_rows + starts + lens).astype(np.int32)
        d = ops.h2d(blob, dev)
        nt, ns = base, sum(counts)
        tok_row, run_start, run_len = d[:nt], d[nt:nt + ns], d[nt + ns:]
        mode = 0 if self.grid_mode == "mean" else 1
        cat = Fn.SegReduceFn.apply(x, tok_row, run_start, run_len, mode)
        return cat, counts

    @staticmethod
    def pack_boxes(coors: Tuple[torch.Tensor]):
        """tuple of int32 [S_b,4] device tensors -> (boxes [N,4], box_off [B+1], box_doc [N]) int32 on device."""
        dev = coors[0].device
        counts = [int(c.shape[0]) for c in coors]
        boxes = torch.cat([c.reshape(-1, 4).int() for c in coors], 0).contiguous()
        off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        doc = np.repeat(np.arange(len(counts), dtype=np.int32), counts)
        d = ops.h2d(np.concatenate([off, doc]), dev)
        return boxes, d[:len(off)], d[len(off):]

    def BERTgrid_embedding(self, image_shape: Tuple, BERT_embeddings: Tuple[torch.Tensor], coors: Tuple[torch.Tensor], layout: int = 1):
        """zeros[B, C, int(H/stride), int(W/stride)] + last-writer-wins rectangle fill (reference :193-245).
        layout 1 = the reference's NCHW tensor; 0 = NHWC for the fused backbone."""
        for e, c in zip(BERT_embeddings, coors):
            assert e.shape[0] == c.shape[0]
        boxes, box_off, box_doc = self.pack_boxes(coors)
        emb = torch.cat(list(BERT_embeddings), 0)
        return self._scatter(image_shape, emb, boxes, box_off, box_doc, len(coors), layout)

    def _scatter(self, image_shape, emb, boxes, box_off, box_doc, B, layout):
        gh, gw = int(image_shape[0] / self.stride), int(image_shape[1] / self.stride)
        owner = ops.owner_map(boxes, box_off, B, gh, gw, self.stride)
        return Fn.GridScatterFn.apply(emb.to(torch.float32), owner, boxes, box_doc, self.stride, layout)

    def forward(self, image_shape: Tuple, seg_indices: Tuple[torch.Tensor], corpus: torch.Tensor, mask: torch.Tensor,
                coor: Tuple[torch.Tensor]):
        BERT_embeddings = self.BERT_embedding(corpus=corpus, mask=mask, seg_indices=seg_indices)
        BERTgrid_embeddings = self.BERTgrid_embedding(image_shape=image_shape, BERT_embeddings=BERT_embeddings, coors=coor)
        return BERT_embeddings, BERTgrid_embeddings
