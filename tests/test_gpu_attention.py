"""Fused attention kernels (csrc/attn.hip, through the C-ABI) against a plain fp64 statement of transformers' BertSelfAttention
(softmax(Q K^T / sqrt(dh)) -> dropout -> P V, model/BERTgrid_generator.py:134) and torch autograd of it, on packed variable-length
sequences.  Tolerances are written at each assert (fp32-grade: the kernels split every operand exactly into three bf16 pieces)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _meta(seq_len, heads):
    from model.BERTgrid_generator import flash_tables
    from vbg import functions as Fn
    dev = torch.device("cuda")
    sl = np.asarray(seq_len, np.int64)
    row0, pad_off, ntok_pad, tok_pad, mask_off, mask_words, tasks = flash_tables(sl, heads)
    m = Fn.AttnMeta()
    m.nseq, m.heads, m.dh, m.maxlen, m.ntok = len(sl), heads, 64, int(sl.max()), int(sl.sum())
    m.lens = torch.from_numpy(sl).int().to(dev)
    m.seq_row0 = torch.from_numpy(row0).int().to(dev)
    m.pad_off = torch.from_numpy(pad_off).int().to(dev)
    m.tok_pad = torch.from_numpy(tok_pad).int().to(dev)
    m.mask_off = torch.from_numpy(mask_off).to(dev)
    m.tasks = torch.from_numpy(tasks.reshape(-1)).int().to(dev)
    m.ntok_pad, m.mask_words, m.ntasks = ntok_pad, mask_words, int(tasks.shape[0])
    return m, row0, pad_off, mask_off


def _keep_matrix(words, off, head, L):
    """mask words [L_pad, nkb] of one (sequence, head) -> bool [L, L] (query, key)"""
    nkb = (L + 31) // 32
    w = words[off + head * nkb * 32 * nkb: off + (head + 1) * nkb * 32 * nkb].reshape(nkb * 32, nkb).astype(np.uint32)
    bits = ((w[:, :, None] >> np.arange(32, dtype=np.uint32)[None, None, :]) & 1).reshape(nkb * 32, nkb * 32)
    return bits[:L, :L].astype(bool)


def _run(seq_len, heads, p, seed=0, form=0, dscale=1.0):
    """form 0: three bf16 planes / six piece products; 1: fp16-pair planes / three piece products (dO's planes scaled by the power of two of
    its largest magnitude, like the bound-scaled planes of the encoder's backward); 2: inside an autocast region -- the hi planes alone"""
    from vbg import ops
    from vbg.lib import ATTN_DKV, ATTN_DQ, ATTN_FWD
    dev = torch.device("cuda")
    meta, row0, pad_off, mask_off = _meta(seq_len, heads)
    hid, ntok = heads * 64, meta.ntok
    g = torch.Generator().manual_seed(seed)
    # wide dynamic range across columns, like real q / k / v
    qkv = torch.randn(ntok, 3 * hid, generator=g) * torch.exp2(torch.randint(-3, 2, (3 * hid,), generator=g).float())
    dO = torch.randn(ntok, hid, generator=g) * dscale
    slot_do = None
    if form == 0:
        pq = ops.split_planes(qkv.to(dev))
        pdo = ops.split_planes(dO.to(dev))
    else:
        pq = ops.split_planes_pair(qkv.to(dev))
        slot_do = ops.amax(dO.to(dev))
        pdo = ops.split_planes_pair(dO.to(dev), amax_slot_=slot_do)
    scale = 0.125
    masks = ops.attn_mask(meta, p, 1234, 5) if p > 0 else None
    O = torch.zeros(ntok, hid, device=dev)
    lse = torch.zeros(2, heads, meta.ntok_pad, device=dev)
    kbar = torch.zeros(ntok, hid, device=dev)
    opl = ops.planes_empty(ntok, hid, dev)
    oq = ops.pair_empty(ntok, hid, dev)
    oq.buf.zero_()
    if form == 2:
        ops.set_amp(True)
    ops.attn(meta, ATTN_FWD, pq, None, O, lse, None, masks, scale, p, kbar=kbar, out_planes=opl, out_pair=oq)
    assert torch.equal(opl.buf, ops.split_planes(O).buf), "planes of O written by the forward kernel != split(O)"
    assert torch.equal(oq.buf, ops.split_planes_pair(O).buf), "fp16-pair planes of O written by the forward kernel != split_pair(O)"
    delta = torch.zeros_like(lse[0])
    dqkv = torch.full((ntok, 3 * hid), float("nan"), device=dev)
    slot = ops.amax_slot(dev)          # the largest magnitude of d(qkv) rides on the two backward kernels (fp16-pair planes' scale)
    ops.attn(meta, ATTN_DQ, pq, pdo, dqkv, lse, delta, masks, scale, p, kbar=kbar, o=O, out_amax=slot, do_amax=slot_do)
    ops.attn(meta, ATTN_DKV, pq, pdo, dqkv, lse, delta, masks, scale, p, out_amax=slot, do_amax=slot_do)
    ops.set_amp(False)
    torch.cuda.synchronize()
    assert int(slot.max().item()) == int(dqkv.abs().max().view(torch.int32).item())
    # ---- reference: fp64, per (sequence, head) --------------------------------------------------------------------
    ks = ops.attn_keep_scale(p) if p > 0 else 1.0
    mq = masks[0].cpu().numpy().view(np.uint32) if p > 0 else None
    mk = masks[1].cpu().numpy().view(np.uint32) if p > 0 else None
    x = qkv.double().requires_grad_(True)
    Oref = torch.zeros(ntok, hid, dtype=torch.float64)
    lse_ref = torch.zeros(heads, meta.ntok_pad, dtype=torch.float64)
    outs = []
    keep_frac = []
    for s, L in enumerate(seq_len):
        r0 = int(row0[s])
        for h in range(heads):
            q = x[r0:r0 + L, h * 64:(h + 1) * 64]
            k = x[r0:r0 + L, hid + h * 64:hid + (h + 1) * 64]
            v = x[r0:r0 + L, 2 * hid + h * 64:2 * hid + (h + 1) * 64]
            sc = (q @ k.t()) * scale
            pr = torch.softmax(sc, -1)
            lse_ref[h, int(pad_off[s]):int(pad_off[s]) + L] = torch.logsumexp(sc, -1).detach()
            if p > 0:
                keep = _keep_matrix(mq, int(mask_off[s]), h, L)
                keep_t = _keep_matrix(mk, int(mask_off[s]), h, L)
                assert (keep == keep_t.T).all(), "the two mask orientations disagree"
                keep_frac.append(keep.mean() if L >= 128 else None)
                pr = pr * torch.from_numpy(keep).double() * ks
            outs.append((r0, L, h, pr @ v))
    for r0, L, h, o in outs:
        Oref[r0:r0 + L, h * 64:(h + 1) * 64] = o.detach()
    loss = sum((o * dO[r0:r0 + L, h * 64:(h + 1) * 64].double()).sum() for r0, L, h, o in outs)
    loss.backward()
    return dict(O=O.cpu().double(), Oref=Oref, lse=torch.where(lse[1] > 0, lse[0] - torch.log(lse[1].clamp_min(1e-30)), torch.zeros_like(lse[0])).cpu().double(), lse_ref=lse_ref, dqkv=dqkv.cpu().double(), dref=x.grad,
                keep_frac=[f for f in keep_frac if f is not None], hid=hid)


def _relerr(a, b):
    return float((a - b).abs().max() / b.abs().max())


@pytest.mark.parametrize("seq_len,heads", [([512, 4, 130, 33, 200], 3), ([4, 2], 2), ([512] * 2 + [4] * 2, 12), ([129, 128, 127, 97], 2)])
def test_fused_attention_vs_fp64(seq_len, heads):
    r = _run(seq_len, heads, 0.0)
    assert torch.isfinite(r["O"]).all() and torch.isfinite(r["dqkv"]).all()
    # fp32-grade: errors of the size of fp32 rounding of the outputs (max error / max magnitude)
    assert _relerr(r["O"], r["Oref"]) < 2e-6, _relerr(r["O"], r["Oref"])
    assert float((r["lse"] - r["lse_ref"]).abs().max()) < 5e-6
    hid = r["hid"]
    for name, sl in (("dq", slice(0, hid)), ("dk", slice(hid, 2 * hid)), ("dv", slice(2 * hid, 3 * hid))):
        e = _relerr(r["dqkv"][:, sl], r["dref"][:, sl])
        assert e < 5e-6, (name, e)


def test_fused_attention_dropout():
    r = _run([512, 130, 4], 2, 0.1, seed=3)
    assert torch.isfinite(r["O"]).all() and torch.isfinite(r["dqkv"]).all()
    for f in r["keep_frac"]:
        assert abs(f - 0.9) < 0.01, f                      # Bernoulli(0.9) keeps over >= 16k draws
    assert _relerr(r["O"], r["Oref"]) < 2e-6
    hid = r["hid"]
    for name, sl in (("dq", slice(0, hid)), ("dk", slice(hid, 2 * hid)), ("dv", slice(2 * hid, 3 * hid))):
        e = _relerr(r["dqkv"][:, sl], r["dref"][:, sl])
        assert e < 5e-6, (name, e)


@pytest.mark.parametrize("seq_len,heads,dscale", [([512, 4, 130, 33, 200], 3, 1.0), ([4, 2], 2, 1e-7), ([512] * 2 + [4] * 2, 12, 3e-9), ([129, 128, 127, 97], 2, 2e4)])
def test_fused_attention_pair_form_vs_fp64(seq_len, heads, dscale):
    """csrc/attn.hip FORM 1 (round 5): q / k / v / dO as fp16-pair planes, three fp16 piece products per product into ONE accumulator set
    (one operand of each cross product carries the 2^-11), probabilities / score gradients split into two fp16 pieces after a power-of-two
    scaling.  Same gates as the six-product form (fp32-grade), with dO at gradient magnitudes 3e-9 ... 2e4 (its planes are scaled by the
    power of two of the slot, the kernels scale back)."""
    r = _run(seq_len, heads, 0.0, form=1, dscale=dscale)
    assert torch.isfinite(r["O"]).all() and torch.isfinite(r["dqkv"]).all()
    e = _relerr(r["O"], r["Oref"])
    print("pair form: O", e, "lse", float((r["lse"] - r["lse_ref"]).abs().max()),
          [(_relerr(r["dqkv"][:, sl], r["dref"][:, sl])) for sl in (slice(0, r["hid"]), slice(r["hid"], 2 * r["hid"]), slice(2 * r["hid"], 3 * r["hid"]))])
    assert e < 2e-6, e
    assert float((r["lse"] - r["lse_ref"]).abs().max()) < 5e-6
    hid = r["hid"]
    for name, sl in (("dq", slice(0, hid)), ("dk", slice(hid, 2 * hid)), ("dv", slice(2 * hid, 3 * hid))):
        e = _relerr(r["dqkv"][:, sl], r["dref"][:, sl])
        print("pair form:", name, e)
        assert e < 5e-6, (name, e)


def test_fused_attention_pair_form_dropout():
    r = _run([512, 130, 4], 2, 0.1, seed=3, form=1, dscale=1e-6)
    assert torch.isfinite(r["O"]).all() and torch.isfinite(r["dqkv"]).all()
    assert _relerr(r["O"], r["Oref"]) < 2e-6
    hid = r["hid"]
    for name, sl in (("dq", slice(0, hid)), ("dk", slice(hid, 2 * hid)), ("dv", slice(2 * hid, 3 * hid))):
        e = _relerr(r["dqkv"][:, sl], r["dref"][:, sl])
        assert e < 5e-6, (name, e)


def test_fused_attention_one_product_form():
    """FORM 2 (`amp`): the hi planes alone, probabilities and score gradients rounded to fp16 once -- what fp16 autocast multiplies.  Held to
    the fp64 result at reduced-precision level: 3e-3 of the largest magnitude (fp16's 2^-11 per rounded operand through two products)."""
    r = _run([512, 130, 4, 33], 3, 0.1, seed=5, form=2, dscale=1e-5)
    assert torch.isfinite(r["O"]).all() and torch.isfinite(r["dqkv"]).all()
    e = _relerr(r["O"], r["Oref"])
    print("one-product form: O", e)
    assert 1e-7 < e < 3e-3, e
    hid = r["hid"]
    for name, sl in (("dq", slice(0, hid)), ("dk", slice(hid, 2 * hid)), ("dv", slice(2 * hid, 3 * hid))):
        e = _relerr(r["dqkv"][:, sl], r["dref"][:, sl])
        print("one-product form:", name, e)
        assert e < 3e-3, (name, e)
