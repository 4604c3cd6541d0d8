// One encoder layer forward as ONE C call (round 6).
//
// transformers' BertLayer inside BertModel (reference model/BERTgrid_generator.py:134) is seven launches of this library: the stacked
// Q/K/V projection, the fused attention, the output projection, dropout + residual + LayerNorm, FFN1 (+ GELU), FFN2, dropout + residual +
// LayerNorm.  Issued from Python every launch costs 10-25 us of host time (a descriptor built field by field, a ctypes call), 165 us per
// layer, which is what bounds single-document inference and the host-bound phases of the reference's own training loop.  This entry builds
// the same seven descriptors from one layer descriptor and launches them back to back on the caller's stream: the kernels, their
// arguments and their order are exactly those of the per-launch path (vbg/functions.py BertLayerFn.forward), so results are bit-identical
// (tests/test_gpu_layer_entry.py).  No device code here.
#include "vbg_common.h"
#include "../../include/vbg.h"

namespace {
inline bool present(const vbg_planes_ref& p) { return p.buf != nullptr; }
inline void set_a(vbg_plane_gemm_desc& g, const vbg_planes_ref& p) { g.A = p.buf; g.a_plane = p.plane; g.lda = p.ld; }
inline void set_b(vbg_plane_gemm_desc& g, const vbg_planes_ref& p) { g.B = p.buf; g.b_plane = p.plane; g.ldb = p.ld; }
inline void set_cp(vbg_plane_gemm_desc& g, const vbg_planes_ref& p) { if (p.buf) { g.Cp = p.buf; g.c_plane = p.plane; g.ldp = p.ld; } }
inline void set_cq(vbg_plane_gemm_desc& g, const vbg_planes_ref& p) { if (p.buf) { g.Cq = p.buf; g.q_plane = p.plane; g.ldq = p.ld; } }
inline vbg_plane_gemm_desc nt_product(int M, int N, int K, float* C, const float* bias, int tile, int form) {
    vbg_plane_gemm_desc g{};
    g.M = M; g.N = N; g.K = (K + 31) / 32 * 32;
    g.C = C; g.ldc = C ? N : (N + 3) / 4 * 4;
    g.bias = bias; g.epi = VBG_EPI_NONE; g.alpha = 1.0f; g.splitk = 1; g.tile = tile; g.form = form;
    return g;
}
}  // namespace

extern "C" int vbg_bert_layer_fwd(const vbg_bert_layer_fwd_desc* dp, void* stream) {
    VBG_CHECK_ARG(dp != nullptr);
    const vbg_bert_layer_fwd_desc& d = *dp;
    VBG_CHECK_ARG(d.ntok > 0 && d.hidden > 0 && d.inter > 0 && d.heads > 0 && d.hidden % 32 == 0 && d.inter % 32 == 0 && d.hidden == d.heads * 64);
    VBG_CHECK_ARG(d.x && present(d.xa) && present(d.wqkv) && present(d.wo) && present(d.wi) && present(d.wo2));
    VBG_CHECK_ARG(d.bqkv && d.bo && d.bi && d.bo2 && d.g1 && d.b1 && d.g2 && d.b2);
    VBG_CHECK_ARG(present(d.pqkv) && d.ctx && d.lse && d.ao && d.x1 && d.xhat1 && d.rstd1 && d.h && d.fo && d.y && d.xhat2 && d.rstd2);
    VBG_CHECK_ARG(d.form_qkv >= 0 && d.form_qkv <= 2 && d.form_ao >= 0 && d.form_ao <= 2 && d.form_ffn >= 0 && d.form_ffn <= 2 && d.form_attn >= 0 && d.form_attn <= 2);
    // the operand each product reads must be there in the form the product runs
    VBG_CHECK_ARG(present(d.form_ao ? d.pctxq : d.pctx) && present(d.form_ffn ? d.px1q : d.px1) && present(d.form_ffn ? d.pgq : d.pg));
    VBG_CHECK_ARG((d.mask_q == nullptr) == (d.mask_k == nullptr) && (d.mask_q == nullptr || d.mask_off != nullptr));
    const int ntok = d.ntok, hid = d.hidden, inter = d.inter;
    int rc;
    // ---- Q, K, V: one product over the stacked projections; the result leaves as planes only (the attention kernels' operands)
    {
        vbg_plane_gemm_desc g = nt_product(ntok, 3 * hid, hid, nullptr, d.bqkv, d.tile_qkv, d.form_qkv);
        set_a(g, d.xa); set_b(g, d.wqkv);
        if (d.form_attn) set_cq(g, d.pqkv); else set_cp(g, d.pqkv);
        if ((rc = vbg_plane_gemm(&g, stream)) != VBG_OK) return rc;
    }
    // ---- fused attention
    {
        vbg_attn_desc a{};
        a.mode = VBG_ATTN_FWD; a.heads = d.heads; a.ntasks = d.ntasks; a.max_len = d.max_len;
        a.tasks = d.tasks; a.seq_len = d.seq_len; a.seq_row0 = d.seq_row0; a.pad_off = d.pad_off; a.ntok_pad = d.ntok_pad;
        a.qkv = d.pqkv.buf; a.qkv_plane = d.pqkv.plane; a.qkv_ld = d.pqkv.ld;
        a.out = d.ctx; a.ldo = hid; a.lse = d.lse;
        if (d.kbar) { a.kbar = d.kbar; a.ldk = hid; }
        if (present(d.pctx)) { a.out_planes = d.pctx.buf; a.op_plane = d.pctx.plane; a.op_ld = d.pctx.ld; }
        if (present(d.pctxq)) { a.out_pair = d.pctxq.buf; a.oq_plane = d.pctxq.plane; a.oq_ld = d.pctxq.ld; }
        if (d.mask_q) { a.mask_q = d.mask_q; a.mask_k = d.mask_k; a.mask_off = d.mask_off; a.keep_scale = d.keep_scale; }
        else a.keep_scale = 1.0f;
        a.scale = d.attn_scale; a.form = d.form_attn;
        if ((rc = vbg_attn(&a, stream)) != VBG_OK) return rc;
    }
    // ---- attention output projection
    {
        vbg_plane_gemm_desc g = nt_product(ntok, hid, hid, d.ao, d.bo, d.tile_ao, d.form_ao);
        set_a(g, d.form_ao ? d.pctxq : d.pctx); set_b(g, d.wo);
        if ((rc = vbg_plane_gemm(&g, stream)) != VBG_OK) return rc;
    }
    // ---- x1 = LayerNorm(dropout(ao) + x), with its planes for FFN1
    if ((rc = vbg_dropout_add_ln_fwd_planes(d.ao, d.x, ntok, hid, d.g1, d.b1, d.eps, d.drop_p, d.seed, d.stream_id0 + 1, d.x1, d.xhat1, d.rstd1,
                                            d.px1.buf, (int)d.px1.ld, d.px1.plane, d.px1q.buf, (int)d.px1q.ld, d.px1q.plane, stream)) != VBG_OK) return rc;
    // ---- FFN1: h = x1 Wi^T + bi, gelu(h) as planes only
    {
        vbg_plane_gemm_desc g = nt_product(ntok, inter, hid, d.h, d.bi, d.tile_ffn1, d.form_ffn);
        g.epi = VBG_EPI_GELU_DUAL;
        set_a(g, d.form_ffn ? d.px1q : d.px1); set_b(g, d.wi);
        set_cp(g, d.pg);
        if (d.form_ffn) set_cq(g, d.pgq);
        if ((rc = vbg_plane_gemm(&g, stream)) != VBG_OK) return rc;
    }
    // ---- FFN2
    {
        vbg_plane_gemm_desc g = nt_product(ntok, hid, inter, d.fo, d.bo2, d.tile_ffn2, d.form_ffn);
        set_a(g, d.form_ffn ? d.pgq : d.pg); set_b(g, d.wo2);
        if ((rc = vbg_plane_gemm(&g, stream)) != VBG_OK) return rc;
    }
    // ---- y = LayerNorm(dropout(fo) + x1), with its planes for the next layer's first product
    return vbg_dropout_add_ln_fwd_planes(d.fo, d.x1, ntok, hid, d.g2, d.b2, d.eps, d.drop_p, d.seed, d.stream_id0 + 2, d.y, d.xhat2, d.rstd2,
                                         d.py.buf, (int)d.py.ld, d.py.plane, d.pyq.buf, (int)d.pyq.ld, d.pyq.plane, stream);
}
