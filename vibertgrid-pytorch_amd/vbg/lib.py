"""ctypes binding of libvbg.so (the C-ABI declared in include/vbg.h).

The library is the product: there is NO fallback.  If `libvbg.so` is missing or a symbol is
absent the import fails loudly (build it with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C vibertgrid-pytorch_amd/csrc`).
"""
import ctypes as C
import os

import torch  # noqa: F401  -- MUST be imported before libvbg.so is dlopen'ed: pointers and streams come from torch, so
#                              both have to bind to the SAME libamdhip64 instance (torch ships its own copy)

_HERE = os.path.dirname(os.path.abspath(__file__))
# VBG_LIB_PATH: another build of the SAME library (A/B measurements of compile-time switches, tools/calls/*.sh); never a different product
LIB_PATH = os.environ.get("VBG_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "libvbg.so")

c_int, c_ll, c_f, c_d, c_vp, c_ull = C.c_int, C.c_longlong, C.c_float, C.c_double, C.c_void_p, C.c_ulonglong


class ConvGeo(C.Structure):
    _fields_ = [("Hs", c_int), ("Ws", c_int), ("Cs", c_int), ("Hr", c_int), ("Wr", c_int), ("kh", c_int), ("kw", c_int),
                ("stride", c_int), ("pad", c_int), ("dgrad", c_int)]


class GemmDesc(C.Structure):
    _fields_ = [("M", c_int), ("N", c_int), ("K", c_int),
                ("A", c_vp), ("lda", c_ll), ("a_kind", c_int), ("a_vec", c_int),
                ("a_nseg", c_int), ("a_seg_ptr", c_vp * 4), ("a_seg_kend", c_int * 4), ("a_seg_ld", c_ll * 4),
                ("a_seg_shift", c_int * 4), ("a_H", c_int), ("a_W", c_int),
                ("a_prologue", c_int), ("a_scale", c_f),
                ("B", c_vp), ("ldb", c_ll), ("b_kind", c_int), ("b_vec", c_int),
                ("geo", ConvGeo),
                ("C", c_vp), ("ldc", c_ll), ("C2", c_vp), ("bias", c_vp),
                ("epi", c_int), ("alpha", c_f), ("accumulate", c_int), ("splitk", c_int), ("tile", c_int),
                ("grp", c_vp), ("ngroups", c_int), ("grp_maxM", c_int), ("grp_maxN", c_int), ("bk", c_int),
                ("stats", c_vp), ("stats_slots", c_int), ("bf16", c_int), ("slab_stride", c_ll)]


class Conv3WprepEntry(C.Structure):          # include/vbg.h vbg_conv3_wprep_entry
    _fields_ = [("w", c_vp), ("out", c_vp), ("Cout", c_int), ("Cin", c_int), ("flip", c_int), ("bn", c_int)]


class PlaneGroup(C.Structure):
    _fields_ = [("A", c_vp), ("B", c_vp), ("C", c_vp), ("a_plane", c_ll), ("lda", c_ll), ("b_plane", c_ll), ("ldb", c_ll), ("ldc", c_ll),
                ("M", c_int), ("N", c_int), ("tiles_m", c_int), ("tiles_n", c_int), ("a_amax", c_vp)]


class PlaneGemmDesc(C.Structure):
    _fields_ = [("M", c_int), ("N", c_int), ("K", c_int),
                ("A", c_vp), ("a_plane", c_ll), ("lda", c_ll),
                ("B", c_vp), ("b_plane", c_ll), ("ldb", c_ll),
                ("C", c_vp), ("ldc", c_ll), ("C2", c_vp), ("bias", c_vp),
                ("Cp", c_vp), ("c_plane", c_ll), ("ldp", c_ll),
                ("epi", c_int), ("alpha", c_f), ("accumulate", c_int), ("splitk", c_int), ("tile", c_int), ("trans", c_int),
                ("ngroups", c_int), ("grp", PlaneGroup * 4),
                ("sk_ws", c_vp), ("sk_cnt", c_vp), ("sk_blocks", c_int), ("sk_full", c_int), ("sk_tiles_m", c_int), ("sk_tiles_n", c_int),
                ("colsum", c_vp), ("form", c_int), ("Cq", c_vp), ("q_plane", c_ll), ("ldq", c_ll), ("a_amax", c_vp), ("c_amax", c_vp),
                ("cq_ref_in", c_vp), ("cq_l1_in", c_vp), ("cq_mul", c_f), ("cq_ref_out", c_vp)]


class L1Entry(C.Structure):                  # include/vbg.h vbg_l1_entry
    _fields_ = [("w", c_vp), ("ld", c_ll), ("rows", c_int), ("cols", c_int)]


class AttnDesc(C.Structure):
    _fields_ = [("mode", c_int), ("heads", c_int), ("ntasks", c_int), ("max_len", c_int),
                ("tasks", c_vp), ("seq_len", c_vp), ("seq_row0", c_vp), ("pad_off", c_vp), ("ntok_pad", c_ll),
                ("qkv", c_vp), ("qkv_plane", c_ll), ("qkv_ld", c_ll),
                ("dO", c_vp), ("do_plane", c_ll), ("do_ld", c_ll),
                ("out", c_vp), ("ldo", c_ll), ("lse", c_vp), ("delta", c_vp), ("out_planes", c_vp), ("op_plane", c_ll), ("op_ld", c_ll), ("kbar", c_vp), ("ldk", c_ll), ("o", c_vp),
                ("mask_q", c_vp), ("mask_k", c_vp), ("mask_off", c_vp),
                ("scale", c_f), ("keep_scale", c_f), ("out_amax", c_vp), ("out_pair", c_vp), ("oq_plane", c_ll), ("oq_ld", c_ll),
                ("form", c_int), ("do_amax", c_vp)]


class PlanesRef(C.Structure):                # include/vbg.h vbg_planes_ref
    _fields_ = [("buf", c_vp), ("plane", c_ll), ("ld", c_ll)]


class BertLayerFwdDesc(C.Structure):         # include/vbg.h vbg_bert_layer_fwd_desc
    _fields_ = [("ntok", c_int), ("hidden", c_int), ("inter", c_int), ("heads", c_int),
                ("eps", c_f), ("drop_p", c_f), ("seed", c_ull), ("stream_id0", c_ull),
                ("form_qkv", c_int), ("form_attn", c_int), ("form_ao", c_int), ("form_ffn", c_int),
                ("tile_qkv", c_int), ("tile_ao", c_int), ("tile_ffn1", c_int), ("tile_ffn2", c_int),
                ("ntasks", c_int), ("max_len", c_int), ("tasks", c_vp), ("seq_len", c_vp), ("seq_row0", c_vp), ("pad_off", c_vp), ("ntok_pad", c_ll),
                ("mask_q", c_vp), ("mask_k", c_vp), ("mask_off", c_vp), ("attn_scale", c_f), ("keep_scale", c_f),
                ("x", c_vp), ("xa", PlanesRef),
                ("wqkv", PlanesRef), ("wo", PlanesRef), ("wi", PlanesRef), ("wo2", PlanesRef),
                ("bqkv", c_vp), ("bo", c_vp), ("bi", c_vp), ("bo2", c_vp), ("g1", c_vp), ("b1", c_vp), ("g2", c_vp), ("b2", c_vp),
                ("pqkv", PlanesRef),
                ("ctx", c_vp), ("lse", c_vp), ("kbar", c_vp), ("pctx", PlanesRef), ("pctxq", PlanesRef),
                ("ao", c_vp), ("x1", c_vp), ("xhat1", c_vp), ("rstd1", c_vp), ("px1", PlanesRef), ("px1q", PlanesRef),
                ("h", c_vp), ("pg", PlanesRef), ("pgq", PlanesRef),
                ("fo", c_vp), ("y", c_vp), ("xhat2", c_vp), ("rstd2", c_vp), ("py", PlanesRef), ("pyq", PlanesRef)]


ATTN_FWD, ATTN_DQ, ATTN_DKV = 0, 1, 2
OP_DENSE_K, OP_DENSE_R, OP_CONV_K, OP_CONV_R, OP_WT_R = 0, 1, 2, 3, 4
EPI_NONE, EPI_RELU, EPI_GELU_DUAL, EPI_MUL_GELU_GRAD = 0, 1, 2, 3

# name -> (restype, argtypes); must list EVERY symbol include/vbg.h declares (tests check this)
SIGNATURES = {
    "vbg_version": (c_int, []),
    "vbg_gemm": (c_int, [C.POINTER(GemmDesc), c_vp]),
    "vbg_gemm_timed": (c_int, [C.POINTER(GemmDesc), c_vp, c_vp, c_vp]),
    "vbg_timer_create": (c_int, [c_vp]),
    "vbg_timer_destroy": (c_int, [c_vp]),
    "vbg_timer_elapsed_ms": (c_int, [c_vp, c_vp, c_vp]),
    "vbg_slab_reduce": (c_int, [c_vp, c_int, c_ll, c_int, c_int, c_ll, c_vp, c_int, c_vp, c_ll, c_vp]),
    "vbg_plane_gemm": (c_int, [C.POINTER(PlaneGemmDesc), c_vp]),
    "vbg_plane_gemm_timed": (c_int, [C.POINTER(PlaneGemmDesc), c_vp, c_vp, c_vp]),
    "vbg_split_planes": (c_int, [c_vp, c_ll, c_int, c_int, c_vp, c_int, c_ll, c_int, c_vp, c_vp]),
    "vbg_col_l1_max": (c_int, [c_vp, c_int, c_int, c_vp, c_vp]),
    "vbg_split_planes_pair": (c_int, [c_vp, c_ll, c_int, c_int, c_vp, c_int, c_ll, c_vp, c_vp, c_vp]),
    "vbg_split_planes_t": (c_int, [c_vp, c_ll, c_int, c_int, c_vp, c_int, c_ll, c_vp]),
    "vbg_split_planes_t_batched": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_ll, c_vp]),
    "vbg_split_planes_pair_t_batched": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_ll, c_vp]),
    "vbg_attn": (c_int, [C.POINTER(AttnDesc), c_vp]),
    "vbg_bert_layer_fwd": (c_int, [C.POINTER(BertLayerFwdDesc), c_vp]),
    "vbg_attn_drop_thr16": (C.c_uint, [c_f]),
    "vbg_attn_mask": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_f, c_ull, c_ull, c_vp, c_vp, c_vp]),
    "vbg_attn_mask_layers": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_f, c_ull, c_ull, c_ull, c_int, c_ll, c_vp, c_vp, c_vp]),
    "vbg_colsum": (c_int, [c_vp, c_ll, c_int, c_int, c_vp, c_int, c_vp]),
    "vbg_colsum_f64": (c_int, [c_vp, c_ll, c_int, c_int, c_vp, c_int, c_vp, c_vp]),
    "vbg_conv3x3": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_int, c_vp]),
    "vbg_conv3x3_split": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "vbg_conv3x3_pw": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_int, c_int, c_vp]),
    "vbg_conv3x3_pw_amp": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_int, c_int, c_vp]),
    "vbg_conv3x3_wprep": (c_int, [c_vp, c_vp, c_int, c_vp]),
    "vbg_conv3x3_wprep_bytes": (c_ll, [c_int, c_int, c_int, c_int]),
    "vbg_amax": (c_int, [c_vp, c_ll, c_vp, c_vp]),
    "vbg_conv3x3_wflip": (c_int, [c_vp, c_int, c_int, c_vp, c_vp]),
    "vbg_conv3x3_wgrad_strips": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "vbg_conv3x3_wgrad": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp]),
    "vbg_im2col": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp]),
    "vbg_normalize_resize": (c_int, [c_vp, c_int, c_int, c_int, c_int, C.POINTER(c_f), C.POINTER(c_f), c_vp, c_int, c_int, c_int, c_vp]),
    "vbg_rescale_boxes": (c_int, [c_vp, c_int, c_f, c_f, c_vp, c_vp]),
    "vbg_embed_ln_fwd": (c_int, [c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_f, c_f, c_ull, c_ull, c_vp, c_vp, c_vp, c_vp]),
    "vbg_embed_ln_bwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_f, c_ull, c_ull, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "vbg_dropout_add_ln_fwd": (c_int, [c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_f, c_f, c_ull, c_ull, c_vp, c_vp, c_vp, c_vp]),
    "vbg_dropout_add_ln_fwd_planes": (c_int, [c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_f, c_f, c_ull, c_ull, c_vp, c_vp, c_vp, c_vp, c_int, c_ll, c_vp, c_int, c_ll, c_vp]),
    "vbg_ln_bwd_ws_rows": (c_int, [c_int]),
    "vbg_dropout_add_ln_bwd": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_f, c_ull, c_ull, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "vbg_dropout_add_ln_bwd_pair": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_f, c_ull, c_ull, c_vp, c_int, c_ll, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "vbg_dropout_add_ln_bwd_planes": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_f, c_ull, c_ull, c_vp, c_int, c_ll, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "vbg_softmax_fwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_f, c_f, c_ull, c_ull, c_vp]),
    "vbg_softmax_bwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_f, c_f, c_vp]),
    "vbg_row_softmax": (c_int, [c_vp, c_int, c_int, c_vp, c_vp]),
    "vbg_gelu_bwd": (c_int, [c_vp, c_vp, c_ll, c_vp]),
    "vbg_relu_bwd": (c_int, [c_vp, c_vp, c_ll, c_vp]),
    "vbg_add_inplace": (c_int, [c_vp, c_vp, c_ll, c_vp]),
    "vbg_scale_inplace": (c_int, [c_vp, c_ll, c_f, c_vp]),
    "vbg_seg_reduce_fwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp]),
    "vbg_seg_reduce_bwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp]),
    "vbg_owner_map": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp]),
    "vbg_grid_scatter_fwd": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp]),
    "vbg_grid_scatter_bwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp]),
    "vbg_label_raster": (c_int, [c_vp, c_vp, c_ll, c_vp, c_vp, c_vp]),
    "vbg_bn_stats": (c_int, [c_vp, c_ll, c_int, c_vp, c_vp]),
    "vbg_bn_slots": (c_int, []),
    "vbg_bn_finalize": (c_int, [c_vp, c_int, c_int, c_d, c_vp, c_int, c_f, c_f, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "vbg_bn_apply": (c_int, [c_vp, c_vp, c_ll, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp]),
    "vbg_bn_bwd_reduce": (c_int, [c_vp, c_vp, c_vp, c_ll, c_int, c_vp, c_vp, c_int, c_vp, c_vp]),
    "vbg_bn_bwd_apply": (c_int, [c_vp, c_vp, c_vp, c_ll, c_int, c_vp, c_vp, c_vp, c_vp, c_d, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "vbg_bn_param_grad": (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    "vbg_bn_apply_fold": (c_int, [c_vp, c_vp, c_ll, c_int, c_vp, c_int, c_d, c_vp, c_f, c_f, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp]),
    "vbg_bn_fold_count": (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_d, c_vp]),
    "vbg_bn_bwd_apply_fold": (c_int, [c_vp, c_vp, c_vp, c_ll, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_d, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "vbg_maxpool3x3s2_fwd": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp]),
    "vbg_maxpool3x3s2_bwd": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp]),
    "vbg_avgpool2_fwd": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp]),
    "vbg_avgpool2_bwd": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp]),
    "vbg_upsample2_add": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp]),
    "vbg_sumpool": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_int, c_vp]),
    "vbg_nchw_to_nhwc": (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_vp]),
    "vbg_nhwc_to_nchw": (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_vp]),
    "vbg_upsample_nhwc_to_nchw": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp]),
    "vbg_roi_align_fwd": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_int, c_int, c_f, c_vp, c_vp]),
    "vbg_roi_align_bwd": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_int, c_int, c_f, c_vp, c_vp]),
    "vbg_ce_fwd": (c_int, [c_vp, c_ll, c_int, c_vp, c_vp, c_ll, c_vp, c_int, c_int, c_int, c_vp, c_vp]),
    "vbg_ce_bwd": (c_int, [c_vp, c_ll, c_int, c_vp, c_vp, c_ll, c_vp, c_vp, c_f, c_int, c_int, c_int, c_vp, c_vp]),
    "vbg_compact_ws_bytes": (c_ll, [c_ll]),
    "vbg_compact": (c_int, [c_vp, c_ll, c_int, c_int, c_vp, c_vp, c_vp, c_ll, c_vp]),
    "vbg_sort_ws_bytes": (c_ll, [c_ll]),
    "vbg_sort_desc": (c_int, [c_vp, c_ll, c_vp, c_vp, c_vp, c_ll, c_vp]),
    "vbg_gather_f32": (c_int, [c_vp, c_vp, c_ll, c_vp, c_vp]),
    "vbg_gather_i32": (c_int, [c_vp, c_vp, c_ll, c_vp, c_vp]),
    "vbg_sum_f32": (c_int, [c_vp, c_ll, c_vp, c_vp]),
    "vbg_sumsq": (c_int, [c_vp, c_ll, c_vp, c_vp]),
    "vbg_gather_rows": (c_int, [c_vp, c_vp, c_ll, c_int, c_vp, c_vp]),
    "vbg_scatter_rows_add": (c_int, [c_vp, c_vp, c_ll, c_int, c_vp, c_vp]),
    "vbg_crf_nll_fwd": (c_int, [c_vp, c_vp, c_vp, c_int, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    "vbg_crf_nll_bwd": (c_int, [c_vp, c_vp, c_vp, c_int, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "vbg_crf_viterbi": (c_int, [c_vp, c_vp, c_int, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    "vbg_sgd_step": (c_int, [c_vp, c_vp, c_vp, c_ll, c_f, c_f, c_f, c_int, c_f, c_vp]),
    "vbg_adamw_step": (c_int, [c_vp, c_vp, c_vp, c_vp, c_ll, c_f, c_f, c_f, c_f, c_f, c_int, c_f, c_vp]),
}


class VbgError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP library is the product and there is no fallback. "
            "Build it with `make -C vibertgrid-pytorch_amd/csrc` (hipcc, gfx950).")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)           # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(rc: int, what: str):
    if rc != 0:
        kind = "argument error" if rc < 0 else f"hipError {rc}"
        raise VbgError(f"{what} failed: {kind}")
