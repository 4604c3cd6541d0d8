#!/usr/bin/env python3
"""Achieved HBM-side bandwidth of the streaming (non-GEMM) kernels of one cfg2 bench step.

    python tools/hbm_table.py <rocprofv3 kernel_trace.csv> [steps_in_trace]

durations come from the kernel trace; ALGORITHMIC bytes per step are computed here from the cfg2 shapes (B = 8 documents of
512x512, 4128 packed tokens, 1024 segments, r34 + bert-base): what each kernel must read and write once, fp32, no re-reads.
Peak 8.0 TB/s (spec), 6.3 TB/s achievable (MI355X_MICROARCH.md)."""
import collections
import csv
import sys

B, H, W, NTOK, HID, NSEG, NCLS = 8, 512, 512, 4128, 768, 1024, 5
MB = 1e6
f4 = 4
# (rows, channels, has_residual, relu) of every BatchNorm of resnet_34_fpn_pretrained + seg head + ROI embedding at cfg2
bn = [(B * 256 * 256, 64, 0, 1)]
bn += [(B * 128 * 128, 64, i % 2, 1) for i in range(6)]
bn += [(B * 64 * 64, 128, i % 2, 1) for i in range(8)] + [(B * 64 * 64, 128, 0, 0)]
bn += [(B * 32 * 32, 256, i % 2, 1) for i in range(12)] + [(B * 32 * 32, 256, 0, 0)]
bn += [(B * 16 * 16, 512, i % 2, 1) for i in range(6)] + [(B * 16 * 16, 512, 0, 0)]
bn += [(B * 128 * 128, 256, 0, 1)] * 2 + [(NSEG * 49, 256, 0, 1)] * 2
act = lambda m, c: m * c * f4
s_elems = 96 * 514 * 516 + 96 * 6 * 8                      # attention score blocks (8 long + 8 short sequences x 12 heads)
tok = NTOK * HID * f4
bytes_per_step = {
    # (forward statistics are fused into the producing conv's epilogue except for the split-K layer4 3x3 convs)
    "bn_reduce_kernel<false>": 6 * act(B * 16 * 16, 512),
    "bn_apply_kernel": sum(act(m, c) * (2 + r) for m, c, r, a in bn),
    "bn_reduce_kernel<true>": sum(act(m, c) * (2 + a) for m, c, r, a in bn),
    "bn_bwd_apply_kernel": sum(act(m, c) * (3 + a + r) for m, c, r, a in bn),
    "softmax_fwd_kernel": 12 * 2 * s_elems * f4,
    "softmax_bwd_kernel": 12 * 3 * s_elems * f4,
    # LayerNorm forward: reads x, res; writes y, xhat and the fp16-pair planes of y (4 B / element: the A operand of the next product)
    "dropout_add_ln_fwd_kernel": 24 * (4 * tok + NTOK * HID * 4),
    # LayerNorm backward (all-pair path, round 4): reads dy, xhat; writes dres and dx as two fp16 planes (4 B / element) scaled by a bound
    "dropout_add_ln_bwd_kernel": 24 * (3 * tok + NTOK * HID * 4),
    "gelu_bwd_kernel": 12 * 3 * NTOK * 3072 * f4,
    "adamw_kernel": 28 * 108.9e6,
    "sgd_kernel": 20 * 41.8e6,
    "im2col_kernel": B * H * W * 3 * f4 + B * 256 * 256 * 148 * f4,
    "maxpool_fwd_kernel": act(B * 256 * 256, 64) + 2 * act(B * 128 * 128, 64),
    "maxpool_bwd_kernel": 2 * act(B * 128 * 128, 64) + act(B * 256 * 256, 64),
    "roi_align_fwd_kernel": NSEG * 49 * 256 * f4 + act(B * 128 * 128, 256),
    "roi_align_bwd_sep_kernel": NSEG * 49 * 256 * f4 + act(B * 128 * 128, 256),
    "roi_align_bwd_kernel": NSEG * 49 * 256 * f4 + act(B * 128 * 128, 256),
    "grid_scatter_nhwc_kernel": act(B * 64 * 64, HID) + NSEG * HID * f4,
    "grid_scatter_bwd_kernel": act(B * 64 * 64, HID) + NSEG * HID * f4,
    "upsample2_add_kernel": sum(act(B * s * s, 256) * 2 + act(B * s * s // 4, 256) for s in (32, 64, 128)),
    "normalize_resize_kernel": 2 * B * H * W * 3 * f4,
    "seg_reduce_fwd_kernel": 4096 * HID * f4 + NSEG * HID * f4,
    "seg_reduce_bwd_kernel": 4096 * HID * f4 + NSEG * HID * f4,
    # gradient operands that still take a split pass of their own (end of round 4): d(qkv) [ntok, 2304] per layer, the encoder input and
    # one more [ntok, 768] tensor per step -- fp32 in, two fp16 planes out (4 + 4 B / element), column sums riding along
    "split_planes_pair_kernel": 12 * NTOK * 2304 * 8 + 2 * NTOK * HID * 8,
    # fused attention: q / k / v planes (6 B / element) in, O + planes + Kbar out (forward); planes of q, k, v, dO in, dq (dk, dv) out;
    # K / V (Q / dO) are re-streamed once per 128-row block of the other side: 4 blocks at L = 512 (algorithmic = one pass)
    "attn_kernel<0": 12 * (NTOK * 2304 * 6 + NTOK * HID * (4 + 6 + 4)),
    "attn_kernel<1": 12 * (NTOK * 2304 * 6 + NTOK * HID * (6 + 4 + 4)),
    "attn_kernel<2": 12 * (NTOK * 2304 * 6 + NTOK * HID * (6 + 8)),
    "attn_mask_kernel": 12 * 2 * 12 * 8 * 512 * 16 * 4,
}


def main():
    path = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 13
    t = collections.defaultdict(float)
    n = collections.defaultdict(int)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        for k in bytes_per_step:
            if k in name:
                t[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                n[k] += 1
                break
    print(f"{'kernel':32s} {'launches':>8s} {'MB/step':>10s} {'us/step':>9s} {'TB/s':>6s} {'of 6.3':>7s} {'of 8.0':>7s}")
    for k, by in sorted(bytes_per_step.items(), key=lambda kv: -t.get(kv[0], 0)):
        if n[k] == 0:
            continue
        us = t[k] / steps
        tb = by / us / 1e6
        # (a row above the physical peak means the byte model no longer describes the launches of that name: flag it, never print it as evidence)
        flag = "   <-- byte model does not match this build's launches" if tb > 8.0 else ""
        print(f"{k:32s} {n[k] / steps:8.1f} {by / MB:10.1f} {us:9.1f} {tb:6.2f} {tb / 6.3:7.2f} {tb / 8.0:7.2f}{flag}")


if __name__ == "__main__":
    main()
