"""Known-answer tests pinning the oracle's RoIAlign (torchvision 0.14.1 `RoIAlign(7, 1/4, sampling_ratio=-1,
aligned=False)` is not installed anywhere here and the reference has no test at that boundary, so parity for this
op is pinned by hand-computable cases derived from the published definition)."""
import numpy as np
import torch

import vbg_oracle as O


def test_constant_map_gives_constant():
    f = torch.full((1, 3, 16, 20), 2.5)
    out = O.roi_align(f, [torch.tensor([[4., 8., 40., 36.], [0., 0., 80., 64.]])], 7, 0.25)
    assert out.shape == (2, 3, 7, 7) and torch.allclose(out, torch.full_like(out, 2.5))


def test_linear_ramp_is_sampled_at_bin_centres():
    # f(y, x) = 10*y + x ; bilinear interpolation and averaging are exact for an affine map, so every bin returns the
    # ramp at the bin centre: x_c = x1*s + (pw + .5)*bin_w, y_c likewise (aligned=False: no half-pixel shift)
    H, W = 32, 40
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    f = (10 * yy + xx)[None, None]
    box = torch.tensor([[8., 12., 92., 96.]])                  # *0.25 -> x 2..23, y 3..24 : 21 x 21 -> bins of 3, 3 samples/bin
    out = O.roi_align(f, [box], 7, 0.25)[0, 0]
    x1, y1, bw, bh = 2.0, 3.0, 3.0, 3.0
    exp = torch.tensor([[10 * (y1 + (ph + .5) * bh) + (x1 + (pw + .5) * bw) for pw in range(7)] for ph in range(7)])
    assert torch.allclose(out, exp, rtol=1e-5, atol=1e-4)


def test_box_fully_outside_is_zero_and_partial_clamps():
    f = torch.ones((1, 1, 8, 8))
    out = O.roi_align(f, [torch.tensor([[400., 400., 440., 440.]])], 7, 0.25)
    assert float(out.abs().max()) == 0.0                       # every sample has y > H and x > W -> 0
    # a sample exactly at -1 <= coord <= 0 is clamped to 0 (not dropped): box starting at x = -2 px (-0.5 feature px)
    out = O.roi_align(f, [torch.tensor([[-2., 0., 26., 28.]])], 7, 0.25)
    assert torch.allclose(out, torch.ones_like(out))


def test_degenerate_box_is_clamped_to_one_feature_pixel():
    # zero-area box -> roi_w = roi_h = max(0, 1) = 1 feature pixel, 1 sample per bin (ceil(1/7) = 1), bins of 1/7
    H, W = 16, 16
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    f = (100 * yy + xx)[None, None]
    out = O.roi_align(f, [torch.tensor([[20., 24., 20., 24.]])], 7, 0.25)[0, 0]     # start (5, 6) in feature coords
    exp = torch.tensor([[100 * (6 + (ph + .5) / 7) + (5 + (pw + .5) / 7) for pw in range(7)] for ph in range(7)])
    assert torch.allclose(out, exp, rtol=1e-6, atol=1e-3)


def test_adaptive_sample_count_and_batch_index():
    # 2 images; the box on image 1 must read image 1.  roi 56x28 feature px -> 8 x 4 samples per bin.
    f = torch.stack([torch.zeros(1, 64, 64), torch.full((1, 64, 64), 7.0)])
    out = O.roi_align(f, [torch.zeros((0, 4)), torch.tensor([[0., 0., 224., 112.]])], 7, 0.25)
    assert out.shape == (1, 1, 7, 7) and torch.allclose(out, torch.full_like(out, 7.0))


def test_gradient_spreads_with_bilinear_weights():
    f = torch.zeros((1, 1, 12, 12), requires_grad=True)
    out = O.roi_align(f, [torch.tensor([[8., 8., 36., 36.]])], 7, 0.25)       # 7x7 feature px -> 1 sample per bin at (2.5+i)
    out.sum().backward()
    # each of the 49 samples sits at a pixel centre + 0.5 -> weight 1/4 on 4 neighbours; total mass 49
    assert abs(float(f.grad.sum()) - 49.0) < 1e-4
    assert abs(float(f.grad[0, 0, 5, 5]) - 1.0) < 1e-5 and abs(float(f.grad[0, 0, 2, 2]) - 0.25) < 1e-5
