"""CPU-side boundary checks: the C-ABI library loads and exports every symbol include/vbg.h declares
(no compute calls without a GPU), and the ctypes table covers exactly that set."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "vbg.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(vbg_[a-z0-9_]+)\s*\(", src))


def test_library_exports_every_declared_symbol():
    from vbg import lib as L
    declared = header_symbols()
    assert declared, "no declarations parsed"
    assert declared == set(L.SIGNATURES.keys()), declared ^ set(L.SIGNATURES.keys())
    for name in declared:
        assert hasattr(L.lib, name), name
    assert L.lib.vbg_version() == 100


def test_argument_errors_do_not_need_a_gpu():
    from vbg import lib as L
    # NULL descriptor / bad sizes are rejected before any launch
    assert L.lib.vbg_gemm(None, None) == -1
    assert L.lib.vbg_colsum(None, 0, 1, 1, None, 0, None) == -1
    assert L.lib.vbg_sgd_step(None, None, None, 0, 0.1, 0.9, 0.0, 1, 1.0, None) == 0      # n == 0 is a no-op


def test_conv3_host_logic_without_a_gpu():
    """csrc/conv3.hip entry points: shapes outside the kernels' geometry are argument errors (no launch), and the strip count of the
    weight gradient gives whole rounds of workgroups with at least 8 k-tiles per strip on the cfg2 shapes"""
    from vbg import lib as L
    f = L.lib.vbg_conv3x3
    assert f(None, None, None, None, None, 0, 1, 16, 128, 16, 128, 0, 0, None, None, None, 1, None) == -1   # null operands
    split = L.lib.vbg_conv3x3_split
    # late trunk stages (cfg2: 8 x 32 x 32 x 256, 8 x 16 x 16 x 512): >= 256 workgroups; wide maps, region maps, 64 filters: no split
    assert split(8, 32, 32, 256, 256) * 128 >= 256 and split(8, 16, 16, 512, 512) * 64 >= 256
    assert split(8, 128, 128, 256, 256) == 1 and split(8, 64, 64, 128, 128) == 1 and split(1024, 7, 7, 256, 256) == 1 and split(8, 32, 32, 64, 64) == 1
    strips = L.lib.vbg_conv3x3_wgrad_strips
    for (B, H, W, Cs, Cout, blocks) in [(8, 128, 128, 256, 256, 512), (8, 128, 128, 128, 128, 256), (8, 64, 64, 128, 128, 256),
                                        (8, 32, 32, 256, 256, 256), (8, 16, 16, 512, 512, 256)]:
        s = strips(B, H, W, Cs, Cout)
        tiles = (Cout // 128) * (Cs // 32)
        nchunks = B * H * W // 16
        assert s >= 1 and s * tiles == blocks, (B, H, W, Cs, Cout, s)
        assert nchunks // s >= 8
    assert strips(1, 4, 16, 32, 128) == 1                                                    # 4 chunks: one strip
    assert L.lib.vbg_conv3x3_wgrad(None, None, None, None, 1, 16, 16, 32, 128, 0, None, None, None) == -1
    assert L.lib.vbg_conv3x3_wflip(None, 1, 1, None, None) == -1
    # pre-split filter images (round 4): byte counts of the k-tile-ordered plane image, argument errors without a launch
    nb = L.lib.vbg_conv3x3_wprep_bytes
    assert nb(256, 256, 0, 0) == 256 * 256 * 9 * 4 and nb(64, 64, 1, 0) == 64 * 64 * 9 * 4 and nb(256, 256, 0, 64) == 256 * 256 * 9 * 4          # 4 bytes per element, like the fp32 filter
    assert nb(132, 16, 0, 0) == 2 * 9 * 1 * 64 * 128                                             # rows padded to whole filter tiles
    assert nb(128, 24, 0, 0) == 0 and nb(24, 128, 1, 0) == 0 and nb(128, 32, 0, 96) == 0                                         # reduction width must be a multiple of 16
    assert L.lib.vbg_conv3x3_wprep(None, None, 1, None) == -1 and L.lib.vbg_conv3x3_wprep(None, None, 0, None) == 0
    assert L.lib.vbg_conv3x3_pw(None, None, None, None, None, 0, 1, 16, 128, 16, 128, 0, None, None, None, 1, 0, None) == -1
    assert L.lib.vbg_conv3x3_pw_amp(None, None, None, None, None, 0, 1, 16, 128, 16, 128, 0, None, None, None, 1, 0, None) == -1          # the one-product form: same checks


def test_gemm_desc_layout_matches_header():
    import ctypes as C
    from vbg.lib import GemmDesc, ConvGeo
    # offsets are part of the ABI; recompute them with a tiny C program equivalent: natural alignment
    assert C.sizeof(ConvGeo) == 40
    assert GemmDesc.A.offset == 16 and GemmDesc.a_seg_ptr.offset % 8 == 0 and GemmDesc.grp.offset % 8 == 0


def test_gemm_desc_offsets_against_the_c_compiler(tmp_path):
    """every field of vbg_gemm_desc / vbg_conv_geo: offsetof + sizeof from gcc over include/vbg.h == the ctypes mirror"""
    import ctypes as C
    import shutil
    import subprocess
    from vbg.lib import AttnDesc, BertLayerFwdDesc, Conv3WprepEntry, GemmDesc, ConvGeo, PlaneGemmDesc, PlaneGroup, PlanesRef
    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "vbg.h"', 'int main(void) {']
    for st, cls in (("vbg_gemm_desc", GemmDesc), ("vbg_conv_geo", ConvGeo), ("vbg_plane_gemm_desc", PlaneGemmDesc), ("vbg_plane_group", PlaneGroup),
                    ("vbg_attn_desc", AttnDesc), ("vbg_conv3_wprep_entry", Conv3WprepEntry), ("vbg_planes_ref", PlanesRef),
                    ("vbg_bert_layer_fwd_desc", BertLayerFwdDesc)):
        src.append(f'printf("{st} sizeof %zu\\n", sizeof({st}));')
        for name, _ in cls._fields_:
            src.append(f'printf("{st} {name} %zu\\n", offsetof({st}, {name}));')
    src += ['return 0; }']
    cfile = tmp_path / "off.c"
    cfile.write_text("\n".join(src))
    exe = str(tmp_path / "off")
    subprocess.run(["gcc", "-I", os.path.join(root, "include"), str(cfile), "-o", exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split("\n")
    got = {(a, b): int(c) for a, b, c in (ln.split() for ln in out if ln)}
    for st, cls in (("vbg_gemm_desc", GemmDesc), ("vbg_conv_geo", ConvGeo), ("vbg_plane_gemm_desc", PlaneGemmDesc), ("vbg_plane_group", PlaneGroup),
                    ("vbg_attn_desc", AttnDesc), ("vbg_conv3_wprep_entry", Conv3WprepEntry), ("vbg_planes_ref", PlanesRef),
                    ("vbg_bert_layer_fwd_desc", BertLayerFwdDesc)):
        assert got[(st, "sizeof")] == C.sizeof(cls)
        for name, _ in cls._fields_:
            assert got[(st, name)] == getattr(cls, name).offset, (st, name)
