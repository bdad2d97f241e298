"""Kernel-level parity through the C ABI (include/dcvc_hip.h) on a real MI355X.

Floating-point kernels are compared with a plain PyTorch fp32 CPU reference of the same op
(F.conv2d, F.pixel_shuffle, ...), resampling kernels additionally with fixtures produced by
the reference itself (tests/golden/warp.npz), the integer-producing kernels with the oracle.
Tolerance for fp32 convolutions: 2e-5 of the output's max magnitude (different summation
order only; the MFMA path is an exact fmaf chain)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import dcvc_ref as R
from tests.util import golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from vcm_ts_amd.engine import Engine

    return Engine("cuda:0")


def to_view(eng, name, x, cs=None):
    N, C, H, W = x.shape
    return eng.from_nchw(x.cuda(), eng.buf(name, N, H, W, C, cs=cs))


def rel_err(got, want):
    return ((got.cpu() - want).abs().max() / (want.abs().max() + 1e-20)).item()


CONV_CASES = [
    # cin segments, cout, ks, stride, H, W, ps, in_slope, out_slope, res, gate, res2
    ((64,), 64, 3, 1, 40, 72, False, None, 0.01, True, False, False),
    ((64,), 64, 3, 1, 33, 31, False, 0.01, None, True, False, True),
    ((3, 64), 64, 3, 2, 32, 64, False, None, None, False, False, False),
    ((8,), 32, 7, 1, 17, 45, False, None, 0.0, False, False, False),
    ((32,), 64, 7, 1, 16, 40, False, None, 0.0, False, False, False),
    ((16,), 2, 7, 1, 24, 33, False, None, None, True, False, False),
    ((2,), 64, 3, 2, 64, 64, False, None, 0.01, False, False, False),
    ((2,), 64, 1, 2, 64, 64, False, None, None, False, False, False),
    ((64,), 256, 1, 1, 16, 24, True, None, 0.01, False, False, False),
    ((96,), 256, 3, 1, 8, 12, True, None, None, False, False, False),
    ((64,), 8, 1, 1, 20, 20, True, None, None, False, False, False),
    ((192, 192, 96), 384, 3, 1, 4, 8, False, None, 0.2, False, False, False),
    ((128, 64), 192, 3, 1, 17, 30, False, None, 0.2, False, False, False),
    ((64, 64), 64, 1, 1, 16, 48, False, None, None, True, True, False),
    ((144,), 576, 1, 1, 8, 8, True, None, 0.01, False, False, False),
    ((64,), 3, 3, 1, 32, 96, False, None, "clamp01", False, False, False),
    ((96,), 144, 3, 1, 16, 16, False, None, 0.01, False, False, False),
]


@pytest.fixture(scope="module")
def eng_split():
    from vcm_ts_amd.engine import Engine

    return Engine("cuda:0", precision="fp16x3")


@pytest.mark.parametrize("case", CONV_CASES, ids=[f"c{i}" for i in range(len(CONV_CASES))])
def test_split_fp16_conv_matches_fp64(eng_split, case):
    """fast mode (fp16 hi/lo operands, 3 MFMAs, fp32 accumulate) against an fp64 reference:
    error bound 3e-6 of the output scale (fp32 mode: ~2e-7), incl. operands far below fp16's
    normal range (handled by the pre-scales + MFMA subnormal support)."""
    segs, cout, ks, stride, H, W, ps, in_slope, out_slope, use_res, use_gate, use_res2 = case
    if use_res or use_res2:
        pytest.skip("epilogue identical to the fp32 kernel; covered there")
    eng = eng_split
    g = torch.Generator().manual_seed(CONV_CASES.index(case) + 7)
    N = 1
    cin = sum(segs)
    mag = torch.tensor([1e-4, 1.0, 30.0])[torch.randint(0, 3, (N, cin, 1, 1), generator=g)]
    x = torch.randn(N, cin, H, W, generator=g) * mag
    w = torch.randn(cout, cin, ks, ks, generator=g) / math.sqrt(cin * ks * ks)
    b = torch.randn(cout, generator=g) * 0.1
    xin = x.double() if in_slope is None else F.leaky_relu(x.double(), in_slope)
    want = F.conv2d(xin, w.double(), b.double(), stride=stride, padding=ks // 2)
    scale = F.conv2d(xin.abs(), w.double().abs(), None, stride=stride, padding=ks // 2).max().item()
    pk = eng.pack(("t64", case), torch.nn.Parameter(w), torch.nn.Parameter(b), segs, False)
    views, c0 = [], 0
    for i, c in enumerate(segs):
        views.append(to_view(eng, f"t64/in{i}", x[:, c0 : c0 + c]))
        c0 += c
    out = eng.buf("t64/out", N, want.shape[2], want.shape[3], cout)
    eng.conv(pk, views, out, stride=stride, in_slope=in_slope)
    got = eng.to_nchw(out).cpu().double()
    # 3e-6 of the accumulated magnitude + the fp32 rounding of the stored result itself
    assert (got - want).abs().max().item() < 3e-6 * scale + 2e-7 * want.abs().max().item()


PAIR_CASES = [
    # cin, cout, H, W, N, in_slope, out_slope, residual
    (8, 32, 40, 72, 1, None, 0.0, False),     # SpyNet MEBasic conv1 (flow_estimation.py): partial tiles both ways
    (8, 32, 17, 45, 2, None, 0.0, False),     # two images, ragged
    (6, 64, 33, 31, 1, 0.1, None, True),      # channel tail (6 of 8), 64-column variant, activation on load, residual
    (3, 48, 8, 32, 1, None, 0.01, False),     # 3 channels, Cout padded to 64, exactly one tile
]


@pytest.mark.parametrize("case", PAIR_CASES, ids=[f"p{i}" for i in range(len(PAIR_CASES))])
def test_tap_paired_7x7_conv_matches_fp64_and_the_unpaired_kernel(eng_split, case):
    """dcvc_conv2d with pair_taps (7x7, <= 8 input channels: two taps per 16-deep K step, weights from
    dcvc_conv_pack_weights_paired) against an fp64 reference within the split-fp16 bound, and against the unpaired
    kernel on the same layer (same operand values, another grouping of the sum)."""
    cin, cout, H, W, N, in_slope, out_slope, use_res = case
    eng = eng_split
    g = torch.Generator().manual_seed(PAIR_CASES.index(case) + 400)
    mag = torch.tensor([1e-3, 1.0, 20.0])[torch.randint(0, 3, (N, cin, 1, 1), generator=g)]
    x = torch.randn(N, cin, H, W, generator=g) * mag
    w = torch.randn(cout, cin, 7, 7, generator=g) / math.sqrt(cin * 49)
    b = torch.randn(cout, generator=g) * 0.1
    xin = x.double() if in_slope is None else F.leaky_relu(x.double(), in_slope)
    want = F.conv2d(xin, w.double(), b.double(), padding=3)
    scale = F.conv2d(xin.abs(), w.double().abs(), None, padding=3).max().item()
    if out_slope is not None:
        want = F.leaky_relu(want, out_slope)
    res = None
    if use_res:
        r = torch.randn(want.shape, generator=g)
        res = to_view(eng, "pair/res", r)
        want = want + r.double()
    pk = eng.pack(("pair", case), torch.nn.Parameter(w.cuda()), torch.nn.Parameter(b.cuda()), (cin,), False)
    xv = to_view(eng, "pair/in", x)
    outs = {}
    try:
        for paired in (True, False):
            eng.use_pairs = paired
            assert eng.pair_capable(pk, 1) == paired
            out = eng.buf(f"pair/out{int(paired)}", N, H, W, cout)
            out.base.fill_(float("nan"))
            for rep in range(2):  # twice: same bits
                eng.conv(pk, [xv], out, in_slope=in_slope, out_slope=out_slope, res=res)
                got = eng.to_nchw(out)
                if rep:
                    assert torch.equal(got, outs[paired])
                outs[paired] = got
    finally:
        eng.use_pairs = True
    o1, o0 = outs[True].cpu().double(), outs[False].cpu().double()
    assert not torch.isnan(o1).any()
    assert (o1 - want).abs().max().item() < 3e-6 * scale + 2e-7 * want.abs().max().item()
    assert (o1 - o0).abs().max().item() < 2e-6 * scale + 2e-7 * want.abs().max().item()


SMALL_CASES = [
    # cin segments, cout, ks, H, W, N, in_slope, out_slope, residual
    ((32,), 16, 7, 40, 72, 1, None, 0.0, False),       # SpyNet conv4
    ((16,), 2, 7, 33, 45, 2, None, None, True),        # SpyNet conv5 + flow residual, ragged tiles, two images
    ((64,), 3, 3, 32, 96, 1, None, "clamp01", False),  # recon_conv
    ((24,), 8, 3, 17, 31, 1, 0.1, 0.01, True),         # chunk tail (24 = 16 + 8), activation on load
    ((16, 32), 12, 7, 9, 33, 1, None, 0.2, False),     # two segments
    ((3,), 16, 3, 64, 64, 1, None, None, False),       # 3-channel picture input
]


@pytest.mark.parametrize("case", SMALL_CASES, ids=[f"m{i}" for i in range(len(SMALL_CASES))])
def test_small_cout_conv_matches_fp64_and_the_32_column_kernel(eng_split, case):
    """dcvc_conv2d_small (<= 16 output channels: 16x16x32 MFMA whose K carries the hi/lo split, conv_small.hip)
    against an fp64 reference within the split-fp16 error bound of test_split_fp16_conv_matches_fp64, and against
    conv_mfma on the same layer (same operand values, other accumulation order: fp32 rounding apart)."""
    segs, cout, ks, H, W, N, in_slope, out_slope, use_res = case
    eng = eng_split
    g = torch.Generator().manual_seed(SMALL_CASES.index(case) + 40)
    cin = sum(segs)
    mag = torch.tensor([1e-3, 1.0, 20.0])[torch.randint(0, 3, (N, cin, 1, 1), generator=g)]
    x = torch.randn(N, cin, H, W, generator=g) * mag
    w = torch.randn(cout, cin, ks, ks, generator=g) / math.sqrt(cin * ks * ks)
    b = torch.randn(cout, generator=g) * 0.1
    xin = x.double() if in_slope is None else F.leaky_relu(x.double(), in_slope)
    want = F.conv2d(xin, w.double(), b.double(), padding=ks // 2)
    scale = F.conv2d(xin.abs(), w.double().abs(), None, padding=ks // 2).max().item()
    if out_slope == "clamp01":
        want = want.clamp(0, 1)
    elif out_slope is not None:
        want = F.leaky_relu(want, out_slope)
    res = None
    if use_res:
        r = torch.randn(want.shape, generator=g)
        want = want + r.double()
        res = to_view(eng, "sm/res", r)
    pk = eng.pack(("small", case), torch.nn.Parameter(w.cuda()), torch.nn.Parameter(b.cuda()), segs, False)
    views, c0 = [], 0
    for i, c in enumerate(segs):
        views.append(to_view(eng, f"sm/in{i}", x[:, c0 : c0 + c]))
        c0 += c
    outs = {}
    for small in (True, False):
        eng.use_small = small
        out = eng.buf(f"sm/out{int(small)}", N, H, W, cout)
        out.base.fill_(float("nan"))
        assert eng.small_capable(pk, 1, None, None, None) == small
        eng.conv(pk, views, out, in_slope=in_slope, out_slope=out_slope, res=res)
        outs[small] = eng.to_nchw(out).cpu().double()
    eng.use_small = True
    assert (outs[True] - want).abs().max().item() < 3e-6 * scale + 2e-7 * want.abs().max().item()
    assert (outs[True] - outs[False]).abs().max().item() < 2e-6 * scale + 2e-7 * want.abs().max().item()


K32_CASES = [
    # cin segments, cout, H, W, N, ps, in_slope, out_slope, res, gate, res2, cs of the inputs (None: dense)
    ((64,), 64, 40, 72, 1, False, None, 0.01, True, False, False, None),      # ResBlock conv2 shape, partial tiles both ways
    ((64,), 64, 33, 95, 2, False, 0.01, None, True, False, True, 128),        # res + res2, activation on load, slices of wider buffers
    ((32, 64), 64, 48, 64, 1, False, None, None, False, False, False, None),  # two segments (recon first_conv)
    ((128,), 128, 34, 66, 1, False, 0.0, None, True, False, False, None),     # two output-channel blocks (context_refine)
    ((128,), 256, 18, 34, 2, True, None, 0.01, False, False, False, None),    # PixelShuffle
    ((64,), 32, 16, 32, 1, False, None, 0.01, False, False, False, None),     # 32-column variant (BN = 32)
    ((32,), 32, 50, 50, 1, False, None, None, True, True, False, None),       # gated residual (SE block), one chunk
    ((96,), 144, 17, 33, 1, False, None, "clamp01", False, False, False, None),  # Cout padded to 160, clamp
    ((192, 192, 96), 384, 9, 20, 1, False, None, 0.2, False, False, False, None),  # three segments, 15 chunks
    ((64, 64), 64, 8, 32, 1, False, None, None, False, False, False, None),   # exactly one tile
    # 1x1 layers (a 13th field: kernel size)
    ((64,), 64, 40, 72, 1, False, None, None, False, False, False, None, 1),    # feature_adaptor_P
    ((32, 32), 64, 33, 50, 2, False, None, None, True, True, False, None, 1),   # SE block's up_dim: gated residual, two segments
    ((64,), 32, 16, 40, 1, False, None, None, True, True, False, None, 1),      # 32-column variant
    ((128,), 256, 18, 34, 1, True, None, None, False, False, False, None, 1),   # PixelShuffle upsampler
    ((64,), 256, 17, 30, 1, True, None, 0.01, False, False, False, None, 1),    # subpel_conv + LeakyReLU
]


@pytest.mark.parametrize("case", K32_CASES, ids=[f"k{i}" for i in range(len(K32_CASES))])
def test_k32_conv_matches_fp64_and_the_32x32_kernel(eng_split, case):
    """dcvc_conv2d_k32 (v_mfma_f32_16x16x32_f16, 32-channel chunks, conv_k32.hip) against an fp64 reference within the
    split-fp16 bound of test_split_fp16_conv_matches_fp64, and against conv_mfma (32x32x16) on the same layer: same
    operand values, another grouping of the sum -- fp32 rounding apart.  Both kernels are deterministic."""
    segs, cout, H, W, N, ps, in_slope, out_slope, use_res, use_gate, use_res2, in_cs = case[:12]
    ks = case[12] if len(case) > 12 else 3
    eng = eng_split
    eng.k32_sizes, eng.k32_everywhere = (1, 3), True  # (the engine routes only large 3x3 layers to it; the kernel covers more)
    g = torch.Generator().manual_seed(K32_CASES.index(case) + 90)
    cin = sum(segs)
    mag = torch.tensor([1e-3, 1.0, 20.0])[torch.randint(0, 3, (N, cin, 1, 1), generator=g)]
    x = torch.randn(N, cin, H, W, generator=g) * mag
    w = torch.randn(cout, cin, ks, ks, generator=g) / math.sqrt(cin * ks * ks)
    b = torch.randn(cout, generator=g) * 0.1
    xin = x.double() if in_slope is None else F.leaky_relu(x.double(), in_slope)
    want = F.conv2d(xin, w.double(), b.double(), padding=ks // 2)
    scale = F.conv2d(xin.abs(), w.double().abs(), None, padding=ks // 2).max().item()
    if out_slope == "clamp01":
        want = want.clamp(0, 1)
    elif out_slope is not None:
        want = F.leaky_relu(want, out_slope)
    if ps:
        want = F.pixel_shuffle(want, 2)
    res = res2 = gate = None
    if use_res:
        r = torch.randn(want.shape, generator=g)
        res = to_view(eng, "k32/res", r)
        if use_gate:
            gt = torch.rand(N, want.shape[1], generator=g)
            gate = gt.cuda().contiguous()
            r = r * gt[:, :, None, None]
        want = want + r.double()
    if use_res2:
        r2 = torch.randn(want.shape, generator=g)
        res2 = to_view(eng, "k32/res2", r2)
        want = want + r2.double()
    pk = eng.pack(("k32", case), torch.nn.Parameter(w.cuda()), torch.nn.Parameter(b.cuda()), segs, ps)
    views, c0 = [], 0
    for i, c in enumerate(segs):
        v = to_view(eng, f"k32/in{i}", x[:, c0 : c0 + c], cs=in_cs)
        views.append(v)
        c0 += c
    outs = {}
    try:
        for k32 in (True, False):
            eng.use_k32 = k32
            out = eng.buf(f"k32/out{int(k32)}", N, want.shape[2], want.shape[3], want.shape[1])
            out.base.fill_(float("nan"))
            assert eng.k32_capable(pk, 1, out, res, res2, gate) == k32
            for rep in range(2):  # twice: same bits
                eng.conv(pk, views, out, in_slope=in_slope, out_slope=out_slope, res=res, gate=gate, res2=res2)
                got = eng.to_nchw(out)
                if rep:
                    assert torch.equal(got, outs[k32])
                outs[k32] = got
    finally:
        eng.use_k32 = True
        eng.k32_sizes, eng.k32_everywhere = (3,), False
    assert eng.read_status() == 0
    o1, o0 = outs[True].cpu().double(), outs[False].cpu().double()
    assert not torch.isnan(o1).any()
    assert (o1 - want).abs().max().item() < 3e-6 * scale + 2e-7 * want.abs().max().item()
    assert (o1 - o0).abs().max().item() < 2e-6 * scale + 2e-7 * want.abs().max().item()


def test_k32_conv_writes_nothing_outside_its_output(eng_split):
    """conv_k32's epilogue masks lanes (pixels beyond the picture in partial tiles, channels beyond Cout in the padded last
    block) by giving their buffer stores an out-of-range offset, which the hardware drops, and reads out-of-picture patch
    slots the same way.  Output and residual live inside larger sentinel-filled allocations here: every float before and
    after the N*H*W*cs region must still be the sentinel, and the inside must equal conv_mfma's result on a dense tensor
    within the usual bound."""
    from vcm_ts_amd.engine import View

    eng = eng_split
    N, cin, cout, H, W = 2, 64, 96, 21, 45  # partial tiles both ways; Cout padded to 128: 32 masked channels in block 1
    g = torch.Generator().manual_seed(321)
    x = torch.randn(N, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    b = torch.randn(cout, generator=g) * 0.1
    r = torch.randn(N, cout, H, W, generator=g)
    pk = eng.pack(("k32guard",), torch.nn.Parameter(w.cuda()), torch.nn.Parameter(b.cuda()), (cin,), False)
    xin = to_view(eng, "k32g/in", x)
    guard, sentinel = 4096, -12345.0
    n_out = N * H * W * cout

    def framed(fill):
        big = torch.full((guard + n_out + guard,), sentinel, device="cuda")
        if fill is not None:
            big[guard : guard + n_out] = fill.permute(0, 2, 3, 1).reshape(-1).cuda()
        return big, View(big, cout, 0, geom=(N, H, W, cout, big.data_ptr() + 4 * guard))

    res_big, res = framed(r)
    out_big, out = framed(None)
    ref = eng.buf("k32g/ref", N, H, W, cout)
    eng.k32_sizes, eng.k32_everywhere = (3,), True
    try:
        assert eng.k32_capable(pk, 1, out, res, None, None)
        eng.conv(pk, [xin], out, out_slope=0.01, res=res)
        eng.use_k32 = False
        eng.conv(pk, [xin], ref, out_slope=0.01, res=to_view(eng, "k32g/res", r))
    finally:
        eng.use_k32, eng.k32_everywhere = True, False
    torch.cuda.synchronize()
    for big in (out_big, res_big):
        assert (big[:guard] == sentinel).all() and (big[guard + n_out :] == sentinel).all()
    got = out_big[guard : guard + n_out].reshape(N, H, W, cout).permute(0, 3, 1, 2)
    assert not (got == sentinel).any()
    want = eng.to_nchw(ref)
    assert (got - want).abs().max().item() < 1e-4 * want.abs().max().item()
    assert eng.read_status() == 0


def test_k32_conv_flags_outputs_beyond_the_split_fp16_range(eng_split):
    """The always-on range guard of the k32 kernel: an output beyond +-8188 sets DCVC_STATUS_ACT_SATURATED."""
    eng = eng_split
    x = torch.full((1, 32, 16, 32), 100.0)
    w = torch.full((32, 32, 3, 3), 1.0)
    pk = eng.pack(("k32sat",), torch.nn.Parameter(w.cuda()), torch.nn.Parameter(torch.zeros(32).cuda()), (32,), False)
    out = eng.buf("k32/sat", 1, 16, 32, 32)
    assert eng.read_status() == 0
    eng.k32_everywhere = True
    try:
        assert eng.k32_capable(pk, 1, out, None, None, None)
        eng.conv(pk, [to_view(eng, "k32/satin", x)], out)
    finally:
        eng.k32_everywhere = False
    assert eng.read_status() & 1
    assert eng.read_status() == 0


@pytest.mark.parametrize("case", CONV_CASES, ids=[f"c{i}" for i in range(len(CONV_CASES))])
def test_conv_matches_torch(eng, case):
    segs, cout, ks, stride, H, W, ps, in_slope, out_slope, use_res, use_gate, use_res2 = case
    g = torch.Generator().manual_seed(CONV_CASES.index(case))
    N = 2
    xs = [torch.randn(N, c, H, W, generator=g) for c in segs]
    cin = sum(segs)
    w = torch.randn(cout, cin, ks, ks, generator=g) / math.sqrt(cin * ks * ks)
    b = torch.randn(cout, generator=g) * 0.1
    x = torch.cat(xs, 1)
    xin = x if in_slope is None else F.leaky_relu(x, in_slope)
    want = F.conv2d(xin, w, b, stride=stride, padding=ks // 2)
    if out_slope == "clamp01":
        want = want.clamp(0, 1)
    elif out_slope is not None:
        want = F.leaky_relu(want, out_slope) if out_slope > 0 else F.relu(want)
    if ps:
        want = F.pixel_shuffle(want, 2)
    res = gate = res2 = None
    if use_res:
        res = torch.randn(want.shape, generator=g)
        r = res
        if use_gate:
            gate = torch.rand(N, want.shape[1], generator=g)
            r = res * gate[:, :, None, None]
        want = want + r
    if use_res2:
        res2 = torch.randn(want.shape, generator=g)
        want = res2 + want
    wp, bp = torch.nn.Parameter(w), torch.nn.Parameter(b)
    pk = eng.pack(("t", case), wp, bp, segs, ps)
    views = []
    for i, (xx, c) in enumerate(zip(xs, segs)):
        wide = eng.buf(f"t/in{i}", N, H, W, c + 8)  # odd channel stride + offset slice
        views.append(eng.from_nchw(xx.cuda(), wide.slice(4, c)))
    out = eng.buf("t/out", N, want.shape[2], want.shape[3], want.shape[1] + 4).slice(4, want.shape[1])
    rv = to_view(eng, "t/res", res) if res is not None else None
    r2 = to_view(eng, "t/res2", res2) if res2 is not None else None
    gt = gate.cuda().contiguous() if gate is not None else None
    eng.conv(pk, views, out, stride=stride, in_slope=in_slope, out_slope=out_slope, res=rv, gate=gt, res2=r2)
    got = eng.to_nchw(out)
    assert rel_err(got, want) < 2e-5


def _sweep_cases(n, seed):
    """Seeded random layer signatures: ragged pictures (down to 1 pixel), channel counts that are not multiples of
    4 / 16 / 32, up to three input segments, every epilogue combination the C ABI accepts."""
    rng = np.random.RandomState(seed)
    out = []
    while len(out) < n:
        ks = int(rng.choice([1, 3, 3, 3, 7]))
        stride = int(rng.choice([1, 1, 2])) if ks != 7 else 1
        nseg = int(rng.choice([1, 1, 2, 3]))
        segs = tuple(int(rng.choice([1, 2, 3, 5, 8, 12, 16, 24, 31, 32, 48, 64, 96, 130])) for _ in range(nseg))
        ps = bool(rng.rand() < 0.2)
        cout = int(rng.choice([1, 2, 3, 4, 8, 12, 16, 17, 32, 40, 64, 96, 128, 200]))
        if ps:
            cout = max(4, cout // 4 * 4)
        if ks == 7 and sum(segs) * cout > 64 * 64:
            continue  # keep the CPU reference quick
        H, W = int(rng.randint(1, 70)), int(rng.randint(1, 70))
        use_res = bool(rng.rand() < 0.5)
        out.append((segs, cout, ks, stride, H, W, ps, [None, 0.1][rng.randint(2)],
                    [None, 0.0, 0.01, "clamp01"][rng.randint(4)], use_res, use_res and bool(rng.rand() < 0.3),
                    bool(rng.rand() < 0.25), int(rng.randint(1, 4))))
    return out


@pytest.mark.parametrize("precision", ["fp32", "fp16x3"])
def test_conv_random_signature_sweep(eng, eng_split, precision):
    """48 seeded random signatures per precision through Engine.conv (so the small-Cout and the 32/64-column kernels
    are chosen as in the product) against an fp64 F.conv2d: fp32 mode within 1e-6, fast mode within 3e-6 of the
    accumulated magnitude (the bounds of the fixed cases above)."""
    e = eng if precision == "fp32" else eng_split
    tol = 1e-6 if precision == "fp32" else 3e-6
    for k, case in enumerate(_sweep_cases(48, 2024)):
        segs, cout, ks, stride, H, W, ps, in_slope, out_slope, use_res, use_gate, use_res2, N = case
        g = torch.Generator().manual_seed(1000 + k)
        cin = sum(segs)
        xs = [torch.randn(N, c, H, W, generator=g) for c in segs]
        w = torch.randn(cout, cin, ks, ks, generator=g) / math.sqrt(cin * ks * ks)
        b = torch.randn(cout, generator=g) * 0.1
        x = torch.cat(xs, 1).double()
        xin = x if in_slope is None else F.leaky_relu(x, in_slope)
        want = F.conv2d(xin, w.double(), b.double(), stride=stride, padding=ks // 2)
        scale = F.conv2d(xin.abs(), w.double().abs(), None, stride=stride, padding=ks // 2).max().item() + 1.0
        if out_slope == "clamp01":
            want = want.clamp(0, 1)
        elif out_slope is not None:
            want = F.leaky_relu(want, out_slope) if out_slope > 0 else F.relu(want)
        if ps:
            want = F.pixel_shuffle(want, 2)
        res = gate = res2 = None
        if use_res:
            res = torch.randn(want.shape, generator=g)
            r = res.double()
            if use_gate:
                gate = torch.rand(N, want.shape[1], generator=g)
                r = r * gate.double()[:, :, None, None]
            want = want + r
        if use_res2:
            res2 = torch.randn(want.shape, generator=g)
            want = want + res2.double()
        pk = e.pack(("sweep", precision, k), torch.nn.Parameter(w.cuda()), torch.nn.Parameter(b.cuda()), segs, ps)
        views = []
        for i, (xx, c) in enumerate(zip(xs, segs)):
            wide = e.buf(f"sw/in{i}", N, H, W, c + 8)
            views.append(e.from_nchw(xx.cuda(), wide.slice(4, c)))
        full = e.buf("sw/out", N, want.shape[2], want.shape[3], want.shape[1] + 4)
        full.base.fill_(float("nan"))
        out = full.slice(4, want.shape[1])
        rv = to_view(e, "sw/res", res) if res is not None else None
        r2 = to_view(e, "sw/res2", res2) if res2 is not None else None
        gt = gate.cuda().contiguous() if gate is not None else None
        e.conv(pk, views, out, stride=stride, in_slope=in_slope, out_slope=out_slope, res=rv, gate=gt, res2=r2)
        got = e.to_nchw(out).cpu().double()
        assert got.shape == want.shape, case
        err = (got - want).abs().max().item()
        assert err < tol * scale + 2e-7 * want.abs().max().item(), (k, case, err, scale)
    e.check_status()


def test_conv_rejects_bad_arguments(eng):
    from vcm_ts_amd import lib

    a = lib.ConvArgs()
    assert eng.L.dcvc_conv2d(a, None) == -1  # nseg = 0
    w = torch.nn.Parameter(torch.zeros(4, 4, 5, 5))
    with pytest.raises(lib.KernelError):
        eng.pack(("bad",), w, None, (4,), False)  # 5x5 unsupported


def test_warp_and_resamplers_match_reference_fixtures(eng):
    fx = golden("warp")
    for k in range(int(fx["n_warp"])):
        im, fl = torch.from_numpy(fx[f"warp{k}_im"]), torch.from_numpy(fx[f"warp{k}_flow"])
        N, C, H, W = im.shape
        out = eng.warp(to_view(eng, "w/im", im), to_view(eng, "w/fl", fl), eng.buf("w/out", N, H, W, C))
        got = eng.to_nchw(out).cpu().numpy()
        # positions are rebuilt with the reference's own fp32 arithmetic: tight even at W=1920
        np.testing.assert_allclose(got, fx[f"warp{k}_out"], rtol=0, atol=3e-6 * np.abs(fx[f"warp{k}_im"]).max())
    x = torch.from_numpy(fx["resamp_in"])
    N, C, H, W = x.shape
    up = eng.up2(to_view(eng, "r/in", x), eng.buf("r/up", N, 2 * H, 2 * W, C))
    np.testing.assert_allclose(eng.to_nchw(up).cpu().numpy(), fx["resamp_up"], rtol=1e-6, atol=1e-7)
    dn = eng.down2(to_view(eng, "r/in", x), eng.buf("r/dn", N, H // 2, W // 2, C))
    np.testing.assert_allclose(eng.to_nchw(dn).cpu().numpy(), fx["resamp_down"], rtol=1e-6, atol=1e-7)
    ap = eng.down2(to_view(eng, "r/in", x), eng.buf("r/ap", N, H // 2, W // 2, C), avgpool_order=True)
    np.testing.assert_allclose(eng.to_nchw(ap).cpu().numpy(), F.avg_pool2d(x, 2, 2).numpy(), rtol=1e-6, atol=1e-7)
    mp = eng.maxpool2(to_view(eng, "r/in", x), eng.buf("r/mp", N, H // 2, W // 2, C))
    np.testing.assert_array_equal(eng.to_nchw(mp).cpu().numpy(), F.max_pool2d(x, 2).numpy())


@pytest.mark.parametrize("C,N,H,W", [(64, 1, 24, 40), (8, 2, 9, 31), (16, 1, 17, 23), (32, 2, 12, 20), (128, 1, 10, 14),
                                     (24, 1, 11, 13), (64, 1, 7, 5)])
def test_warp_vector_path_and_border(eng, C, N, H, W):
    """Vector paths of the warp: the wave-shuffle kernel (one lane per pixel computes the tap, ds_swizzle
    broadcasts it to the pixel's 2 / 4 / 8 / 16 / 32 channel lanes), the every-lane kernel for other channel
    counts (24), pictures whose pixel count is not a multiple of the lanes per workgroup, two images."""
    g = torch.Generator().manual_seed(C + H)
    im = torch.randn(N, C, H, W, generator=g)
    fl = torch.randn(N, 2, H, W, generator=g) * 30  # mostly out of the picture -> border clamp
    fl[:, :, : H // 2] *= 0.05                       # and sub-pixel motion in the upper half
    want = R.warp(im, fl)
    out = eng.warp(to_view(eng, "wv/im", im), to_view(eng, "wv/fl", fl), eng.buf("wv/out", N, H, W, C))
    assert rel_err(eng.to_nchw(out), want) < 3e-6


def test_layout_round_trip(eng):
    x = torch.randn(2, 67, 19, 23)
    v = to_view(eng, "l/x", x, cs=72)
    np.testing.assert_array_equal(eng.to_nchw(v).cpu().numpy(), x.numpy())
    np.testing.assert_array_equal(v.nchw().cpu().numpy(), x.numpy())  # zero-copy logical NCHW view
    from vcm_ts_amd.engine import View

    al = View.alias(v.nchw())
    assert al is not None and al.ptr == v.ptr and al.cs == 72


def test_se_gate(eng):
    g = torch.Generator().manual_seed(2)
    t = torch.randn(2, 32, 48, 80, generator=g)
    w1, w2 = torch.randn(2, 32, generator=g), torch.randn(32, 2, generator=g)
    want = torch.sigmoid(F.linear(F.relu(F.linear(t.mean(dim=(-1, -2)), w1)), w2))
    gate = eng.se_gate("se/t", to_view(eng, "se/in", t), w1.cuda(), w2.cuda())
    torch.testing.assert_close(gate.cpu().reshape(2, 32), want, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("cout,H,W,stride", [(32, 37, 70, 1), (64, 48, 96, 1), (128, 9, 33, 1), (64, 40, 72, 2)])
def test_fused_se_squeeze_matches_mean_of_the_output(eng, eng_split, cout, H, W, stride):
    """SELayer's global average pool (video_net.py:149-162) computed by the producing convolution's epilogue
    (dcvc_conv_args.chan_partial + dcvc_channel_mean_finish): equals the mean of the tensor the launch stored,
    and the gate built from it equals the separate two-pass reduction's; ragged tiles, two images, both modes."""
    g = torch.Generator().manual_seed(cout + H)
    N, cin = 2, 64
    x = torch.randn(N, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    b = torch.randn(cout, generator=g)
    w1, w2 = torch.randn(max(cout // 16, 1), cout, generator=g), torch.randn(cout, max(cout // 16, 1), generator=g)
    for e in (eng, eng_split):
        pk = e.pack(("sq", cout, H, stride), torch.nn.Parameter(w.cuda()), torch.nn.Parameter(b.cuda()), (cin,), False)
        xin = to_view(e, "sq/in", x)
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        out = e.buf("sq/out", N, Ho, Wo, cout)
        buf, parts = e.chan_partial_buf("sq", pk, out, stride)
        buf.fill_(float("nan"))
        e.conv(pk, [xin], out, stride=stride, chan_partial=buf)
        got_out = e.to_nchw(out).cpu()
        gate_fused = e.se_gate("sq/f", out, w1.cuda(), w2.cuda(), partial=(buf, parts, pk.Cout_pad)).clone()
        mean_fused = e.fbuf("sq/f.mean", N * cout).clone()
        torch.testing.assert_close(mean_fused.cpu().reshape(N, cout), got_out.double().mean(dim=(-1, -2)).float(), rtol=2e-6,
                                   atol=2e-6)
        gate_sep = e.se_gate("sq/s", out, w1.cuda(), w2.cuda())
        torch.testing.assert_close(gate_fused, gate_sep, rtol=1e-5, atol=1e-6)


def test_k32_wave_count_changes_no_bit(eng_split):
    """conv_k32's 64-channel 3x3 kernel as 4- and as 8-wave workgroups (dcvc_conv_k32_set_waves): output, residual path
    AND the fused SELayer channel sums are bit-identical -- the sums inside a tile are ordered by tile row, not by the
    wave that owns the row (the two builds used to differ in the last bit of the SE gate, which made the
    codec's deviation from the reference depend on a tuning knob)."""
    from vcm_ts_amd import lib

    e = eng_split
    g = torch.Generator().manual_seed(77)
    N, cin, cout, H, W = 2, 64, 64, 45, 75  # ragged tiles in both directions
    x = torch.randn(N, cin, H, W, generator=g)
    r = torch.randn(N, cout, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    b = torch.randn(cout, generator=g)
    pk = e.pack(("waves", cout), torch.nn.Parameter(w.cuda()), torch.nn.Parameter(b.cuda()), (cin,), False)
    xin, res = to_view(e, "wv/in", x), to_view(e, "wv/res", r)
    out = e.buf("wv/out", N, H, W, cout)
    buf, parts = e.chan_partial_buf("wv", pk, out, 1)
    e.k32_everywhere = True
    got = {}
    try:
        assert e.k32_capable(pk, 1, out, res, None, None, [xin])
        for waves in (4, 8):
            lib.check(e.L.dcvc_conv_k32_set_waves(waves), "set_waves")
            buf.fill_(float("nan"))
            out.base.fill_(float("nan"))
            e.conv(pk, [xin], out, res=res, out_slope=0.1, chan_partial=buf)
            got[waves] = (out.base.clone(), buf.clone())
    finally:
        e.L.dcvc_conv_k32_set_waves(8)
        e.k32_everywhere = False
    assert torch.equal(got[4][0], got[8][0])
    assert torch.equal(got[4][1][: N * parts * pk.Cout_pad], got[8][1][: N * parts * pk.Cout_pad])
    assert torch.isfinite(got[8][1][: N * parts * pk.Cout_pad]).all()


@pytest.mark.parametrize("ks,cin,cout,H,W", [(3, 64, 64, 68, 120), (3, 96, 32, 68, 120), (1, 64, 128, 68, 120), (3, 128, 48, 61, 97), (7, 32, 64, 68, 120),
                                             (7, 16, 32, 34, 60)])
def test_small_launches_on_4_row_tiles_change_no_bit(eng, eng_split, ks, cin, cout, H, W):
    """dcvc_conv2d runs a stride-1 layer with few workgroups on 4-row tiles instead of 8-row tiles (round 4: twice the
    workgroups for the 1/16-resolution stages and for small training batches).  The tile shape must not enter an output's
    sum: the same picture alone (few workgroups: 4-row tiles) and as the first of a batch of twelve (enough workgroups:
    8-row tiles) gives the same bits, with residual and activation, in both arithmetic modes."""
    g = torch.Generator().manual_seed(ks * 100 + cin)
    x = torch.randn(12, cin, H, W, generator=g)
    r = torch.randn(12, cout, H, W, generator=g)
    w = torch.randn(cout, cin, ks, ks, generator=g) / math.sqrt(cin * ks * ks)
    b = torch.randn(cout, generator=g)
    for e in (eng, eng_split):
        pk = e.pack(("rows4", ks, cin, cout), torch.nn.Parameter(w.cuda()), torch.nn.Parameter(b.cuda()), (cin,), False)
        outs = []
        for n in (1, 12):
            xin, res = to_view(e, f"r4/in{n}", x[:n]), to_view(e, f"r4/res{n}", r[:n])
            out = e.buf(f"r4/out{n}", n, H, W, cout)
            out.base.fill_(float("nan"))
            e.conv(pk, [xin], out, res=res, out_slope=0.1, in_slope=0.01)
            outs.append(e.to_nchw(out)[0].clone())
        assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1]), (e.precision, ks, cin, cout)


def test_build_indexes_bit_exact_against_reference_planes(eng):
    """GaussianEncoder.build_indexes (entropy_models.py:264-268) on the device, integer-equal to what
    the REFERENCE produced: its sweep over bin edges / zeros / negatives (tables.npz) and the index
    planes it derived from its own scale planes in every seq_*.npz; plus every fp32 value within 3 ulp
    of each of the 255 bin edges and 4M random scales against the oracle (which test_oracle_golden
    pins to the same fixtures)."""
    from vcm_ts_amd.entropy import scale_index_edges

    fx = golden("tables")
    s = torch.from_numpy(fx["idx_sweep_in"]).cuda()
    for dist, key in (("laplace", "idx_sweep_laplace"), ("gaussian", "idx_sweep_gauss")):
        np.testing.assert_array_equal(eng.scale_indexes(s, dist).cpu().numpy(), fx[key])
    n_planes = 0
    for name in ("seq_64", "seq_128", "seq_192x320", "seq_256"):
        sq = golden(name)
        for k in sq.files:
            if "scale_" not in k:
                continue
            dist = "gaussian" if k.startswith("i_") else "laplace"
            got = eng.scale_indexes(torch.from_numpy(sq[k]).cuda(), dist).cpu().numpy()
            np.testing.assert_array_equal(got.astype(np.int16), sq[k.replace("scale_", "idx_")], err_msg=f"{name}:{k}")
            n_planes += 1
    assert n_planes >= 10
    g = torch.Generator().manual_seed(9)
    for dist in ("laplace", "gaussian"):
        eb = scale_index_edges(dist)[:-1].view(torch.int32)
        near = torch.cat([(eb + d).view(torch.float32) for d in range(-3, 4)])
        rnd = torch.exp(torch.empty(4_000_000).uniform_(-13.0, 6.0, generator=g))
        edges = scale_index_edges(dist)
        for t, what in ((near, "within 3 ulp of an edge"), (rnd, "random")):
            got = eng.scale_indexes(t.cuda(), dist).cpu()
            # the kernels count the CONSTANT edges (vcm_ts_amd/index_edges.py, derived where the reference fixtures were made)
            assert torch.equal(got, torch.searchsorted(edges, t, right=True).int())
            # the reference formula evaluated by THIS host's torch-CPU logf may sit one ulp off at an edge (round 3: the
            # GPU box's CPU bins 15 of the 1785 near-edge floats differently from the build container's -- which is
            # why the edges are constants now: two hosts that each derived their own would not decode each other)
            host = R.scale_indexes(t.clone(), dist)
            n_off = int((host != got).sum())
            print(f"[{dist}] {what}: {n_off} of {t.numel()} binned differently by this host's logf")
            assert n_off <= (0.02 * t.numel() if t is near else 8) and int((host - got).abs().max()) <= 1


@pytest.mark.parametrize("dist,C", [("laplace", 64), ("gaussian", 192)])
def test_dual_prior_matches_oracle(eng, dist, C):
    """Encoder kernels (estimate + symbol planes) against oracle.dual_prior with the spatial
    prior replaced by a fixed random tensor, then the decoder kernels fed the same symbols."""
    g = torch.Generator().manual_seed(3)
    N, H, W = 2, 10, 14
    y = torch.randn(N, C, H, W, generator=g) * 4
    fusion = torch.randn(N, 3 * C, H, W, generator=g)
    fusion[:, C : 2 * C] = fusion[:, C : 2 * C].abs() * 2  # scales (some below the table minimum)
    spatial = torch.randn(N, 2 * C, H, W, generator=g)
    spatial[:, : C // 2] = spatial[:, : C // 2].abs()
    spatial[:, C : C + C // 2] = spatial[:, C : C + C // 2].abs()
    qb, qs = torch.rand(C, generator=g) + 0.3, torch.tensor([1.2, 0.7])

    class FixedPrior(dict):
        pass

    # oracle with the spatial prior conv stack monkey-patched to return `spatial`
    orig = R.three_convs
    R.three_convs = lambda w, name, x, slope=0.2: spatial
    try:
        qs_, sc_, mu_ = fusion.chunk(3, 1)
        o = R.dual_prior({}, "x", y, mu_, sc_, qs_)
    finally:
        R.three_convs = orig
    curr_q = (qb.clamp_min(0.5)[None, :, None, None] * qs[:, None, None, None])
    want_out = o["y_hat"] * curr_q

    yv, fv, sv = to_view(eng, "dp/y", y), to_view(eng, "dp/f", fusion), to_view(eng, "dp/s", spatial)
    params = eng.buf("dp/params", N, H, W, 4 * C)
    n = N * H * W * C
    y_hat, y_q, sh = eng.fbuf("dp/yh", n), eng.fbuf("dp/yq", n), eng.fbuf("dp/sh", n)
    sym = [eng.ibuf(f"dp/sym{k}", n // 2) for k in (0, 1)]
    idx = [eng.ibuf(f"dp/idx{k}", n // 2) for k in (0, 1)]
    out = eng.buf("dp/out", N, H, W, C)
    common = dict(y=yv, fusion=fv, params=params, y_hat=y_hat, y_q=y_q, scales_hat=sh, distribution=dist)
    eng.dual_prior("enc", 0, sym=sym[0], idx=idx[0], **common)
    want_params = torch.cat((o["q_w0"] * 0, ) , 1)  # placeholder to keep shapes obvious
    eng.dual_prior("enc", 1, spatial=sv, sym=sym[1], idx=idx[1], out=out, q_basic=qb.cuda(), q_scale=qs.cuda(), **common)
    nhwc = lambda t: t.permute(0, 2, 3, 1).reshape(-1)
    torch.testing.assert_close(y_q.cpu(), nhwc(o["y_q"]), rtol=0, atol=0)
    torch.testing.assert_close(sh.cpu(), nhwc(o["scales_hat"]), rtol=0, atol=0)
    torch.testing.assert_close(eng.to_nchw(out).cpu(), want_out, rtol=1e-6, atol=1e-6)
    for k, (qw, sw) in enumerate(((o["q_w0"], o["s_w0"]), (o["q_w1"], o["s_w1"]))):
        np.testing.assert_array_equal(sym[k].cpu().numpy().reshape(N, C // 2, H, W), qw.numpy().astype(np.int32))
        want_idx = R.scale_indexes(sw, dist).numpy()
        got_idx = idx[k].cpu().numpy().reshape(N, C // 2, H, W)
        # bit-exact: the kernel counts the reference's own fp32 bin edges (no device logarithm)
        np.testing.assert_array_equal(got_idx, want_idx)
    # params buffer after step 0 = [y_hat_0_0 | y_hat_1_1 | means | scales | q_step]
    # decoder: same indexes, then apply the encoder's symbols -> identical y_hat, bit for bit
    y_hat2 = eng.fbuf("dp/yh2", n)
    out2 = eng.buf("dp/out2", N, H, W, C)
    params2 = eng.buf("dp/params2", N, H, W, 4 * C)
    idx2 = eng.ibuf("dp/idx_d", n // 2)
    dcommon = dict(fusion=fv, params=params2, y_hat=y_hat2, distribution=dist)
    eng.dual_prior("dec_index", 0, idx=idx2, **dcommon)
    np.testing.assert_array_equal(idx2.cpu().numpy(), idx[0].cpu().numpy())
    eng.dual_prior("dec_apply", 0, sym=sym[0], **dcommon)
    np.testing.assert_array_equal(params2.base.cpu().numpy(), params.base.cpu().numpy())
    eng.dual_prior("dec_index", 1, spatial=sv, idx=idx2, **dcommon)
    np.testing.assert_array_equal(idx2.cpu().numpy(), idx[1].cpu().numpy())
    eng.dual_prior("dec_apply", 1, spatial=sv, sym=sym[1], out=out2, q_basic=qb.cuda(), q_scale=qs.cuda(), **dcommon)
    np.testing.assert_array_equal(out2.base.cpu().numpy(), out.base.cpu().numpy())


def test_bit_estimates_match_oracle(eng):
    g = torch.Generator().manual_seed(4)
    yq = torch.round(torch.randn(2, 5000, generator=g) * 3)
    sc = torch.rand(2, 5000, generator=g) * 4 - 0.2
    for gaussian, fn in ((False, R.laplace_bits), (True, R.gaussian_bits)):
        want = fn(yq, sc).sum(dim=1)
        got = eng.scale_bits(yq.cuda().contiguous(), sc.cuda().contiguous(), 2, 5000, gaussian=gaussian)
        torch.testing.assert_close(got.cpu(), want, rtol=2e-5, atol=1e-3)
    from tests.util import oracle_weights
    from vcm_ts_amd import entropy as E

    w = oracle_weights("dmc")
    z = torch.round(torch.randn(2, 64, 6, 9, generator=g) * 2)
    want = R.z_bits(w, "bit_estimator_z", z).sum(dim=(1, 2, 3))
    blk = E.factorized_param_block(E.factorized_params(w, "bit_estimator_z")).cuda()
    got = eng.factorized_bits(to_view(eng, "fb/z", z), blk)
    torch.testing.assert_close(got.cpu(), want, rtol=2e-5, atol=1e-3)
    a, b = torch.rand(2, 3, 20, 30, generator=g), torch.rand(2, 3, 20, 30, generator=g)
    got = eng.sq_err(to_view(eng, "se/a", a, cs=8), to_view(eng, "se/b", b))
    torch.testing.assert_close(got.cpu(), ((a - b) ** 2).sum(dim=(1, 2, 3)), rtol=1e-5, atol=1e-5)


def test_round_symbols_half_to_even(eng):
    z = torch.tensor([0.5, 1.5, 2.5, -0.5, -1.5, 3.49999, -7.5, 0.0]).reshape(1, 8, 1, 1)
    zh = eng.buf("rs/zh", 1, 1, 1, 8)
    sym = eng.ibuf("rs/sym", 8)
    eng.round_symbols(to_view(eng, "rs/z", z), zh, sym)
    np.testing.assert_array_equal(sym.cpu().numpy(), torch.round(z).reshape(-1).int().numpy())
    back = eng.symbols_to_nhwc(sym, eng.buf("rs/back", 1, 1, 1, 8))
    np.testing.assert_array_equal(eng.to_nchw(back).cpu().numpy(), torch.round(z).numpy())
