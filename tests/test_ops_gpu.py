"""-m gpu: every HIP op (through the C ABI) against the fp32 CPU oracle building blocks (torch.nn.functional
on the CPU -- the exact ops the reference executes), forward and backward, on the layer shapes of the path:
odd PatchGAN sizes, Cin 38/41, Cout 3/1, reflect pads, stride-2 phases, split-K wgrad."""
import ctypes
import pytest
import torch
import torch.nn.functional as F

from util import assert_close

pytestmark = pytest.mark.gpu

DEV = 'cuda'


def _ops():
    from neurips18_hierchical_image_manipulation_amd import ops
    return ops


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _ref_conv(x, w, b, stride, pad, pad_mode, act):
    if pad_mode == 'reflect':
        x = F.pad(x, (pad, pad, pad, pad), mode='reflect')
        pad = 0
    y = F.conv2d(x, w, b, stride, pad)
    if act == 'relu':
        y = F.relu(y)
    elif act == 'lrelu':
        y = F.leaky_relu(y, 0.2)
    elif act == 'tanh':
        y = torch.tanh(y)
    return y


CONV_CASES = [
    # B, Cin, H, W, Cout, k, stride, pad, pad_mode, act
    (2, 38, 20, 36, 64, 7, 1, 3, 'reflect', 'none'),     # G stem
    (2, 16, 16, 32, 16, 3, 1, 1, 'reflect', 'none'),     # ResnetBlock conv
    (1, 160, 9, 13, 136, 3, 1, 1, 'reflect', 'none'),    # M,N,K tails in every tile dim
    (2, 64, 16, 32, 128, 3, 2, 1, 'zero', 'none'),       # downsample
    (2, 41, 32, 64, 64, 4, 2, 2, 'zero', 'lrelu'),       # D first block (odd output 17x33)
    (2, 64, 17, 33, 128, 4, 2, 2, 'zero', 'none'),       # D block on odd input
    (2, 96, 9, 17, 160, 4, 1, 2, 'zero', 'none'),        # D stride-1 block
    (2, 160, 10, 18, 1, 4, 1, 2, 'zero', 'none'),        # D head -> 1 channel
    (2, 64, 20, 36, 3, 7, 1, 3, 'reflect', 'tanh'),      # G head -> 3 channels + tanh
    (2, 3, 16, 32, 64, 3, 1, 1, 'zero', 'relu'),         # VGG conv1_1 + bias + ReLU
    (1, 256, 8, 16, 256, 3, 1, 1, 'zero', 'relu'),       # VGG mid
    (3, 8, 5, 7, 8, 3, 1, 1, 'reflect', 'none'),         # tiny everything
    (1, 16, 3, 3, 16, 3, 1, 1, 'reflect', 'none'),       # folded reflect dgrad: both mirror rows hit y = 1
    (2, 20, 4, 5, 24, 3, 1, 1, 'reflect', 'none'),       # folded reflect dgrad, channel tails (Cout 24)
    (1, 32, 2, 4, 16, 3, 1, 1, 'reflect', 'none'),       # H = 2: padded-gradient + fold fallback
    (2, 256, 16, 32, 256, 3, 1, 1, 'reflect', 'none'),   # folded reflect dgrad under split-K
    (2, 20, 9, 70, 2, 3, 1, 1, 'zero', 'none'),          # tiny-M sliding-window wgrad, zero pad, W > one wave
    (1, 12, 70, 9, 4, 5, 1, 2, 'reflect', 'none'),       # tiny-M sliding-window wgrad 5x5, H > one row chunk
    (2, 64, 32, 32, 96, 1, 2, 0, 'zero', 'none'),        # ConvResnetBlock shortcut: 1x1 stride 2 (empty stride phases)
    (2, 16, 9, 11, 16, 1, 2, 0, 'zero', 'none'),         # 1x1 stride 2 on odd sizes
    (2, 64, 8, 8, 32, 1, 1, 0, 'zero', 'none'),          # 1x1 stride 1: the 'concat' feature fusion (2C -> C on the latent plane)
    (2, 256, 4, 8, 128, 1, 1, 0, 'zero', 'none'),        # the same at 32 positions per image
    (2, 70, 32, 32, 64, 7, 2, 3, 'zero', 'none'),        # box2mask stem conv7 stride 2
    (3, 128, 8, 10, 256, 4, 1, 2, 'zero', 'none'),       # PatchGAN plane 9x11 = 99 (not a multiple of 4): fast wgrad, scalar dY quads
    (2, 64, 7, 9, 64, 3, 1, 1, 'reflect', 'none'),       # odd plane 63 with reflect gather in the fast wgrad
    (1, 8, 20, 70, 2, 7, 1, 3, 'zero', 'none'),          # tiny-M 7x7 wgrad, 2 outputs: filter-row band split (4 + 3 rows)
    (2, 8, 12, 66, 4, 7, 1, 3, 'reflect', 'none'),       # tiny-M 7x7 wgrad, 4 outputs, reflect, two column strips
    (2, 128, 16, 32, 128, 3, 1, 1, 'zero', 'relu'),      # fused Winograd kernel: VGG conv2_2-like (fwd + zero-pad dgrad)
    (3, 256, 9, 13, 192, 3, 1, 1, 'zero', 'none'),       # fused Winograd: odd plane (partial 8x8 tile blocks), Cout 192
    (2, 136, 18, 20, 64, 3, 1, 1, 'reflect', 'none'),    # fused Winograd forward with reflection (dgrad: direct form)
    (1, 256, 2, 2, 256, 3, 1, 1, 'reflect', 'none'),     # fused Winograd: a single 2x2 tile, both borders in one patch
    (8, 24, 128, 256, 3, 7, 1, 3, 'reflect', 'tanh'),    # LDS-tiled few-output kernel: G head form (512 tiles of 16x64)
    (40, 16, 40, 70, 3, 7, 1, 3, 'reflect', 'none'),     # tiled few-output kernel, ragged tiles on both axes (W = 70, H = 40)
    (36, 3, 40, 66, 16, 3, 1, 1, 'zero', 'relu'),        # tiled few-output kernel as the DATA GRADIENT of a 3 -> 16 3x3 layer (flipped taps), ragged
    (8, 2, 130, 250, 8, 3, 1, 1, 'zero', 'none'),        # the same with 2 input channels, W % 4 != 0 (scalar stores)
    (34, 12, 33, 65, 4, 3, 1, 1, 'zero', 'none'),        # tiled few-output forward, 4 outputs, 3x3 zero padding, odd plane
    (8, 32, 128, 250, 3, 7, 1, 3, 'reflect', 'none'),    # few-INPUT tiled kernel: data gradient of the G head (3 -> 32 on the padded plane + fold)
    (8, 16, 130, 250, 2, 3, 1, 1, 'zero', 'none'),       # few-input tiled kernel: data gradient of a 2-output 3x3 layer
    (8, 3, 128, 256, 32, 3, 1, 1, 'zero', 'relu'),       # few-input tiled kernel, forward form (VGG conv1_1), bias + ReLU
    (8, 4, 120, 250, 16, 7, 1, 3, 'zero', 'none'),       # few-input tiled kernel, forward 7x7, ragged tiles
    (2, 96, 37, 150, 3, 7, 1, 3, 'zero', 'none'),        # MFMA few-channel wgrad, head form: 3 channel groups of 32, 2 column chunks, zero pad
    (3, 64, 40, 300, 4, 5, 1, 2, 'reflect', 'none'),     # MFMA few-channel wgrad, head form 5x5, 4 outputs, 3 chunks, several row bands
    (2, 3, 37, 150, 64, 7, 1, 3, 'reflect', 'none'),     # MFMA few-channel wgrad, stem form (3 dense inputs -> 64), reflect
    (2, 4, 20, 140, 32, 5, 1, 2, 'zero', 'none'),        # MFMA few-channel wgrad, stem form 5x5, 4 inputs -> 32, zero pad
    (8, 64, 64, 128, 3, 7, 1, 3, 'reflect', 'none'),     # MFMA few-channel wgrad: every workgroup walks several tasks
]


@pytest.mark.parametrize('case', CONV_CASES, ids=lambda c: 'x'.join(map(str, c)))
def test_conv2d_fwd_bwd(case):
    ops = _ops()
    B, Cin, H, W, Cout, k, s, p, pm, act = case
    x = _rand(B, Cin, H, W, seed=1).requires_grad_(True)
    w = _rand(Cout, Cin, k, k, seed=2, scale=(Cin * k * k) ** -0.5).requires_grad_(True)
    b = _rand(Cout, seed=3, scale=0.1).requires_grad_(True)
    y_ref = _ref_conv(x, w, b, s, p, pm, act)
    gy = _rand(*y_ref.shape, seed=4)
    gx_ref, gw_ref, gb_ref = torch.autograd.grad(y_ref, (x, w, b), gy)

    xd, wd, bd = (t.detach().to(DEV).requires_grad_(True) for t in (x, w, b))
    y = ops.conv2d(xd, wd, bd, s, p, pm, act, 0.2)
    assert_close('conv fwd', y, y_ref)
    gx, gw, gb = torch.autograd.grad(y, (xd, wd, bd), gy.to(DEV))
    assert_close('conv dgrad', gx, gx_ref)
    assert_close('conv wgrad', gw, gw_ref)
    assert_close('conv bgrad', gb, gb_ref)


def test_conv2d_wgrad_splitk_large_spatial():
    """K = B*OH*OW = 131072 -> many split-K slabs, fixed-order reduction; also accumulate mode."""
    ops = _ops()
    B, Cin, H, W, Cout = 2, 8, 256, 256, 16
    x = _rand(B, Cin, H, W, seed=1)
    w = _rand(Cout, Cin, 3, 3, seed=2, scale=0.1).requires_grad_(True)
    y_ref = F.conv2d(x, w, None, 1, 1)
    gy = _rand(*y_ref.shape, seed=4, scale=0.1)
    (gw_ref,) = torch.autograd.grad(y_ref, (w,), gy)
    wd = w.detach().to(DEV).requires_grad_(True)
    y = ops.conv2d(x.to(DEV), wd, None, 1, 1, 'zero', 'none')
    (gw,) = torch.autograd.grad(y, (wd,), gy.to(DEV))
    assert_close('wgrad split-K', gw, gw_ref, rtol=1e-4)
    (gw2,) = torch.autograd.grad(ops.conv2d(x.to(DEV), wd, None, 1, 1, 'zero', 'none'), (wd,), gy.to(DEV))
    assert torch.equal(gw, gw2), 'wgrad must be run-to-run deterministic'


DECONV_CASES = [(2, 64, 8, 16, 32), (2, 128, 5, 9, 64), (1, 16, 16, 32, 8), (2, 136, 4, 6, 72)]


@pytest.mark.parametrize('case', DECONV_CASES, ids=lambda c: 'x'.join(map(str, c)))
@pytest.mark.parametrize('act', ['none', 'relu'])
def test_conv_transpose2d_fwd_bwd(case, act):
    ops = _ops()
    B, Cin, H, W, Cout = case
    x = _rand(B, Cin, H, W, seed=1).requires_grad_(True)
    w = _rand(Cin, Cout, 3, 3, seed=2, scale=(Cin * 9) ** -0.5).requires_grad_(True)
    b = _rand(Cout, seed=3, scale=0.1).requires_grad_(True)
    y_ref = F.conv_transpose2d(x, w, b, stride=2, padding=1, output_padding=1)
    if act == 'relu':
        y_ref = F.relu(y_ref)
    gy = _rand(*y_ref.shape, seed=4)
    gx_ref, gw_ref, gb_ref = torch.autograd.grad(y_ref, (x, w, b), gy)
    xd, wd, bd = (t.detach().to(DEV).requires_grad_(True) for t in (x, w, b))
    y = ops.conv_transpose2d(xd, wd, bd, 2, 1, 1, act)
    assert_close('deconv fwd', y, y_ref)
    gx, gw, gb = torch.autograd.grad(y, (xd, wd, bd), gy.to(DEV))
    assert_close('deconv dgrad', gx, gx_ref)
    assert_close('deconv wgrad', gw, gw_ref)
    assert_close('deconv bgrad', gb, gb_ref)


@pytest.mark.parametrize('shape', [(2, 16, 16, 32), (2, 8, 2, 3), (1, 4, 33, 65), (2, 3, 64, 128), (1, 2, 300, 301),
                                   (1, 3, 256, 512), (1, 2, 181, 183)],
                         ids=str)
@pytest.mark.parametrize('act', ['none', 'relu', 'lrelu'])
@pytest.mark.parametrize('with_res', [False, True])
def test_instance_norm(shape, act, with_res):
    if with_res and act != 'none':
        pytest.skip('residual only follows an un-activated norm')
    ops = _ops()
    x = (_rand(*shape, seed=1) * 2 + 0.5).requires_grad_(True)
    r = _rand(*shape, seed=2).requires_grad_(True) if with_res else None
    y_ref = F.instance_norm(x, eps=1e-5)
    if act == 'relu':
        y_ref = F.relu(y_ref)
    elif act == 'lrelu':
        y_ref = F.leaky_relu(y_ref, 0.2)
    if with_res:
        y_ref = r + y_ref
    gy = _rand(*shape, seed=3)
    ins = (x, r) if with_res else (x,)
    g_ref = torch.autograd.grad(y_ref, ins, gy)
    xd = x.detach().to(DEV).requires_grad_(True)
    rd = r.detach().to(DEV).requires_grad_(True) if with_res else None
    y = ops.instance_norm(xd, rd, act, 0.2)
    assert_close('IN fwd', y, y_ref, rtol=2e-5)
    g = torch.autograd.grad(y, (xd, rd) if with_res else (xd,), gy.to(DEV))
    assert_close('IN bwd x', g[0], g_ref[0], rtol=1e-4)
    if with_res:
        assert_close('IN bwd res', g[1], g_ref[1])


@pytest.mark.parametrize('hw', [(16, 32), (17, 33), (9, 17), (256, 512), (5, 4)], ids=str)
def test_avgpool3s2(hw):
    ops = _ops()
    x = _rand(2, 5, *hw, seed=1).requires_grad_(True)
    y_ref = F.avg_pool2d(x, 3, 2, 1, count_include_pad=False)
    gy = _rand(*y_ref.shape, seed=2)
    (gx_ref,) = torch.autograd.grad(y_ref, x, gy)
    xd = x.detach().to(DEV).requires_grad_(True)
    y = ops.avgpool3s2(xd)
    assert_close('avgpool fwd', y, y_ref, rtol=1e-6)
    (gx,) = torch.autograd.grad(y, xd, gy.to(DEV))
    assert_close('avgpool bwd', gx, gx_ref, rtol=1e-6)


@pytest.mark.parametrize('k,hw', [(2, (16, 32)), (2, (8, 8)), (16, (64, 64)), (8, (32, 64))], ids=str)
def test_maxpool(k, hw):
    ops = _ops()
    x = F.relu(_rand(2, 4, *hw, seed=1)).requires_grad_(True)    # ReLU'd input -> ties at 0
    y_ref = F.max_pool2d(x, k, k)
    gy = _rand(*y_ref.shape, seed=2)
    (gx_ref,) = torch.autograd.grad(y_ref, x, gy)
    xd = x.detach().to(DEV).requires_grad_(True)
    y = ops.maxpool(xd, k)
    assert torch.equal(y.cpu(), y_ref)
    (gx,) = torch.autograd.grad(y, xd, gy.to(DEV))
    assert torch.equal(gx.cpu(), gx_ref), 'max-pool gradient routing (first max in window order)'


@pytest.mark.parametrize('n', [7, 4096, 8 * 64 * 129 * 257 // 16, 1 << 20], ids=str)
def test_losses(n):
    ops = _ops()
    a = _rand(n, seed=1).requires_grad_(True)
    b = _rand(n, seed=2)
    b[: n // 3] = a.detach()[: n // 3]                        # exact zeros of a-b -> sign(0) = 0
    l_ref = F.l1_loss(a, b)
    (ga_ref,) = torch.autograd.grad(l_ref * 3.0, a)
    ad = a.detach().to(DEV).requires_grad_(True)
    l = ops.l1_mean(ad, b.to(DEV))
    assert_close('l1', l, l_ref, rtol=2e-6)
    (ga,) = torch.autograd.grad(l * 3.0, ad)
    assert_close('l1 grad', ga, ga_ref, rtol=1e-6)
    for t in (0.0, 1.0):
        m_ref = F.mse_loss(a, torch.full_like(a, t))
        (gm_ref,) = torch.autograd.grad(m_ref, a)
        m = ops.mse_const(ad, t)
        assert_close('mse', m, m_ref, rtol=2e-6)
        (gm,) = torch.autograd.grad(m, ad)
        assert_close('mse grad', gm, gm_ref, rtol=1e-6)


def test_adam_matches_torch_over_steps():
    from neurips18_hierchical_image_manipulation_amd.optim import FusedAdam
    ps = [torch.nn.Parameter(_rand(33, 7, seed=1)), torch.nn.Parameter(_rand(130, seed=2))]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    o_ref = torch.optim.Adam(ref, lr=2e-4, betas=(0.5, 0.999))
    dps = [torch.nn.Parameter(p.detach().clone().to(DEV)) for p in ps]
    o = FusedAdam(dps, lr=2e-4, betas=(0.5, 0.999))
    for step in range(25):
        o.zero_grad()
        o_ref.zero_grad()
        for i, (p, q) in enumerate(zip(ref, dps)):
            g = _rand(*p.shape, seed=100 + 7 * step + i) * (10.0 ** (i - 2))
            p.grad = g.clone()
            q.grad.copy_(g)
        o_ref.step()
        o.step()
    for p, q in zip(ref, dps):
        assert_close('adam param', q, p, rtol=1e-6)


def test_cat_blend_encode():
    ops = _ops()
    B, H, W = 2, 16, 32
    a = _rand(B, 5, H, W, seed=1).requires_grad_(True)
    b = _rand(B, 3, H, W, seed=2).requires_grad_(True)
    m = (torch.rand(B, 1, H, W, generator=torch.Generator().manual_seed(3)) > 0.5).float()
    ref = torch.cat((a, b), 1) * m
    gy = _rand(*ref.shape, seed=4)
    ga_ref, gb_ref = torch.autograd.grad(ref, (a, b), gy)
    ad, bd = a.detach().to(DEV).requires_grad_(True), b.detach().to(DEV).requires_grad_(True)
    out = ops.cat_channels([ad, bd], m.to(DEV), 1)
    assert torch.equal(out.cpu(), ref.detach())
    ga, gb = torch.autograd.grad(out, (ad, bd), gy.to(DEV))
    assert torch.equal(ga.cpu(), ga_ref) and torch.equal(gb.cpu(), gb_ref)
    # one mask mode per tensor: cat((1-m)*a, m*b), the 'concat' feature fusion's input (Pix2Pix_NET.py:215-217)
    ref = torch.cat(((1 - m) * a, m * b), 1)
    ga_ref, gb_ref = torch.autograd.grad(ref, (a, b), gy)
    out = ops.cat_channels([ad, bd], m.to(DEV), (2, 1))
    assert torch.equal(out.cpu(), ref.detach())
    ga, gb = torch.autograd.grad(out, (ad, bd), gy.to(DEV))
    assert torch.equal(ga.cpu(), ga_ref) and torch.equal(gb.cpu(), gb_ref)
    with pytest.raises(ValueError):
        ops.cat_channels([ad, bd], m.to(DEV), (2, 1, 0))
    # blend with a channel-sliced first operand (output gate) and full (two-stream fusion)
    img = _rand(B, 9, H, W, seed=5).requires_grad_(True)
    gen = _rand(B, 3, H, W, seed=6).requires_grad_(True)
    ref = (1 - m) * img[:, 6:] + m * gen
    gi_ref, gg_ref = torch.autograd.grad(ref, (img, gen), gy[:, :3])
    imd, gd = img.detach().to(DEV).requires_grad_(True), gen.detach().to(DEV).requires_grad_(True)
    out = ops.blend(imd, gd, m.to(DEV), a0=6)
    assert torch.equal(out.cpu(), ref.detach())
    gi, gg = torch.autograd.grad(out, (imd, gd), gy[:, :3].to(DEV))
    assert torch.equal(gi.cpu(), gi_ref) and torch.equal(gg.cpu(), gg_ref)
    # encode: one-hot | edges | (1-mask)*image
    label = torch.randint(0, 35, (B, 1, H, W), generator=torch.Generator().manual_seed(7)).float()
    inst = torch.randint(0, 3, (B, 1, H, W), generator=torch.Generator().manual_seed(8)).float()
    image = _rand(B, 3, H, W, seed=9)
    buf, nl, nc = ops.encode_channels(label.to(DEV), inst.to(DEV), image.to(DEV), m.to(DEV), 35, True)
    onehot = torch.zeros(B, 35, H, W).scatter_(1, label.long(), 1.0)
    from oracle.ref_cpu import get_edges
    ref = torch.cat((onehot, get_edges(inst), (1 - m) * image), 1)
    assert (nl, nc) == (36, 3)
    assert torch.equal(buf.cpu(), ref)
    s = ops.add(ad, ad)
    assert torch.equal(s.cpu(), (a + a).detach())


def test_object_gated_combination_of_stream_logits():
    """ops.gate_comb = (1 - p) * ctx + p * obj with p, obj (B,1,H,W) broadcast over ctx's channels
    (MaskTwoStreamConv_NET.py:213-221): forward bit-identical to torch (separately rounded products), three gradients."""
    ops = _ops()
    B, C, H, W = 2, 35, 9, 13
    a = _rand(B, C, H, W, seed=1).requires_grad_(True)
    l = _rand(B, 1, H, W, seed=2).requires_grad_(True)
    p = torch.sigmoid(_rand(B, 1, H, W, seed=3)).requires_grad_(True)
    g = p.expand_as(a)
    ref = (1 - g) * a + g * l
    gy = _rand(B, C, H, W, seed=4)
    ga_ref, gp_ref, gl_ref = torch.autograd.grad(ref, (a, p, l), gy)
    ad, pd, ld = (t.detach().to(DEV).requires_grad_(True) for t in (a, p, l))
    out = ops.gate_comb(ad, pd, ld)
    assert torch.equal(out.cpu(), ref.detach())
    ga, gp, gl = torch.autograd.grad(out, (ad, pd, ld), gy.to(DEV))
    assert_close('gate_comb dctx', ga, ga_ref, rtol=1e-6)
    assert_close('gate_comb dgate', gp, gp_ref, rtol=1e-5)
    assert_close('gate_comb dobj', gl, gl_ref, rtol=1e-5)


@pytest.mark.parametrize('fusion,norm', [('add', 'instance'), ('concat', 'instance'), ('concat', 'batch')])
def test_feature_fusion_block_both_call_forms(fusion, norm):
    """FeatureFusionBlock (reference layer_util.py:295-330): the reference's own call ``fuser(x, y)`` on pre-masked features
    and the generator's ``fuser(ctx, obj, m)`` (masks folded into the copy kernels) against the oracle's torch block."""
    from oracle import ref_cpu
    from neurips18_hierchical_image_manipulation_amd import synth
    from neurips18_hierchical_image_manipulation_amd.models.Pix2Pix_NET import FeatureFusionBlock
    from neurips18_hierchical_image_manipulation_amd.models.layer_util import get_norm_layer
    B, C, H, W = 2, 16, 6, 10
    o = ref_cpu.FeatureFusionBlock(C, fusion, ref_cpu.get_norm_layer(norm))
    h = FeatureFusionBlock(C, fusion, get_norm_layer(norm)).to(DEV)
    sd = synth.init_state_dict(o.state_dict(), 9)
    o.load_state_dict(sd)
    h.load_state_dict(sd)
    ctx, obj = _rand(B, C, H, W, seed=1).requires_grad_(True), _rand(B, C, H, W, seed=2).requires_grad_(True)
    m = (torch.rand(B, 1, H, W, generator=torch.Generator().manual_seed(3)) > 0.5).float()
    ref = o((1 - m) * ctx, m * obj)
    gy = _rand(*ref.shape, seed=4)
    gc_ref, go_ref = torch.autograd.grad(ref, (ctx, obj), gy)
    cd, od = ctx.detach().to(DEV).requires_grad_(True), obj.detach().to(DEV).requires_grad_(True)
    ops = _ops()
    for out in (h(cd, od, m.to(DEV)), h(ops.cat_channels([cd], m.to(DEV), 2), ops.mul_mask(od, m.to(DEV)))):
        assert_close('fusion fwd', out, ref, rtol=1e-5)
        gc, go = torch.autograd.grad(out, (cd, od), gy.to(DEV))
        assert_close('fusion d ctx', gc, gc_ref, rtol=1e-4)
        assert_close('fusion d obj', go, go_ref, rtol=1e-4)


def test_masked_mean_color():
    ops = _ops()
    from oracle.ref_cpu import color_embedding
    B, H, W = 3, 16, 16
    image = _rand(B, 3, H, W, seed=1)
    m = torch.zeros(B, 1, H, W)
    m[0, :, 4:12, 4:12] = 1
    m[1, :, :, :] = 1                                          # image 2 keeps an empty mask -> zeros
    noise = torch.rand(B, 3, generator=torch.Generator().manual_seed(2)) * 0.06 + 0.97
    for nz in (None, noise):
        ref = color_embedding(m, image, nz)
        got = ops.masked_mean_color(image.to(DEV), m.to(DEV), None if nz is None else nz.to(DEV))
        assert_close('colour embedding', got, ref, rtol=2e-6)


@pytest.mark.parametrize('shape', [(16, 8, 3, 3), (512, 256, 4, 4), (64, 41, 4, 4)], ids=str)
def test_spectral_norm_sigma_and_full_gradient(shape):
    ops = _ops()
    from oracle.ref_cpu import max_singular_value
    g = torch.Generator().manual_seed(13)
    W = (torch.randn(*shape, generator=g) * 0.05).requires_grad_(True)
    u = torch.randn(1, shape[0], generator=g)
    sig_ref, u_ref = max_singular_value(W, u, 1)
    Wbar_ref = W / sig_ref
    gy = _rand(*shape, seed=5)
    (gW_ref,) = torch.autograd.grad(Wbar_ref, W, gy)
    Wd = W.detach().to(DEV).requires_grad_(True)
    sig, u_new = ops.sn_max_singular_value(Wd, u.to(DEV))
    assert_close('sigma', sig, sig_ref, rtol=2e-6)
    assert_close('u', u_new, u_ref, rtol=1e-5)
    Wbar = ops.div_scalar(Wd, sig)
    assert_close('W/sigma', Wbar, Wbar_ref, rtol=1e-5)
    (gW,) = torch.autograd.grad(Wbar, Wd, gy.to(DEV))
    assert_close('d(W/sigma)/dW through both normalisations', gW, gW_ref, rtol=2e-5)


def test_weight_panel_cache_matches_plain_path_and_tracks_updates():
    """Cached weight panels (nn.Parameter weights) give bit-identical results to the per-launch regroup, and follow
    weight updates made through torch (version counter) or announced with invalidate_panels."""
    ops = _ops()
    for kind in ('conv', 'deconv'):
        x = _rand(2, 32, 12, 20, seed=1).to(DEV).requires_grad_(True)
        if kind == 'conv':
            w0 = _rand(48, 32, 3, 3, seed=2, scale=0.1).to(DEV)
            run = lambda xx, ww: ops.conv2d(xx, ww, None, 1, 1, 'reflect', 'none')  # noqa: E731
        else:
            w0 = _rand(32, 48, 3, 3, seed=2, scale=0.1).to(DEV)
            run = lambda xx, ww: ops.conv_transpose2d(xx, ww, None, 2, 1, 1, 'none')  # noqa: E731
        plain_w = w0.clone().requires_grad_(True)            # plain tensor: regrouped on every launch
        param_w = torch.nn.Parameter(w0.clone())             # Parameter: cached panels
        y0 = run(x, plain_w)
        gy = torch.randn_like(y0)
        (gx0,) = torch.autograd.grad(y0, x, gy)
        for _ in range(2):                                   # second round hits the cache
            y1 = run(x, param_w)
            (gx1,) = torch.autograd.grad(y1, x, gy)
            assert torch.equal(y0, y1) and torch.equal(gx0, gx1)
        assert len(param_w._him_panels) == 2
        with torch.no_grad():
            param_w.mul_(2.0)                                # version bump -> lazy rebuild
        assert torch.equal(run(x, param_w), run(x, (w0 * 2.0)))
        param_w.data.copy_(w0 * 3.0)                         # behind torch's back
        ops.invalidate_panels([param_w])
        y3 = run(x, param_w)
        (gx3,) = torch.autograd.grad(y3, x, gy)
        y3r = run(x, (w0 * 3.0).requires_grad_(True))
        (gx3r,) = torch.autograd.grad(y3r, x, gy)
        assert torch.equal(y3, y3r) and torch.equal(gx3, gx3r)


WINO_CASES = [
    # B, Cin, H, W, Cout, pad_mode   (threshold forced down to 16 channels so small shapes take the Winograd path)
    (2, 32, 16, 32, 32, 'reflect'),      # ResnetBlock shape family
    (2, 32, 16, 32, 48, 'zero'),         # VGG-style zero padding
    (1, 48, 9, 13, 32, 'reflect'),       # odd H and W: partial 2x2 tiles on both edges
    (3, 16, 2, 2, 16, 'reflect'),        # a single tile per image
    (1, 128, 6, 10, 128, 'reflect'),     # 128-multiple channels: Winograd weight gradient (batched NT GEMM)
    (2, 128, 8, 8, 256, 'zero'),         # Winograd weight gradient, rectangular
    (2, 32, 4, 4, 32, 'reflect'),        # folded reflect gradient: top and bottom border tiles are neighbours
    (1, 32, 4, 36, 16, 'reflect'),       # folded reflect gradient, wide plane
    (1, 16, 6, 5, 16, 'reflect'),        # even H, odd W: padded-gradient fallback
    (8, 1024, 16, 32, 1024, 'reflect'),  # THE benchmark shape: every ResnetBlock conv of config C2 (K = 9216 per output)
    (16, 512, 16, 16, 512, 'zero'),      # VGG conv4/conv5 shape of config C4 (bs 16, 512 ch @16x16)
]


@pytest.mark.parametrize('case', WINO_CASES, ids=lambda c: 'x'.join(map(str, c)))
def test_winograd_conv3x3_fwd_bwd(case):
    """Winograd F(2x2,3x3) forward / data gradient / weight gradient against the fp32 torch reference of the SAME op
    (tolerance 2e-5 of max|ref|: the transform-domain rounding is a few ulps above the direct form)."""
    ops = _ops()
    B, Cin, H, W, Cout, pm = case
    tol = 2e-5 if Cin < 512 else 5e-5          # full-size reductions (K = 4608 / 9216): the general op tolerance
    # full-size shapes: no activation -- among 4M outputs a handful land within rounding of 0, their ReLU decision flips
    # between two fp32 summation orders and moves the data gradient by O(|gy * w|) there (seen: 1e-2 of max|ref|); the
    # activation backward itself is covered by the small cases
    act = 'relu' if Cin < 512 else 'none'
    prev = ops.set_winograd_min_channels(16)
    try:
        x = _rand(B, Cin, H, W, seed=1).requires_grad_(True)
        w = _rand(Cout, Cin, 3, 3, seed=2, scale=(Cin * 9) ** -0.5).requires_grad_(True)
        b = _rand(Cout, seed=3, scale=0.1).requires_grad_(True)
        y_ref = _ref_conv(x, w, b, 1, 1, pm, act)
        gy = _rand(*y_ref.shape, seed=4)
        gx_ref, gw_ref, gb_ref = torch.autograd.grad(y_ref, (x, w, b), gy)
        xd, bd = (t.detach().to(DEV).requires_grad_(True) for t in (x, b))
        for as_param in (False, True):           # plain tensor: per-launch weight transform; Parameter: cached panel
            wd = w.detach().to(DEV).requires_grad_(True)
            if as_param:
                wd = torch.nn.Parameter(wd.detach())
            y = ops.conv2d(xd, wd, bd, 1, 1, pm, act, 0.2)
            assert_close('wino fwd', y, y_ref, rtol=tol)
            gx, gw, gb = torch.autograd.grad(y, (xd, wd, bd), gy.to(DEV))
            assert_close('wino dgrad', gx, gx_ref, rtol=tol)
            assert_close('wino wgrad', gw, gw_ref, rtol=tol)
            assert_close('wino bgrad', gb, gb_ref, rtol=tol)
    finally:
        ops.set_winograd_min_channels(prev)


WINO4_CASES = [
    # B, Cin, H, W, Cout   (frozen weights, zero padding, planes multiples of 4, >= 256 channels on both sides)
    (2, 256, 16, 32, 256),       # VGG conv3_x family: 64 tiles, padded to 128 GEMM columns
    (3, 256, 16, 24, 512),       # conv4_1 family: rectangular, 72 tiles padded to 128
    (1, 256, 8, 12, 512),        # 6 tiles: BELOW the 64-tile rule (the padded GEMMs would execute more than the direct
    (3, 512, 4, 4, 512),         # form) -- the frozen mark must NOT change the kernel selection here
    (8, 256, 64, 128, 256),      # conv3_2..3_4 at the benchmark size (C2)
    (8, 512, 32, 64, 512),       # conv4_2..4_4 at the benchmark size
    # round 6: the 128-channel shapes that moved to F(4x4) when wino4_min_c went 256 -> 128 -- K = 128
    # per position = 8 K-steps, the shortest GEMM the batched kernel ever runs
    (8, 128, 128, 256, 128),     # conv2_2 at the benchmark size
    (8, 128, 64, 128, 256),      # conv3_1 at the benchmark size
    (3, 128, 20, 36, 128),       # Cin = 128, odd tile count: 5 x 9 tiles per image = 135 tiles, padded to 256 columns
]


@pytest.mark.parametrize('case', WINO4_CASES, ids=lambda c: 'x'.join(map(str, c)))
def test_winograd_f4x4_frozen_conv_fwd_and_gated_dgrad(case):
    """Winograd F(4x4,3x3) of the frozen-weight layers (him_conv_wino4.inc: 36-position transforms + the LDS-DMA batched
    GEMM): conv + bias + ReLU forward and the ReLU-gated data gradient against the fp32 torch reference of the SAME op.
    Tolerance 3e-5 of max|ref| (F(4x4)'s transform constants cost ~3e-6 per convolution); the same call
    WITHOUT the frozen mark must take the F(2x2) / direct kernels and agree to 2e-5."""
    ops = _ops()
    B, Cin, H, W, Cout = case
    x = _rand(B, Cin, H, W, seed=1)
    xin = torch.relu(x).requires_grad_(True)                  # the layer's input is a ReLU output (VGG chain)
    w = _rand(Cout, Cin, 3, 3, seed=2, scale=(Cin * 9) ** -0.5)
    b = _rand(Cout, seed=3, scale=0.1)
    # full-size cases: no ReLU on THIS layer's output -- among 16 M outputs a handful land within rounding of 0, their ReLU
    # decision flips between two fp32 summation orders and moves the data gradient by O(|gy w|) there (as in
    # test_winograd_conv3x3_fwd_bwd); the gate on the INPUT (identical data on both sides) stays
    act = 'relu' if B * H * W < 4096 else 'none'
    y_ref = _ref_conv(xin, w, b, 1, 1, 'zero', act)
    gy = _rand(*y_ref.shape, seed=4)
    (gx_ref,) = torch.autograd.grad(y_ref, xin, gy)
    gx_ref = gx_ref * (xin.detach() > 0)                      # gate_dx: dx = (x > 0) * dgrad
    got = {}
    for frozen in (True, False):
        wd = torch.nn.Parameter(w.to(DEV), requires_grad=False)
        if frozen:
            wd._him_frozen = True
        xd = xin.detach().to(DEV).requires_grad_(True)
        y = ops.conv2d(xd, wd, b.to(DEV), 1, 1, 'zero', act, 0.2, gate_dx=True)
        (gx,) = torch.autograd.grad(y, xd, gy.to(DEV))
        assert_close('fwd frozen=%s' % frozen, y, y_ref, rtol=3e-5 if frozen else 2e-5)
        assert_close('gated dgrad frozen=%s' % frozen, gx, gx_ref, rtol=3e-5 if frozen else 2e-5)
        got[frozen] = y.detach()
    if B * (H // 4) * (W // 4) >= 64:
        assert not torch.equal(got[True], got[False]), 'the frozen mark must select the F(4x4) kernels'
        # round 6: reductions of <= 256 channels run the PERSISTENT form of the batched GEMM (two workgroups per compute unit
        # walk the tiles, the stage ring runs across tile boundaries): the same sums in the same order, bit for bit
        from neurips18_hierchical_image_manipulation_amd._cabi import ALGO_NO_BGEMM_PERSISTENT
        with ops.algo_scope(disable=ALGO_NO_BGEMM_PERSISTENT):
            wd = torch.nn.Parameter(w.to(DEV), requires_grad=False)
            wd._him_frozen = True
            xd = xin.detach().to(DEV).requires_grad_(True)
            y1 = ops.conv2d(xd, wd, b.to(DEV), 1, 1, 'zero', act, 0.2, gate_dx=True)
        assert torch.equal(y1.detach(), got[True]), 'persistent and per-tile GEMM must agree bit for bit'
    else:
        assert torch.equal(got[True], got[False]), 'F(4x4) selected for fewer than 64 tiles (him_conv_wino4.inc: wino4_shape_ok)'


def test_cached_panel_follows_the_layout_the_library_picks_for_each_shape():
    """ONE frozen VGG weight called with a shape that takes Winograd F(4x4,3x3) (36-position panel) and then
    with shapes that leave it -- the last, smaller batch of an epoch (B = 1: fewer than 64 real tiles), a plane with
    H % 4 != 0 -- and back.  The panel cache on the parameter must hold one panel per LAYOUT (him_conv2d_panel_layout): every
    call against the fp32 torch reference, forward and gated data gradient, in both orders of first use."""
    ops = _ops()
    Cin = Cout = 256
    w = _rand(Cout, Cin, 3, 3, seed=2, scale=(Cin * 9) ** -0.5)
    b = _rand(Cout, seed=3, scale=0.1)
    shapes = [(8, 16, 32), (1, 16, 32), (8, 18, 32), (8, 16, 32), (2, 8, 8)]      # (B, H, W)
    for order in (shapes, shapes[::-1]):
        wd = torch.nn.Parameter(w.to(DEV), requires_grad=False)
        wd._him_frozen = True
        layouts = set()
        for (B, H, W) in order:
            xin = torch.relu(_rand(B, Cin, H, W, seed=B + H)).requires_grad_(True)
            y_ref = _ref_conv(xin, w, b, 1, 1, 'zero', 'none')
            gy = _rand(*y_ref.shape, seed=4)
            (gx_ref,) = torch.autograd.grad(y_ref, xin, gy)
            gx_ref = gx_ref * (xin.detach() > 0)
            xd = xin.detach().to(DEV).requires_grad_(True)
            y = ops.conv2d(xd, wd, b.to(DEV), 1, 1, 'zero', 'none', 0.2, gate_dx=True)
            (gx,) = torch.autograd.grad(y, xd, gy.to(DEV))
            assert_close('fwd %s' % ((B, H, W),), y, y_ref, rtol=3e-5)
            assert_close('gated dgrad %s' % ((B, H, W),), gx, gx_ref, rtol=3e-5)
            d = ops._conv_desc(xd, wd, 1, 1, 0, 0, 0.2, frozen=True)
            layouts.add(int(ops.lib.him_conv2d_panel_layout(ctypes.byref(d), 0)))
        assert 4 in layouts and len(layouts) >= 2, 'the shapes must cross the F(4x4) boundary: %s' % layouts
        assert len(wd._him_panels) >= 3, 'one cached panel per (kind, layout): %s' % [k[:3] for k in wd._him_panels]


def test_conv_source_tensor_above_2_gib_is_sliced_along_the_batch():
    """Buffer-resource addressing caps ONE launch's source tensor at 2 GiB (31-bit byte offsets); C2 at >= 64 images per
    GPU crosses it on the 64-channel full-resolution planes.  The dispatch then slices the batch: forward and
    data gradient of a 66 x 64 x 256 x 512 input (2.2 GiB) against the torch reference on three of its images."""
    ops = _ops()
    B, C, H, W = 66, 64, 256, 512
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(B, C, H, W, device=DEV, generator=g).requires_grad_(True)
    w = (_rand(C, C, 3, 3, seed=2) * (C * 9) ** -0.5).to(DEV)
    assert x.numel() * 4 >= (1 << 31)
    y = ops.conv2d(x, w, None, 1, 1, 'zero', 'none', 0.2)
    gy = torch.randn(y.shape, device=DEV, generator=g)
    (gx,) = torch.autograd.grad(y, x, gy)
    for b in (0, 31, 32, 65):
        xr = x.detach()[b:b + 1].cpu().requires_grad_(True)
        yr = F.conv2d(xr, w.cpu(), None, 1, 1)
        (gr,) = torch.autograd.grad(yr, xr, gy[b:b + 1].cpu())
        assert_close('fwd image %d' % b, y[b:b + 1], yr, rtol=2e-5)
        assert_close('dgrad image %d' % b, gx[b:b + 1], gr, rtol=2e-5)


def test_library_is_reentrant():
    """include/him.h "Algorithm selection": two host threads drive the C ABI concurrently -- each on its own HIP stream,
    one with the Winograd forms ON (threshold 16 channels), the other with every Winograd form OFF -- through the
    per-thread HimAlgo of ops.current_algo().  Every result must be bit-identical to the same thread's setting run alone
    (a process-global switch, as round 3's him_set_winograd_min_channels was, would mix the two kernel families)."""
    import threading
    ops = _ops()
    x = _rand(2, 64, 16, 32, seed=1).to(DEV)
    w = _rand(64, 64, 3, 3, seed=2, scale=(64 * 9) ** -0.5).to(DEV)
    gy = _rand(2, 64, 16, 32, seed=4).to(DEV)

    def one(setting):
        prev = ops.set_winograd_min_channels(setting)
        try:
            xd, wd = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
            y = ops.conv2d(xd, wd, None, 1, 1, 'reflect', 'none', 0.2)
            gx, gw = torch.autograd.grad(y, (xd, wd), gy)
            return y.detach(), gx, gw
        finally:
            ops.set_winograd_min_channels(prev)
    alone = {s: one(s) for s in (16, 0)}
    assert not torch.equal(alone[16][0], alone[0][0]), 'the two settings must select different kernels'
    torch.cuda.synchronize()
    errors = []

    def worker(setting):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                for _ in range(25):
                    got = one(setting)
                    stream.synchronize()
                    for a, b in zip(got, alone[setting]):
                        if not torch.equal(a, b):
                            errors.append('setting %d: result differs from the single-threaded run' % setting)
                            return
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))
    ts = [threading.Thread(target=worker, args=(s,)) for s in (16, 0)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    torch.cuda.synchronize()
    assert not errors, errors
    assert ops.resolved_algo()['wino_min_c'] == 256          # the main thread's HimAlgo was never touched


@pytest.mark.parametrize('case', [(2, 35, 3, 24, 40, 64, 7, 1, 3, 'reflect', False), (2, 35, 3, 24, 40, 64, 7, 1, 3, 'reflect', True),
                                  (2, 35, 6, 22, 38, 16, 4, 2, 2, 'zero', True), (2, 49, 0, 20, 24, 32, 7, 1, 3, 'reflect', True)],
                         ids=lambda c: 'x'.join(map(str, c)))
def test_onehot_weight_gradient_in_two_parts_equals_the_one_call_form(case):
    """him_conv2d_onehot_bwd_weight_part: the label-id slice and the dense slice (+ bias gradient) of the weight gradient as
    two launches -- on two streams in the trainer (config.SCHED.stem_wgrad_fork) -- write disjoint elements: their union is
    BIT-identical to the one-call form, overwriting and accumulating, with the dense channels inside the concatenation or as
    their own tensor, at stride 2 (first PatchGAN conv) and without dense channels."""
    import ctypes
    ops = _ops()
    from neurips18_hierchical_image_manipulation_amd._cabi import ONEHOT_PART_IDS, ONEHOT_PART_DENSE
    B, NC, Cd, H, W, Cout, k, stride, pad, pm, dense_only = case
    g = torch.Generator().manual_seed(23)
    coarse = torch.randint(0, NC, (B, 1, (H + 3) // 4, (W + 3) // 4), generator=g)
    label = coarse.repeat_interleave(4, 2).repeat_interleave(4, 3)[:, :, :H, :W].clone().float().to(DEV)
    dense = _rand(B, Cd, H, W, seed=5).to(DEV) if Cd else None
    w = _rand(Cout, NC + Cd, k, k, seed=2, scale=0.05).to(DEV)
    d = ops._ids_conv_desc(label, w, stride, pad, ops.PAD_REFLECT if pm == 'reflect' else ops.PAD_ZERO, ops.ACT_NONE, 0.0)
    if dense_only:
        x = dense
    else:
        onehot = torch.zeros(B, NC, H, W, device=DEV).scatter_(1, label.long(), 1.0)
        x = torch.cat([onehot, dense], 1).contiguous()
    dy = _rand(B, Cout, d.OH, d.OW, seed=4).to(DEV)
    nb = ops.lib.him_conv2d_onehot_bwd_weight_ws(ctypes.byref(d), NC)
    assert nb > 0
    ws1, ws2 = torch.empty(nb, dtype=torch.uint8, device=DEV), torch.empty(nb, dtype=torch.uint8, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    p = lambda t: 0 if t is None else t.data_ptr()       # noqa: E731
    for acc in (0, 1):
        base_w, base_b = _rand(*w.shape, seed=7).to(DEV), _rand(Cout, seed=8).to(DEV)
        dw1, db1, dw2, db2 = base_w.clone(), base_b.clone(), base_w.clone(), base_b.clone()
        ops.lib.him_conv2d_onehot_bwd_weight_part(ctypes.byref(d), p(label), NC, p(x), int(dense_only), p(dy), p(dw1), p(db1), acc,
                                                  p(ws1), nb, ONEHOT_PART_IDS | ONEHOT_PART_DENSE, st)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        ops.lib.him_conv2d_onehot_bwd_weight_part(ctypes.byref(d), p(label), NC, p(x), int(dense_only), p(dy), p(dw2), p(db2), acc,
                                                  p(ws1), nb, ONEHOT_PART_DENSE, st)
        ops.lib.him_conv2d_onehot_bwd_weight_part(ctypes.byref(d), p(label), NC, p(x), int(dense_only), p(dy), p(dw2), 0, acc,
                                                  p(ws2), nb, ONEHOT_PART_IDS, side.cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(dw1, dw2) and torch.equal(db1, db2), 'accumulate=%d' % acc
        if acc == 0:
            one = (ops.lib.him_conv2d_onehot_bwd_weight_dense if dense_only else ops.lib.him_conv2d_onehot_bwd_weight)
            dw3, db3 = base_w.clone(), base_b.clone()
            one(ctypes.byref(d), p(label), NC, p(x), p(dy), p(dw3), p(db3), 0, p(ws1), nb, st)
            torch.cuda.synchronize()
            assert torch.equal(dw1, dw3) and torch.equal(db1, db3)
    with pytest.raises(ops.HimError):
        ops.lib.him_conv2d_onehot_bwd_weight_part(ctypes.byref(d), p(label), NC, p(x), int(dense_only), p(dy), p(dw1), 0, 0,
                                                  p(ws1), nb, 4, st)


ONEHOT_CASES = [
    # B, NC, Cdense, H, W, Cout, k, pad_mode
    (2, 35, 3, 24, 40, 64, 7, 'reflect'),     # GlobalGenerator stem (one-hot 35 + cond image 3)
    (2, 49, 0, 20, 24, 32, 7, 'reflect'),     # two-stream label encoder stem, largest table (49 x 49 x 16 floats)
    (1, 35, 4, 17, 23, 16, 3, 'zero'),        # zero padding, odd sizes, one-hot + edge + image
    (3, 5, 2, 9, 70, 16, 5, 'reflect'),       # rows longer than a wave: several class groups per wave
]


@pytest.mark.parametrize('case', ONEHOT_CASES, ids=lambda c: 'x'.join(map(str, c)))
def test_onehot_stem_conv_matches_dense_conv(case):
    """The label-id evaluation of conv([one-hot | dense]) against the fp32 torch conv on the materialised one-hot
    tensor: forward, weight and bias gradients; run-to-run deterministic.  Includes ids outside [0, NC)."""
    ops = _ops()
    B, NC, Cd, H, W, Cout, k, pm = case
    g = torch.Generator().manual_seed(11)
    # piecewise-constant label map with a few isolated pixels and one out-of-range id
    coarse = torch.randint(0, NC, (B, 1, (H + 3) // 4, (W + 3) // 4), generator=g)
    label = coarse.repeat_interleave(4, 2).repeat_interleave(4, 3)[:, :, :H, :W].clone()
    label[:, :, 1::5, 2::7] = torch.randint(0, NC, label[:, :, 1::5, 2::7].shape, generator=g)
    label[0, 0, 0, 0] = NC                      # invalid id: all-zero one-hot column
    label = label.float()
    onehot = torch.zeros(B, NC, H, W)
    valid = (label >= 0) & (label < NC)
    onehot.scatter_(1, label.clamp(0, NC - 1).long(), valid.float())
    dense = _rand(B, Cd, H, W, seed=5) if Cd else torch.zeros(B, 0, H, W)
    x = torch.cat([onehot, dense], 1)
    w = _rand(Cout, NC + Cd, k, k, seed=2, scale=0.05).requires_grad_(True)
    b = _rand(Cout, seed=3, scale=0.1).requires_grad_(True)
    y_ref = _ref_conv(x, w, b, 1, k // 2, pm, 'none')
    gy = _rand(*y_ref.shape, seed=4)
    gw_ref, gb_ref = torch.autograd.grad(y_ref, (w, b), gy)

    xd = ops.mark_onehot(x.to(DEV), label.to(DEV), NC)
    wd, bd = (t.detach().to(DEV).requires_grad_(True) for t in (w, b))
    y = ops.conv2d(xd, wd, bd, 1, k // 2, pm, 'none')
    assert y.grad_fn.__class__.__name__.startswith('_OneHotConv2d')
    assert_close('onehot conv fwd', y, y_ref, rtol=2e-5)
    gw, gb = torch.autograd.grad(y, (wd, bd), gy.to(DEV))
    assert_close('onehot conv wgrad', gw, gw_ref, rtol=2e-5)
    assert_close('onehot conv bgrad', gb, gb_ref, rtol=2e-5)
    y2 = ops.conv2d(xd, wd, bd, 1, k // 2, pm, 'none')
    gw2, _ = torch.autograd.grad(y2, (wd, bd), gy.to(DEV))
    assert torch.equal(y, y2) and torch.equal(gw, gw2), 'one-hot stem must be run-to-run deterministic'


def test_compact_inputs_uint8_labels_and_device_masked_image():
    """(1) uint8 id maps are widened on the device and drive encode_input exactly like the float maps of
    data/segmentation_dataset.py:82; (2) get_masked_image (data/base_dataset.py:342-357) on the device for a batch of
    boxes against the golden vectors of the real reference function (tests/golden/data_ops.npz)."""
    from util import load_golden
    ops = _ops()
    g = load_golden('data_ops')
    image, bbox = torch.from_numpy(g['image']).to(DEV), torch.from_numpy(g['bbox']).to(DEV)
    for fill in (0, 34):
        mask, obj, ctx = ops.get_masked_image(image, bbox, fill)
        assert torch.equal(mask.cpu(), torch.from_numpy(g['mask_%d' % fill]))
        assert torch.equal(obj.cpu(), torch.from_numpy(g['obj_%d' % fill]))
        assert torch.equal(ctx.cpu(), torch.from_numpy(g['ctx_%d' % fill]))
    ids = torch.randint(0, 35, (2, 1, 16, 24), generator=torch.Generator().manual_seed(3), dtype=torch.uint8)
    assert torch.equal(ops.widen_u8(ids.to(DEV)).cpu(), ids.float())
    img = torch.rand(2, 3, 16, 24, generator=torch.Generator().manual_seed(4)) * 2 - 1
    m = torch.zeros(2, 1, 16, 24)
    m[:, :, 4:12, 6:18] = 1
    bufs = []
    for lab in (ids.to(DEV), ids.float().to(DEV)):
        lab = ops.widen_u8(lab) if lab.dtype == torch.uint8 else lab
        buf, n_label, n_cond = ops.encode_channels(lab, None, img.to(DEV), m.to(DEV), 35, False)
        bufs.append(buf)
    assert torch.equal(bufs[0], bufs[1]) and bufs[0].shape[1] == 38


@pytest.mark.parametrize('shape', [(2, 3, 64, 96), (1, 3, 48, 80)], ids=str)
def test_vgg_loss_gated_relu_backward_is_identical(shape):
    """VGGLoss with every ReLU backward folded into the gradient PRODUCERS (next conv's data-gradient epilogue -- the fused
    Winograd kernel's gate or the gate pass behind the other kernels --, the pool's backward, the L1 backward) against the
    plain form with one activation-backward pass per layer: same loss, bit-identical image gradient; and both against the
    CPU restatement of the reference's VGGLoss."""
    from neurips18_hierchical_image_manipulation_amd.models import losses
    from oracle import ref_cpu
    torch.manual_seed(3)
    crit = losses.VGGLoss().to(DEV)
    x = _rand(*shape, seed=11)
    y = _rand(*shape, seed=12)
    got = {}
    from neurips18_hierchical_image_manipulation_amd import config
    for gated in (True, False):
        with config.schedule(vgg_gated=gated):
            xd = x.to(DEV).requires_grad_(True)
            loss = crit(xd, y.to(DEV))
            (gx,) = torch.autograd.grad(loss, xd)
            got[gated] = (float(loss), gx.cpu())
    assert got[True][0] == got[False][0]
    assert torch.equal(got[True][1], got[False][1])
    vgg = ref_cpu.Vgg19()
    vgg.load_state_dict({k: v.cpu() for k, v in crit.vgg.state_dict().items()})
    xr = x.clone().requires_grad_(True)
    lr = ref_cpu.vgg_loss(vgg, xr, y)
    (gr,) = torch.autograd.grad(lr, xr)
    assert abs(got[True][0] - float(lr)) <= 1e-5 * abs(float(lr))
    # 13 fp32 layers deep with sign() / argmax / ReLU decisions on the way: single elements flip, the field agrees
    rel = float((got[True][1].double() - gr.double()).norm() / gr.double().norm())
    assert rel < 2e-2, rel


def test_gated_ops_match_plain_ops():
    """The three gate carriers one by one against 'plain op, then ReLU mask'."""
    ops = _ops()
    # data gradient with the gate (fused-Winograd shape and a direct-form shape)
    for (B, C, H, W, Co, k, p) in ((2, 64, 16, 24, 64, 3, 1), (2, 24, 9, 11, 40, 3, 1), (1, 16, 12, 12, 8, 4, 2)):
        x = torch.relu(_rand(B, C, H, W, seed=5)).to(DEV).requires_grad_(True)
        w = _rand(Co, C, k, k, seed=6, scale=0.1).to(DEV)
        gy = None
        outs = []
        for gate in (True, False):
            y = ops.conv2d(x, w, None, 1, p, 'zero', 'none', 0.0, gate_dx=gate)
            gy = _rand(*y.shape, seed=7).to(DEV) if gy is None else gy
            (gx,) = torch.autograd.grad(y, x, gy)
            outs.append(gx)
        assert torch.equal(outs[0], outs[1] * (x.detach() > 0)), (B, C, H, W, Co, k)
    # pool
    x = torch.relu(_rand(2, 5, 12, 16, seed=8)).to(DEV).requires_grad_(True)
    x.data[:, :, :4] = 0                                       # whole windows of zeros
    gy = _rand(2, 5, 6, 8, seed=9).to(DEV)
    (g1,) = torch.autograd.grad(ops.maxpool(x, 2, relu_gate=True), x, gy)
    (g0,) = torch.autograd.grad(ops.maxpool(x, 2), x, gy)
    assert torch.equal(g1, g0 * (x.detach() > 0)) and float(g0[:, :, :4].abs().sum()) > 0
    # L1
    a = torch.relu(_rand(2, 3, 8, 8, seed=10)).to(DEV).requires_grad_(True)
    b = _rand(2, 3, 8, 8, seed=11).to(DEV)
    (g1,) = torch.autograd.grad(ops.l1_weighted_sum([(a, b)], [0.5], gate_relu=True), a)
    (g0,) = torch.autograd.grad(ops.l1_weighted_sum([(a, b)], [0.5]), a)
    assert torch.equal(g1, g0 * (a.detach() > 0))


@pytest.mark.parametrize('shape', [(2, 640, 8, 12), (1, 1024, 16, 16)], ids=str)
def test_fused_resnet_block_matches_layerwise_path_and_torch(shape):
    """ResnetBlock with the InstanceNorms fused into the Winograd transforms (include/him.h "ResnetBlock", reference
    models/layer_util.py:333-378) against (1) the layer-by-layer HIP path (same GEMMs and transform arithmetic: only the
    order of the plane-statistics sums differs) and (2) the torch CPU block -- forward, input gradient, both weight
    gradients."""
    import torch.nn as tnn
    from neurips18_hierchical_image_manipulation_amd import nn as hn
    ops = _ops()
    B, Cn, H, W = shape
    blk = hn.ResnetBlock(Cn)
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():
        for conv in (blk.conv_block[1], blk.conv_block[5]):
            conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (Cn * 9) ** -0.5)
            conv.bias.copy_(torch.randn(Cn, generator=g) * 0.1)
    ref = tnn.Sequential(tnn.ReflectionPad2d(1), tnn.Conv2d(Cn, Cn, 3), tnn.InstanceNorm2d(Cn), tnn.ReLU(),
                         tnn.ReflectionPad2d(1), tnn.Conv2d(Cn, Cn, 3), tnn.InstanceNorm2d(Cn))
    with torch.no_grad():
        ref[1].weight.copy_(blk.conv_block[1].weight)
        ref[1].bias.copy_(blk.conv_block[1].bias)
        ref[5].weight.copy_(blk.conv_block[5].weight)
        ref[5].bias.copy_(blk.conv_block[5].bias)
    x = _rand(B, Cn, H, W, seed=5).requires_grad_(True)
    gy = _rand(B, Cn, H, W, seed=6)
    y_ref = x + ref(x)
    gx_ref, gw1_ref, gw2_ref = torch.autograd.grad(y_ref, (x, ref[1].weight, ref[5].weight), gy)
    blk.to(DEV)
    w1, w2 = blk.conv_block[1].weight, blk.conv_block[5].weight

    from neurips18_hierchical_image_manipulation_amd import config

    def run(fused):
        with config.schedule(resblock_fused=fused):
            xd = x.detach().to(DEV).requires_grad_(True)
            assert ops.resblock_supported(xd, w1, w2) == fused
            y = blk(xd)
            return (y,) + torch.autograd.grad(y, (xd, w1, w2), gy.to(DEV))
    fused, plain = run(True), run(False)
    for name, a, b in zip(('out', 'dx', 'dw1', 'dw2'), fused, plain):
        assert_close('fused vs layer-wise ' + name, a, b, rtol=2e-5)
    assert_close('block fwd', fused[0], y_ref, rtol=1e-4)
    # a handful of the 1e5 ReLU inputs sit within rounding of zero and flip between two fp32 summation orders
    assert_close('block dx', fused[1], gx_ref, rtol=2e-3)
    assert_close('block dw1', fused[2], gw1_ref, rtol=2e-3)
    assert_close('block dw2', fused[3], gw2_ref, rtol=2e-3)


def test_cond_image_pair_matches_concatenated_discriminator_input():
    """ops.CondImage (condition and image kept apart: pooled condition cached, first-conv gradient returned for the image
    channels only) against the reference's concatenated input (pix2pixHD_condImg_model.py:176-182 + the per-scale
    AvgPool2d of Discriminator_NET.py:47-57): same kernels on the same values -> identical features, image gradient and
    parameter gradients."""
    from neurips18_hierchical_image_manipulation_amd.models.Discriminator_NET import MultiscaleDiscriminator
    ops = _ops()
    torch.manual_seed(3)
    netD = MultiscaleDiscriminator(9 + 3, ndf=16, n_layers=2, num_D=3).to(DEV)
    cond = _rand(2, 9, 40, 56, seed=1).to(DEV)
    image = _rand(2, 3, 40, 56, seed=2).to(DEV)

    def run(split):
        img = image.clone().requires_grad_(True)
        for p in netD.parameters():
            p.grad = None
        x = ops.CondImage(cond.clone(), img) if split else ops.cat_channels([cond, img])
        feats = netD(x)
        loss = sum(f.square().mean() for scale in feats for f in scale)
        loss.backward()
        torch.cuda.synchronize()
        return [f.detach() for scale in feats for f in scale], img.grad, [None if p.grad is None else p.grad.clone() for p in netD.parameters()]
    fa, ga, pa = run(True)
    fb, gb, pb = run(False)
    for a, b in zip(fa, fb):
        assert torch.equal(a, b)
    assert_close('image gradient', ga, gb, rtol=1e-6)
    for (name, _), a, b in zip(netD.named_parameters(), pa, pb):
        assert (a is None) == (b is None)       # biases in front of an InstanceNorm carry no gradient on either path
        if a is not None:
            assert_close('grad ' + name, a, b, rtol=1e-6)


def test_multi_pair_l1_is_bit_identical_to_the_single_pair_kernels():
    """him_l1_multi_fwd / _bwd (all feature-matching / VGG terms of reference models/losses.py:47-60 and
    pix2pixHD_condImg_model.py:235-242 in three launches) against him_l1_mean_fwd / _bwd pair by pair: same work split
    and summation order -> torch.equal; 18 pairs (two chunks of the 16-slot launch table), ragged and unaligned sizes,
    one pair without a gradient."""
    import ctypes
    from neurips18_hierchical_image_manipulation_amd._cabi import lib
    sizes = [1, 3, 4, 1000, 1023, 4096, 65537, 1 << 20, (1 << 20) + 4, 777, 12, 256, 257, 8192 * 256 + 5, 5, 64, 100000, 31]
    a = [_rand(n, seed=2 * i).to(DEV) for i, n in enumerate(sizes)]
    b = [_rand(n, seed=2 * i + 1).to(DEV) for i, n in enumerate(sizes)]
    a[3] = a[3].relu()                       # exact zeros: sign(0) = 0 and the ReLU gate
    b[3] = a[3].clone()
    n = len(sizes)
    st = torch.cuda.current_stream().cuda_stream
    ws1 = torch.empty(lib.him_reduce_ws(1) // 4, device=DEV)
    single = torch.empty(n, device=DEV)
    for i in range(n):
        lib.him_l1_mean_fwd(a[i].data_ptr(), b[i].data_ptr(), sizes[i], single.data_ptr() + 4 * i, ws1.data_ptr(),
                            ws1.numel() * 4, st)
    pa = (ctypes.c_void_p * n)(*[t.data_ptr() for t in a])
    pb = (ctypes.c_void_p * n)(*[t.data_ptr() for t in b])
    pn = (ctypes.c_size_t * n)(*sizes)
    wsm = torch.empty(lib.him_l1_multi_ws(n) // 4, device=DEV)
    multi = torch.empty(n, device=DEV)
    lib.him_l1_multi_fwd(pa, pb, pn, n, multi.data_ptr(), wsm.data_ptr(), wsm.numel() * 4, st)
    assert torch.equal(single, multi)
    ref = torch.stack([(x.double() - y.double()).abs().mean() for x, y in zip(a, b)]).float()
    assert_close('multi l1 vs float64', multi, ref, rtol=1e-5)
    g = _rand(n, seed=99).to(DEV)
    for gate in (0, 2):
        d1 = [torch.empty_like(t) for t in a]
        d2 = [torch.full_like(t, float('nan')) for t in a]
        for i in range(n):
            lib.him_l1_mean_bwd(a[i].data_ptr(), b[i].data_ptr(), sizes[i], g.data_ptr() + 4 * i, d1[i].data_ptr(), gate, st)
        pd = (ctypes.c_void_p * n)(*[0 if i == 7 else t.data_ptr() for i, t in enumerate(d2)])
        lib.him_l1_multi_bwd(pa, pb, pn, n, g.data_ptr(), pd, gate, st)
        torch.cuda.synchronize()
        for i in range(n):
            if i == 7:
                assert torch.isnan(d2[i]).all()      # no gradient tensor: untouched
            else:
                assert torch.equal(d1[i], d2[i]), i


CONV_IN_CASES = [
    # B, Cin, H, W, Cout, k, stride, pad, pad_mode, act, residual     (every one a split-K launch: few output tiles)
    (8, 64, 129, 257, 128, 4, 2, 2, 'zero', 'lrelu', False),    # PatchGAN scale-0 layer 1 at C2: 65x129 planes (odd: scalar streams)
    (8, 128, 65, 129, 256, 4, 2, 2, 'zero', 'lrelu', False),    # layer 2: 33x65 planes (register-cached, 256 threads per plane)
    (8, 256, 33, 65, 512, 4, 1, 2, 'zero', 'lrelu', False),     # layer 3 (stride 1)
    (8, 512, 32, 64, 1024, 3, 2, 1, 'zero', 'relu', False),     # generator down-conv 4: 16x32 planes (one wave per plane)
    (1, 256, 128, 256, 128, 3, 2, 1, 'zero', 'relu', False),    # 64x128 planes (float4 streams, 256 threads per plane)
    (2, 128, 24, 40, 96, 3, 1, 1, 'reflect', 'none', True),     # ResnetBlock tail shape: reflect pad, residual, no activation
]


@pytest.mark.parametrize('case', CONV_IN_CASES, ids=lambda c: 'x'.join(map(str, c)))
def test_conv_instancenorm_act_block_is_bit_identical_to_the_two_launch_path(case):
    """him_conv2d_in_act_fwd (Conv2d -> InstanceNorm -> activation as one call; split-K layers hand their slabs to the
    InstanceNorm kernel, reference blocks models/Pix2Pix_NET.py:74-92, models/Discriminator_NET.py:64-96) against
    (1) conv2d + instance_norm of the same library -- forward, input gradient, weight and bias gradient BIT-identical
    (same sums in the same order) -- and (2) the torch fp32 reference of the block."""
    import ctypes
    from neurips18_hierchical_image_manipulation_amd import config
    from neurips18_hierchical_image_manipulation_amd._cabi import lib
    ops = _ops()
    B, Cin, H, W, Cout, k, stride, pad, pad_mode, act, with_res = case
    x = _rand(B, Cin, H, W, seed=1)
    w = _rand(Cout, Cin, k, k, seed=2, scale=(Cin * k * k) ** -0.5)
    b = _rand(Cout, seed=3, scale=0.1)
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = _rand(B, Cout, OH, OW, seed=5) if with_res else None
    gz = _rand(B, Cout, OH, OW, seed=4)
    d = ops._conv_desc(x.to(DEV), w.to(DEV), stride, pad, ops.PAD_REFLECT if pad_mode == 'reflect' else ops.PAD_ZERO,
                       ops.ACT_NONE, 0.0)
    assert lib.him_conv2d_in_act_fused(ctypes.byref(d)) == 1, 'case is not a split-K launch: nothing fused is tested'
    got = {}
    for fused in (True, False):
        with config.schedule(conv_in_fused=fused):
            xd = x.to(DEV).requires_grad_(True)
            wd = torch.nn.Parameter(w.to(DEV))
            bd = torch.nn.Parameter(b.to(DEV))
            rd = res.to(DEV).requires_grad_(True) if with_res else None
            z = ops.conv2d_in_act(xd, wd, bd, stride, pad, pad_mode, 1e-5, act, 0.2, rd)
            gs = torch.autograd.grad(z, [xd, wd, bd] + ([rd] if with_res else []), gz.to(DEV))
            got[fused] = [z.detach()] + [g.detach() for g in gs]
    for name, a, c in zip(['z', 'dx', 'dw', 'db', 'dres'], got[True], got[False]):
        assert torch.equal(a, c), '%s differs between the fused call and conv2d + instance_norm (max %g)' % (
            name, float((a - c).abs().max()))
    # torch reference of the block
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = _ref_conv(xr, wr, br, stride, pad, pad_mode, 'none')
    zr = torch.nn.functional.instance_norm(y, eps=1e-5)
    zr = torch.relu(zr) if act == 'relu' else torch.nn.functional.leaky_relu(zr, 0.2) if act == 'lrelu' else zr
    if with_res:
        zr = zr + res
    gxr, gwr, _ = torch.autograd.grad(zr, [xr, wr, br], gz)
    assert_close('z', got[True][0], zr, rtol=2e-5)
    # gradients in relative L2: among ~1e6 normalised values a few land within rounding of 0, their ReLU / LeakyReLU
    # decision differs between two fp32 summation orders and moves the gradient by O(|gz w|) in that neighbourhood (as in
    # test_winograd_conv3x3_fwd_bwd) -- a handful of elements, invisible in the norm
    for name, a, r in (('dx', got[True][1], gxr), ('dw', got[True][2], gwr)):
        rel = float((a.double().cpu() - r.double()).norm() / r.double().norm())
        assert rel < 2e-3, '%s: relative L2 distance from the torch block %.3e' % (name, rel)


def test_piecewise_adam_step_is_bit_identical_to_the_whole_step():
    """FusedAdam.begin_step / step_range / step (the generator's optimizer step carried out bucket by bucket during its
    backward pass, config.SCHED.adam_chunked) against one whole-arena step(): parameters, both moments and the step count
    bit-identical over three steps, with two hyper-parameter groups and pieces that cut across them."""
    from neurips18_hierchical_image_manipulation_amd.optim import FusedAdam
    shapes = [(64, 38, 7, 7), (64,), (128, 64, 3, 3), (128,), (3, 64, 7, 7), (1000,), (17,)]

    def make():
        ps = [torch.nn.Parameter(_rand(*s, seed=10 + i, scale=0.02).to(DEV)) for i, s in enumerate(shapes)]
        groups = [dict(params=ps[:3], lr=2e-4), dict(params=ps[3:], lr=5e-5)]
        return ps, FusedAdam(groups, lr=2e-4, betas=(0.5, 0.999))

    pa, oa = make()
    pb, ob = make()
    offs = oa.arena.offsets
    for step in range(3):
        for i, (a, b) in enumerate(zip(pa, pb)):
            g = _rand(*a.shape, seed=100 * step + i).to(DEV)
            a.grad.copy_(g)
            b.grad.copy_(g)
        oa.step()
        ob.begin_step()
        ob.step_range(offs[4], oa.arena.total)           # the LAST parameters first, as the backward pass delivers them
        ob.step_range(offs[2], offs[4])                  # cuts across the two lr groups
        ob.step()                                        # closes the step: whatever is left ([0, offs[2]))
        torch.cuda.synchronize()
        assert oa.step_count == ob.step_count == step + 1
        assert torch.equal(oa.arena.data, ob.arena.data), 'parameters differ after step %d' % step
        assert torch.equal(oa.exp_avg, ob.exp_avg) and torch.equal(oa.exp_avg_sq, ob.exp_avg_sq)
    with pytest.raises(RuntimeError):
        ob.step_range(0, 10)                             # outside begin_step() ... step()


@pytest.mark.parametrize('case', [(4, 3, 256, 512, 32, 7), (2, 4, 250, 516, 16, 3), (8, 2, 128, 256, 16, 7)],
                         ids=lambda c: 'x'.join(map(str, c)))
def test_reflection_padded_few_channel_forward_on_the_tiled_kernel(case):
    """The dense channels of a generator stem (ReflectionPad2d(3) + Conv2d(3, ngf, 7), models/Pix2Pix_NET.py:74) ran on the
    generic implicit-GEMM kernel (K = 147: 0.55 ms at C2); the tiled few-channel kernel now mirrors its patch loads
    (HIM_ALGO_NO_FEWIN_REFLECT switches back).  Against the fp32 torch conv, and bit-identical to the generic kernel."""
    ops = _ops()
    from neurips18_hierchical_image_manipulation_amd._cabi import ALGO_NO_FEWIN_REFLECT
    B, C, H, W, Cout, k = case
    x, w, b = _rand(B, C, H, W, seed=1), _rand(Cout, C, k, k, seed=2, scale=0.05), _rand(Cout, seed=3, scale=0.1)
    ref = _ref_conv(x, w, b, 1, k // 2, 'reflect', 'none')
    got = {}
    for off in (0, ALGO_NO_FEWIN_REFLECT):
        with ops.algo_scope(disable=off):
            got[off] = ops.conv2d(x.to(DEV), w.to(DEV), b.to(DEV), 1, k // 2, 'reflect', 'none')
    assert_close('reflect few-channel fwd (tiled)', got[0], ref, rtol=2e-5)
    assert_close('reflect few-channel fwd (generic)', got[ALGO_NO_FEWIN_REFLECT], ref, rtol=2e-5)
    # same products, same order (channel, tap row, tap column; one fp32 accumulator per output -- the fp32 MFMA adds its two
    # k-terms one after the other): the tiled kernel reproduces the generic one bit for bit, so switching kernels moved no
    # parity figure (which kernel ran is visible in a trace only: gconv_fewin_tiled_kernel<7, false, false>, 0.31 vs 0.55 ms)
    assert torch.equal(got[0], got[ALGO_NO_FEWIN_REFLECT])


@pytest.mark.parametrize('n', [1, 3, 8, 9, 19])
def test_scalar_loss_arithmetic_in_one_launch_matches_the_torch_chain(n):
    """ops.lincomb = the trainer's scalar loss arithmetic (train_mask2image.py:70-71, the per-scale GAN sums of
    models/losses.py:30-38): scale * (w0 t0 + w1 t1 + ...) left to right in fp32.  him_lincomb_* take 8 terms per launch;
    longer sums (--num_D > 8) fold 8 at a time.  Value and every term's gradient bit-identical to the chain of
    one-element fp32 torch ops on the host."""
    ops = _ops()
    g = torch.Generator().manual_seed(n)
    vals = [(torch.randn((), generator=g) * 3).requires_grad_(True) for _ in range(n)]
    ws = [1.0 if i % 3 == 0 else float(torch.randn((), generator=g)) for i in range(n)]
    acc = 0
    for w, t in zip(ws, vals):
        acc = acc + (t if w == 1.0 else t * w)
    ref = acc * 0.5
    ref.backward()
    dv = [v.detach().to(DEV).requires_grad_(True) for v in vals]
    out = ops.lincomb(dv, ws, scale=0.5)
    out.backward()
    assert float(out) == float(ref), (float(out), float(ref))
    for i, (a, b) in enumerate(zip(dv, vals)):
        assert float(a.grad) == float(b.grad), (i, float(a.grad), float(b.grad))


def _label_blocks(B, H, W, NC, block, seed):
    """piecewise-constant id map (blocks of ``block`` pixels, some ids outside [0, NC) -> all-zero one-hot column)."""
    g = torch.Generator().manual_seed(seed)
    small = torch.randint(0, NC, (B, 1, (H + block - 1) // block, (W + block - 1) // block), generator=g).float()
    lab = small.repeat_interleave(block, 2).repeat_interleave(block, 3)[:, :, :H, :W].contiguous()
    lab[:, :, H // 3, : W // 2] = float(NC + 3)       # ids outside the range select nothing (him_onehot semantics)
    return lab


D_IDS_CASES = [
    # B, NC, Cd (dense condition channels), H, W, Cout, label block  -- the first PatchGAN conv: 4x4, stride 2, pad 2
    (8, 35, 3, 256, 512, 64, 16),    # C2 scale 0: 41 -> 64 at 256x512 -> 129x257
    (2, 35, 4, 64, 96, 64, 5),       # edges + image condition, odd runs
    (3, 49, 0, 33, 47, 32, 1),       # no dense condition channel (--no_imgCond), odd plane, per-pixel noise labels
    (1, 8, 3, 16, 1000, 16, 7),      # wide rows (16 dy columns per lane), few classes
]


@pytest.mark.parametrize('case', D_IDS_CASES, ids=lambda c: 'x'.join(map(str, c)))
def test_first_patchgan_conv_from_label_ids_matches_the_dense_conv(case):
    """Round 5 (second half): nn.Conv2d(input_nc, ndf, 4, stride 2, padding 2) + LeakyReLU of
    Discriminator_NET.py:71-74 on cat(one-hot(label) | dense condition | image) (pix2pixHD_condImg_model.py:176-186) evaluated
    from the ids -- ops.cond_image_conv2d on an ops.LabelCond -- against F.conv2d on the materialised concatenation (fp32 CPU):
    forward, weight / bias gradient, the image's data gradient; deterministic run to run; and the materialised HIP path
    (ops.LabelCond.full -> the MFMA kernels) agrees with both."""
    ops = _ops()
    B, NC, Cd, H, W, Cout, block = case
    lab = _label_blocks(B, H, W, NC, block, seed=5)
    dense = _rand(B, Cd, H, W, seed=6) if Cd else None
    img = _rand(B, 3, H, W, seed=7).requires_grad_(True)
    Cin = NC + Cd + 3
    w = (_rand(Cout, Cin, 4, 4, seed=8) * 0.05).requires_grad_(True)
    b = (_rand(Cout, seed=9) * 0.1).requires_grad_(True)
    onehot = torch.zeros(B, NC, H, W)
    valid = (lab >= 0) & (lab < NC)
    onehot.scatter_(1, lab.clamp(0, NC - 1).long(), valid.float())
    xin = torch.cat([onehot] + ([dense] if Cd else []) + [img], 1)
    # full-size case without the LeakyReLU: among 17 M outputs a handful land within rounding of 0, their gate flips between
    # two fp32 summation orders and moves a weight gradient by 0.8 |gy| there (seen: 0.5 on an entry of 40; without the
    # activation every piece sits at 2..6e-7 of the float64 result) -- as in test_winograd_conv3x3_fwd_bwd
    act = 'lrelu' if B * H * W < 100000 else 'none'
    y_ref = F.conv2d(xin, w, b, 2, 2)
    if act == 'lrelu':
        y_ref = F.leaky_relu(y_ref, 0.2)
    gy = _rand(*y_ref.shape, seed=4)
    gw_ref, gb_ref, gi_ref = torch.autograd.grad(y_ref, (w, b, img), gy)

    def run(from_ids):
        from neurips18_hierchical_image_manipulation_amd import config
        with config.schedule(d_from_ids=from_ids):
            cond = ops.LabelCond(lab.to(DEV), NC, dense.to(DEV) if Cd else None)
            wd, bd, imd = (t.detach().to(DEV).requires_grad_(True) for t in (w, b, img))
            y = ops.cond_image_conv2d(cond, imd, wd, bd, 2, 2, act, 0.2)
            assert y.grad_fn.__class__.__name__.startswith('_IdsCondImageConv2d' if from_ids else '_CondImageConv2d')
            return (y,) + torch.autograd.grad(y, (wd, bd, imd), gy.to(DEV))

    got = run(True)
    for name, a, r in zip(('fwd', 'wgrad', 'bgrad', 'image dgrad'), got, (y_ref, gw_ref, gb_ref, gi_ref)):
        assert_close('D layer 0 from ids: ' + name, a, r, rtol=2e-5)
    again = run(True)
    assert all(torch.equal(a, c) for a, c in zip(got, again)), 'the ids path must be run-to-run deterministic'
    mat = run(False)
    for name, a, c in zip(('fwd', 'wgrad', 'bgrad', 'image dgrad'), got, mat):
        assert_close('ids vs materialised HIP path: ' + name, a, c.cpu(), rtol=2e-5)


@pytest.mark.parametrize('shape', [(8, 35, 3, 256, 512), (2, 49, 6, 33, 47), (1, 5, 0, 2, 3)], ids=str)
def test_label_cond_full_and_pooled_equal_the_materialised_tensors(shape):
    """ops.LabelCond: full() == [him_onehot | dense] and pooled() -- the one-hot channels as 3x3 class counts straight from
    the ids (him_onehot_pool3s2) -- is BIT-identical to AvgPool2d(3, 2, 1, count_include_pad=False) of the materialised
    tensor (sums of 0 / 1 are exact), on the HIP pool and on torch's; slices follow the reference's channel order."""
    ops = _ops()
    B, NC, Cd, H, W = shape
    lab = _label_blocks(B, H, W, NC, 3, seed=1)
    dense = _rand(B, Cd, H, W, seed=2) if Cd else None
    cond = ops.LabelCond(lab.to(DEV), NC, dense.to(DEV) if Cd else None)
    onehot = torch.zeros(B, NC, H, W)
    onehot.scatter_(1, lab.clamp(0, NC - 1).long(), ((lab >= 0) & (lab < NC)).float())
    full_ref = torch.cat([onehot] + ([dense] if Cd else []), 1)
    assert tuple(cond.shape) == tuple(full_ref.shape)
    assert torch.equal(cond.full().cpu(), full_ref)
    pooled = cond.pooled()
    assert torch.equal(pooled, ops.avgpool3s2(cond.full()))
    assert_close('pooled vs torch', pooled, F.avg_pool2d(full_ref, 3, 2, 1, count_include_pad=False), rtol=1e-6)
    assert torch.equal(ops.cond_pyramid(cond, 3)[2], ops.avgpool3s2(ops.avgpool3s2(cond.full())))
    if Cd:
        assert torch.equal(ops.slice_channels(cond, NC, Cd).cpu(), dense)
        head = ops.slice_channels(cond, 0, NC + 1)
        assert isinstance(head, ops.LabelCond) and torch.equal(head.full().cpu(), full_ref[:, :NC + 1])
    assert torch.equal(ops.slice_channels(cond, 1, NC - 1).cpu(), full_ref[:, 1:NC])


@pytest.mark.parametrize('shape', [(8, 64, 256, 512, 3, 7), (2, 32, 67, 130, 3, 7), (3, 16, 40, 200, 2, 3), (1, 64, 33, 64, 4, 7),
                                   (16, 16, 20, 70, 3, 7)], ids=str)
def test_reflect_fold_inside_the_few_channel_data_gradient_is_bit_identical(shape):
    """Round 5: the data gradient of a reflection-padded layer with <= 4 output channels (the generator's head,
    ReflectionPad2d(3) + Conv2d(64, 3, 7) of models/Pix2Pix_NET.py:91 -- the FIRST kernel of the generator's backward) folds the
    mirrored border strips inside the LDS-tiled kernel instead of writing the padded gradient and running reflect_fold_kernel
    over it.  Same terms, same order: torch.equal to the two-kernel form (HIM_ALGO_NO_FEWIN_FOLD), and both against the fp32
    torch reference; odd planes, a plane too small for the tiled kernel (falls back), 2..4 channels, 3x3 and 7x7."""
    ops = _ops()
    from neurips18_hierchical_image_manipulation_amd._cabi import ALGO_NO_FEWIN_FOLD
    B, Cin, H, W, Cout, k = shape
    x = _rand(B, Cin, H, W, seed=1).requires_grad_(True)
    w = _rand(Cout, Cin, k, k, seed=2, scale=(Cin * k * k) ** -0.5)
    y_ref = _ref_conv(x, w, None, 1, k // 2, 'reflect', 'none')
    gy = _rand(*y_ref.shape, seed=3)
    (gx_ref,) = torch.autograd.grad(y_ref, x, gy)
    got = {}
    for off in (0, ALGO_NO_FEWIN_FOLD):
        with ops.algo_scope(disable=off):
            xd = x.detach().to(DEV).requires_grad_(True)
            y = ops.conv2d(xd, w.to(DEV), None, 1, k // 2, 'reflect', 'none')
            (got[off],) = torch.autograd.grad(y, xd, gy.to(DEV))
        assert_close('dgrad (disable=%d)' % off, got[off], gx_ref, rtol=2e-5)
    assert torch.equal(got[0], got[ALGO_NO_FEWIN_FOLD]), 'the in-kernel fold must reproduce the fold pass bit for bit'


@pytest.mark.parametrize('case', [(8, 64, 64, 128, 64, 'zero'), (2, 128, 37, 53, 128, 'zero'), (3, 72, 16, 24, 64, 'reflect')],
                         ids=str)
def test_fused_winograd_kernel_at_80_kb_of_lds(case):
    """HimAlgo.wino_fused_chunk = 4 (round 5: 4-channel K-chunks, 80 KB of LDS, two lanes per input patch, four epilogue
    rounds -- measured slower inside the step, kept as a switch: profiles/r05_ab_log.txt): forward and zero-pad data gradient
    against the fp32 torch reference, and within 2e-6 of the shipped 8-channel-chunk kernel."""
    ops = _ops()
    B, Cin, H, W, Cout, mode = case
    x = _rand(B, Cin, H, W, seed=1).requires_grad_(True)
    w = _rand(Cout, Cin, 3, 3, seed=2, scale=(Cin * 9) ** -0.5)
    b = _rand(Cout, seed=3, scale=0.1)
    y_ref = _ref_conv(x, w, b, 1, 1, mode, 'none')
    gy = _rand(*y_ref.shape, seed=4)
    (gx_ref,) = torch.autograd.grad(y_ref, x, gy)
    got = {}
    for ck in (8, 4):
        with ops.algo_scope(wino_fused_chunk=ck, wino_fused_min_c=64, wino_fused_max_c=255, wino_min_c=256):
            xd = x.detach().to(DEV).requires_grad_(True)
            y = ops.conv2d(xd, w.to(DEV), b.to(DEV), 1, 1, mode, 'none')
            (gx,) = torch.autograd.grad(y, xd, gy.to(DEV))
        assert_close('fwd chunk %d' % ck, y, y_ref, rtol=2e-5)
        assert_close('dgrad chunk %d' % ck, gx, gx_ref, rtol=2e-5)
        got[ck] = y.detach()
    assert_close('chunk 4 vs chunk 8', got[4], got[8].cpu(), rtol=2e-6)


FUSED2_CASES = [
    # B, Cin, H, W, Cout, padding
    (8, 64, 256, 512, 64, 'zero'),       # VGG conv1_2 at the benchmark size (C2): 4096 jobs, 16 per workgroup
    (8, 64, 128, 256, 128, 'zero'),      # VGG conv2_1 at C2: two channel blocks
    (16, 64, 256, 256, 64, 'zero'),      # conv1_2 at C4 (256x256, bs 16)
    (1, 64, 128, 256, 64, 'zero'),       # C1: fewer jobs than compute units
    (2, 128, 36, 70, 192, 'reflect'),    # ragged: blocks cut by the bottom / right border, 3 channel blocks, 16 chunks
    (3, 32, 50, 36, 128, 'reflect'),     # 4 chunks (the shortest K the kernel takes), H not a multiple of 8
    (2, 72, 18, 34, 64, 'zero'),         # odd chunk count (9): jobs start on either register set / V buffer
]


@pytest.mark.parametrize('case', FUSED2_CASES, ids=lambda c: 'x'.join(map(str, c)))
def test_persistent_fused_winograd_kernel(case):
    """him_wino_fused2.inc (round 6): the persistent, wave-specialised fused Winograd kernel -- forward
    (conv + bias + ReLU / none, zero or reflection padding), the plain zero-pad data gradient and the ReLU-GATED data gradient
    (the VGG chain's form: gate tensor in the epilogue) against the fp32 torch reference of the same op, at the full benchmark
    shapes and on ragged planes; within 2e-6 of the round-2 kernel it replaces (HIM_ALGO_NO_WINO_FUSED2 selects that one:
    same panel, same products, another order of the 8 channels of a chunk) and NOT bit-equal to it (the switch selects)."""
    ops = _ops()
    from neurips18_hierchical_image_manipulation_amd._cabi import ALGO_NO_WINO_FUSED2
    B, Cin, H, W, Cout, mode = case
    x = _rand(B, Cin, H, W, seed=1)
    xin = torch.relu(x).requires_grad_(True)
    w = _rand(Cout, Cin, 3, 3, seed=2, scale=(Cin * 9) ** -0.5)
    b = _rand(Cout, seed=3, scale=0.1)
    # gradients through a layer WITHOUT an activation on its output: among the outputs a few land within rounding of 0, their
    # ReLU decision flips between two fp32 summation orders and moves the data gradient by O(|gy w|) there (seen: 4e-2 of
    # max|ref| at one element); the ReLU / LeakyReLU epilogues are compared on the forward output
    y_ref = _ref_conv(xin, w, b, 1, 1, mode, 'none')
    gy = _rand(*y_ref.shape, seed=4)
    (gx_ref,) = torch.autograd.grad(y_ref, xin, gy)
    gx_gated_ref = gx_ref * (xin.detach() > 0)
    got = {}
    for bit in (0, ALGO_NO_WINO_FUSED2):
        with ops.algo_scope(disable=bit, wino_fused_min_c=32, wino_fused_max_c=255, wino_min_c=256, wino4_min_c=-1,
                            wino_fused_chunk=8):
            wd = torch.nn.Parameter(w.to(DEV), requires_grad=False)
            with torch.no_grad():
                for act in ('relu', 'lrelu'):
                    ya = ops.conv2d(xin.detach().to(DEV), wd, b.to(DEV), 1, 1, mode, act, 0.2)
                    assert_close('fwd %s (bit %d)' % (act, bit), ya, _ref_conv(xin.detach(), w, b, 1, 1, mode, act), rtol=2e-5)
            for gate in ((False, True) if mode == 'zero' else (False,)):
                xd = xin.detach().to(DEV).requires_grad_(True)
                y = ops.conv2d(xd, wd, b.to(DEV), 1, 1, mode, 'none', 0.2, gate_dx=gate)
                (gx,) = torch.autograd.grad(y, xd, gy.to(DEV))
                assert_close('fwd (bit %d)' % bit, y, y_ref, rtol=2e-5)
                assert_close('dgrad (bit %d, gate %s)' % (bit, gate), gx, gx_gated_ref if gate else gx_ref, rtol=2e-5)
                got[(bit, gate)] = (y.detach(), gx.detach())
    for gate in ((False, True) if mode == 'zero' else (False,)):
        a, o = got[(0, gate)], got[(ALGO_NO_WINO_FUSED2, gate)]
        assert_close('fwd new vs round-2 kernel', a[0], o[0].cpu(), rtol=2e-6)
        assert_close('dgrad new vs round-2 kernel', a[1], o[1].cpu(), rtol=2e-6)
    assert not torch.equal(got[(0, False)][0], got[(ALGO_NO_WINO_FUSED2, False)][0]), 'the switch must select the other kernel'


@pytest.mark.parametrize('case', [(8, 256, 16, 32, 256, 'reflect'), (8, 256, 16, 32, 256, 'zero'), (4, 512, 32, 32, 256, 'reflect')],
                         ids=str)
def test_opt_in_f4x4_forward_of_trainable_layers(case):
    """HIM_ALGO_WINO4_TRAIN_FWD (an OPT-IN reduced-work variant, reported as its own bench line): the
    FORWARD of a trainable 3x3 stride-1 pad-1 layer with >= 256 channels as Winograd F(4x4,3x3), zero or reflection padding;
    data and weight gradients stay on the F(2x2) kernels.  Forward within 3e-5 of max|ref| (F(4x4)'s rounding), gradients
    within 2e-5, and the output differs from the default build's (the switch really selects the other kernels)."""
    ops = _ops()
    from neurips18_hierchical_image_manipulation_amd._cabi import ALGO_WINO4_TRAIN_FWD
    B, Cin, H, W, Cout, mode = case
    x = _rand(B, Cin, H, W, seed=1).requires_grad_(True)
    w = _rand(Cout, Cin, 3, 3, seed=2, scale=(Cin * 9) ** -0.5).requires_grad_(True)
    b = _rand(Cout, seed=3, scale=0.1)
    y_ref = _ref_conv(x, w, b, 1, 1, mode, 'none')
    gy = _rand(*y_ref.shape, seed=4)
    gx_ref, gw_ref = torch.autograd.grad(y_ref, (x, w), gy)
    got = {}
    for bit in (ALGO_WINO4_TRAIN_FWD, 0):
        with ops.algo_scope(disable=bit, wino_min_c=256, wino4_min_c=256):
            xd = x.detach().to(DEV).requires_grad_(True)
            wd = torch.nn.Parameter(w.detach().to(DEV))
            y = ops.conv2d(xd, wd, b.to(DEV), 1, 1, mode, 'none')
            gx, gw = torch.autograd.grad(y, (xd, wd), gy.to(DEV))
            if bit:
                d = ops._conv_desc(xd, wd, 1, 1, 1 if mode == 'reflect' else 0, 0, 0.2)
                assert int(ops.lib.him_conv2d_panel_layout(ctypes.byref(d), 0)) == 4, 'the switch must select the F(4x4) forward'
        assert_close('fwd (bit %d)' % bit, y, y_ref, rtol=3e-5 if bit else 2e-5)
        assert_close('dgrad (bit %d)' % bit, gx, gx_ref, rtol=2e-5)
        assert_close('wgrad (bit %d)' % bit, gw, gw_ref, rtol=2e-5)
        got[bit] = y.detach()
    assert not torch.equal(got[0], got[ALGO_WINO4_TRAIN_FWD])
