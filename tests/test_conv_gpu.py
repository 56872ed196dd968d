"""GPU parity: MFMA implicit-GEMM conv forward / dgrad / wgrad + stem vs a torch-CPU fp32 reference evaluated
on the same bf16-rounded operands.  Tolerance: rel-L2 <= 1e-2 (bf16 output rounding ~4e-3), stated per test."""
import numpy as np
import pytest
import torch
import torch.nn.functional as TF

from tests.util import bf16_round, nchw_to_pm, pm_to_nchw, oihw_to_ohwi, rel_l2, pack_weights

pytestmark = pytest.mark.gpu

TOL = 1e-2


def _ops():
    from basedet_amd import ops
    return ops


CASES = [
    # N, Cin, Cout, H, W, R, stride, pad
    (2, 64, 64, 20, 28, 1, 1, 0),
    (2, 64, 256, 17, 23, 1, 1, 0),
    (1, 256, 128, 24, 20, 3, 1, 1),
    (2, 128, 128, 26, 30, 3, 2, 1),
    (2, 256, 512, 16, 18, 1, 2, 0),
    (1, 256, 720, 13, 21, 3, 1, 1),     # ragged Cout tile (cls_score)
    (1, 256, 40, 13, 21, 3, 1, 1),      # padded bbox_pred
    (3, 32, 64, 9, 11, 3, 1, 1),        # BK=32 path
    (1, 2048, 256, 25, 42, 3, 2, 1),    # P6 conv
    (2, 256, 256, 22, 37, 3, 1, 1),     # staggered 256-channel patch kernel, forward and dgrad (ragged patches)
    (1, 512, 192, 9, 40, 3, 1, 1),      # ... ragged channel tile (192 of 256), 8 K blocks; dgrad 192 -> 512 = two channel tiles
    (1, 200, 136, 11, 19, 3, 1, 1),     # ... K tails: forward Cin = 200 (3 x 64 + 8), dgrad K = 136
    (2, 64, 192, 27, 31, 3, 2, 1),      # stride 2 on odd input sizes (14 x 16 outputs: ragged 4 x 8 weight-gradient patches), three channel tiles
]


@pytest.mark.parametrize("patch3x3", [1, 0, 2, 3])
@pytest.mark.parametrize("case", CASES)
def test_conv_fwd_dgrad_wgrad(case, patch3x3):
    ops = _ops()
    # patch3x3 = 1: specialised kernels (3x3 patch, streaming 1x1); 0: everything through the generic kernel;
    # 2: specialised kernels without any staggered instance (bits 6 + 9); 3: the staggered 128 / 64-channel instances everywhere (bits 6 + 8);
    ops.set_route(patch3x3={0: 0, 1: 3, 2: 3 | 64 | 512, 3: 3 | 64 | 256}[patch3x3])
    N, Cin, Cout, H, W, R, stride, pad = case
    g = torch.Generator().manual_seed(1234 + Cin + Cout + H)
    x = bf16_round(torch.randn(N, Cin, H, W, generator=g))
    w = bf16_round(torch.randn(Cout, Cin, R, R, generator=g) / np.sqrt(Cin * R * R))
    bias = torch.randn(Cout, generator=g)
    gin = ops.single(N, H, W)
    gout = gin.conv_out(R, stride, pad)
    Ho, Wo = gout.H[0], gout.W[0]
    d = ops.conv_desc(gin, gout, Cin, Cout, R, R, stride, pad)
    wf, wd = pack_weights(ops, w)
    xp = nchw_to_pm(x)
    # ---- forward: bias + residual add + relu
    res = bf16_round(torch.randn(N, Cout, Ho, Wo, generator=g))
    y = torch.empty((gout.pixels, Cout), dtype=torch.bfloat16, device="cuda")
    ops.conv2d_fwd(d, xp, wf, bias.cuda(), y, add=nchw_to_pm(res), flags=ops.EPI_RELU | ops.EPI_ADD_BEFORE)
    ref = TF.relu(TF.conv2d(x, w, bias, stride=stride, padding=pad) + res)
    got = pm_to_nchw(y, N, Ho, Wo)
    assert rel_l2(got, ref) < TOL
    # plain
    ops.conv2d_fwd(d, xp, wf, None, y)
    ref = TF.conv2d(x, w, None, stride=stride, padding=pad)
    assert rel_l2(pm_to_nchw(y, N, Ho, Wo), ref) < TOL
    # ---- dgrad with add_before + mask
    gy = bf16_round(torch.randn(N, Cout, Ho, Wo, generator=g))
    addt = bf16_round(torch.randn(N, Cin, H, W, generator=g))
    maskt = bf16_round(torch.randn(N, Cin, H, W, generator=g))
    dx = torch.empty((gin.pixels, Cin), dtype=torch.bfloat16, device="cuda")
    ops.conv2d_dgrad(d, nchw_to_pm(gy), wd, dx, add=nchw_to_pm(addt), mask=nchw_to_pm(maskt), flags=ops.EPI_ADD_BEFORE | ops.EPI_MASK)
    xr = x.clone().requires_grad_(True)
    TF.conv2d(xr, w, None, stride=stride, padding=pad).backward(gy)
    ref = (xr.grad + addt) * (maskt > 0)
    assert rel_l2(pm_to_nchw(dx, N, H, W), ref) < TOL
    # add_after variant
    ops.conv2d_dgrad(d, nchw_to_pm(gy), wd, dx, add=nchw_to_pm(addt), mask=nchw_to_pm(maskt), flags=ops.EPI_ADD_AFTER | ops.EPI_MASK)
    ref = xr.grad * (maskt > 0) + addt
    assert rel_l2(pm_to_nchw(dx, N, H, W), ref) < TOL
    # ---- wgrad (transposing LDS read and the scalar-read fallback must agree with the reference)
    wr = w.clone().requires_grad_(True)
    TF.conv2d(x, wr, None, stride=stride, padding=pad).backward(gy)
    ref_dw = oihw_to_ohwi(wr.grad)
    ws = torch.empty((ops.conv2d_wgrad_workspace_bytes(d) // 4 + 4,), dtype=torch.float32, device="cuda")
    scale = torch.rand(Cout, generator=g) + 0.5
    for use_tr in (0, 1):
        ops.set_route(wgrad=use_tr)
        dw = torch.full((Cout, R, R, Cin), 7.0, dtype=torch.float32, device="cuda")
        ops.conv2d_wgrad(d, xp, nchw_to_pm(gy), dw, ws)
        assert rel_l2(dw.cpu(), ref_dw) < 2e-3, f"use_tr={use_tr}"
        ops.conv2d_wgrad(d, xp, nchw_to_pm(gy), dw, ws, row_scale=scale.cuda(), accumulate=True)
        assert rel_l2(dw.cpu(), ref_dw * (1 + scale.view(-1, 1, 1, 1))) < 2e-3
    ops.set_route(wgrad=1)
    ops.set_route(patch3x3=3)


@pytest.mark.parametrize("chans", [(64, 72), (192, 256)])
def test_conv_multilevel_head_layout(chans):
    """Five pyramid levels through one launch (RetinaNetHead weight sharing, retina_head.py:103-112).  (192, 256): the staggered
    256-channel patch instance on a multi-segment descriptor, forward and dgrad."""
    ops = _ops()
    ops.set_route(patch3x3=3)
    N, (C, Cout) = 2, chans
    Hs, Ws = [12, 6, 3, 2, 1], [20, 10, 5, 3, 2]
    g = torch.Generator().manual_seed(7)
    geo = ops.Geom(N, Hs, Ws)
    w = bf16_round(torch.randn(Cout, C, 3, 3, generator=g) / 24)
    bias = torch.randn(Cout, generator=g)
    wf, wd = pack_weights(ops, w)
    xs = [bf16_round(torch.randn(N, C, h, ww, generator=g)) for h, ww in zip(Hs, Ws)]
    xp = torch.empty((geo.pixels, C), dtype=torch.bfloat16, device="cuda")
    xv = xp.view(N, geo.pix_per_img, C)
    for i, x in enumerate(xs):
        xv[:, geo.off[i]: geo.off[i] + Hs[i] * Ws[i]] = x.permute(0, 2, 3, 1).reshape(N, -1, C).to(torch.bfloat16).cuda()
    d = ops.conv_desc(geo, geo, C, Cout, 3, 3, 1, 1)
    y = torch.empty((geo.pixels, Cout), dtype=torch.bfloat16, device="cuda")
    ops.conv2d_fwd(d, xp, wf, bias.cuda(), y, flags=ops.EPI_RELU)
    yv = y.view(N, geo.pix_per_img, Cout).float().cpu()
    gys = []
    for i, x in enumerate(xs):
        ref = TF.relu(TF.conv2d(x, w, bias, padding=1))
        got = yv[:, geo.off[i]: geo.off[i] + Hs[i] * Ws[i]].reshape(N, Hs[i], Ws[i], Cout).permute(0, 3, 1, 2)
        assert rel_l2(got, ref) < TOL, f"level {i}"
    # wgrad + dgrad over all levels at once
    gy = bf16_round(torch.randn(N, geo.pix_per_img, Cout, generator=g))
    gyp = gy.reshape(-1, Cout).to(torch.bfloat16).cuda()
    ws = torch.empty((ops.conv2d_wgrad_workspace_bytes(d) // 4 + 4,), dtype=torch.float32, device="cuda")
    dw = torch.empty((Cout, 3, 3, C), dtype=torch.float32, device="cuda")
    ops.conv2d_wgrad(d, xp, gyp, dw, ws)
    dx = torch.empty((geo.pixels, C), dtype=torch.bfloat16, device="cuda")
    ops.conv2d_dgrad(d, gyp, wd, dx)
    ref_dw = torch.zeros(Cout, C, 3, 3)
    dxv = dx.view(N, geo.pix_per_img, C).float().cpu()
    for i, x in enumerate(xs):
        xr = x.clone().requires_grad_(True)
        wr = w.clone().requires_grad_(True)
        gl = gy[:, geo.off[i]: geo.off[i] + Hs[i] * Ws[i]].reshape(N, Hs[i], Ws[i], Cout).permute(0, 3, 1, 2)
        TF.conv2d(xr, wr, None, padding=1).backward(gl)
        ref_dw += wr.grad
        got = dxv[:, geo.off[i]: geo.off[i] + Hs[i] * Ws[i]].reshape(N, Hs[i], Ws[i], C).permute(0, 3, 1, 2)
        assert rel_l2(got, xr.grad) < TOL, f"dgrad level {i}"
    assert rel_l2(dw.cpu(), oihw_to_ohwi(ref_dw)) < 2e-3
    ops.set_route(patch3x3=3)


def test_stem_conv_and_pad_normalize():
    """bd_pad_normalize + bd_stem_conv7x7_fwd + bd_maxpool3x3s2_fwd vs conv1/bn1/relu/maxpool (resnet.py:236-241)."""
    ops = _ops()
    N, H, W = 2, 50, 70   # pads to 64 x 96
    Hp, Wp = 64, 96
    g = torch.Generator().manual_seed(3)
    img = torch.rand(N, 3, H, W, generator=g) * 255
    mean, std = [103.530, 116.280, 123.675], [57.375, 57.12, 58.395]
    xh = torch.empty((N, Hp + 6, Wp + 8, 4), dtype=torch.bfloat16, device="cuda")
    ops.pad_normalize(img.cuda(), Hp, Wp, mean, std, xh)
    from oracle import box_ops
    xo = torch.from_numpy(box_ops.data_to_input(img.numpy(), mean, std))
    # exact fp32 semantics through the NCHW variant (bit-exact: same (x-mean)/std in fp32)
    xn = torch.empty((N, 3, Hp, Wp), dtype=torch.float32, device="cuda")
    ops.pad_normalize_nchw(img.cuda(), Hp, Wp, mean, std, xn)
    assert torch.equal(xn.cpu(), xo)
    inner = xh[:, 3:3 + Hp, 4:4 + Wp, :3].float().cpu().permute(0, 3, 1, 2)
    assert torch.equal(inner, bf16_round(xo))
    assert float(xh[:, :3].float().abs().max()) == 0 and float(xh[:, :, :4].float().abs().max()) == 0
    assert float(xh[..., 3].float().abs().max()) == 0
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.05
    scale = torch.rand(64, generator=g) + 0.5
    shift = torch.randn(64, generator=g) * 0.1
    wst = torch.empty((64, 7, 8, 4), dtype=torch.bfloat16, device="cuda")
    ops.stem_weight_pack(oihw_to_ohwi(w).cuda(), scale.cuda(), wst)
    y = torch.empty((N * (Hp // 2) * (Wp // 2), 64), dtype=torch.bfloat16, device="cuda")
    ops.stem_conv7x7_fwd(N, Hp, Wp, xh, wst, shift.cuda(), y)
    weff = bf16_round(w * scale.view(-1, 1, 1, 1))
    ref = TF.relu(TF.conv2d(bf16_round(xo), weff, shift, stride=2, padding=3))
    got = pm_to_nchw(y, N, Hp // 2, Wp // 2)
    assert rel_l2(got, ref) < TOL
    p = torch.empty((N * (Hp // 4) * (Wp // 4), 64), dtype=torch.bfloat16, device="cuda")
    ops.maxpool3x3s2_fwd(y, N, Hp // 2, Wp // 2, 64, p)
    refp = TF.max_pool2d(got, 3, 2, 1)
    assert torch.equal(pm_to_nchw(p, N, Hp // 4, Wp // 4), refp)
    pf = torch.zeros_like(p)
    ops.stem_pool_fwd(N, Hp, Wp, xh, wst, shift.cuda(), pf)
    assert torch.equal(pf, p)


@pytest.mark.parametrize("N,Hp,Wp", [(1, 32, 32), (2, 96, 160), (3, 224, 416), (1, 800, 1344), (2, 34, 58), (5, 800, 1344)])
def test_stem_pool_fused_is_bit_identical(N, Hp, Wp):
    """bd_stem_pool_fwd (one launch, no half-resolution tensor) == bd_stem_conv7x7_fwd + bd_maxpool3x3s2_fwd, bit for bit: strip and
    chunk edges (pooled widths that are / are not multiples of 7, heights that are / are not multiples of 10, odd stem sizes); the 5-image
    case has more strips (4 800) than resident waves (2 048): a wave walks several strips through the same ring."""
    ops = _ops()
    g = torch.Generator().manual_seed(Hp * 7 + Wp)
    xh = torch.zeros((N, Hp + 6, Wp + 8, 4), dtype=torch.bfloat16)
    xh[:, 3:3 + Hp, 4:4 + Wp, :3] = torch.randn(N, Hp, Wp, 3, generator=g).to(torch.bfloat16)
    xh = xh.cuda()
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.05
    shift = (torch.randn(64, generator=g) * 0.1).cuda()
    wst = torch.empty((64, 7, 8, 4), dtype=torch.bfloat16, device="cuda")
    ops.stem_weight_pack(oihw_to_ohwi(w).cuda(), None, wst)
    Ho, Wo = Hp // 2, Wp // 2
    Hq, Wq = (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1
    y = torch.empty((N * Ho * Wo, 64), dtype=torch.bfloat16, device="cuda")
    ops.stem_conv7x7_fwd(N, Hp, Wp, xh, wst, shift, y)
    p = torch.empty((N * Hq * Wq, 64), dtype=torch.bfloat16, device="cuda")
    ops.maxpool3x3s2_fwd(y, N, Ho, Wo, 64, p)
    pf = torch.full_like(p, -1.0)
    ops.stem_pool_fwd(N, Hp, Wp, xh, wst, shift, pf)
    torch.cuda.synchronize()
    assert torch.equal(pf, p), int((pf != p).sum())


def test_upsample_add_fwd_bwd():
    """FPN top-down merge (fpn_backbone.py:143-148) and its gradient."""
    ops = _ops()
    N, C, H, W = 2, 16, 5, 7
    g = torch.Generator().manual_seed(5)
    top = bf16_round(torch.randn(N, C, H, W, generator=g))
    lat = bf16_round(torch.randn(N, C, 2 * H, 2 * W, generator=g))
    gt_, gl_ = ops.single(N, H, W), ops.single(N, 2 * H, 2 * W)
    latp = nchw_to_pm(lat)
    ops.upsample2x_add_fwd(nchw_to_pm(top), gt_, latp, gl_, C)
    ref = lat + TF.interpolate(top, scale_factor=2, mode="bilinear", align_corners=False)
    assert rel_l2(pm_to_nchw(latp, N, 2 * H, 2 * W), ref) < 5e-3
    dl = bf16_round(torch.randn(N, C, 2 * H, 2 * W, generator=g))
    tr = top.clone().requires_grad_(True)
    TF.interpolate(tr, scale_factor=2, mode="bilinear", align_corners=False).backward(dl)
    prev = bf16_round(torch.randn(N, C, H, W, generator=g))
    dt = nchw_to_pm(prev)
    ops.upsample2x_add_bwd(nchw_to_pm(dl), gl_, dt, gt_, C, accumulate=True)
    assert rel_l2(pm_to_nchw(dt, N, H, W), prev + tr.grad) < 5e-3
    ops.upsample2x_add_bwd(nchw_to_pm(dl), gl_, dt, gt_, C, accumulate=False)
    assert rel_l2(pm_to_nchw(dt, N, H, W), tr.grad) < 5e-3


def test_elementwise_pack_colsum_sgd():
    ops = _ops()
    g = torch.Generator().manual_seed(9)
    a = bf16_round(torch.randn(4096, generator=g)); b = bf16_round(torch.randn(4096, generator=g)); c = bf16_round(torch.randn(4096, generator=g))
    ad, bd_, cd = (t.to(torch.bfloat16).cuda() for t in (a, b, c))
    y = torch.empty_like(ad)
    assert torch.equal(ops.relu_bf16(ad, y).float().cpu(), a.clamp(min=0))
    assert torch.equal(ops.add_bf16(ad, bd_, y).float().cpu(), bf16_round(a + b))
    assert torch.equal(ops.relu_bwd_bf16(ad, bd_, y, add=cd).float().cpu(), bf16_round(a * (b > 0) + c))
    # colsum
    for rows, Cn in ((1000, 256), (777, 720), (513, 40)):
        m = bf16_round(torch.randn(rows, Cn, generator=g))
        out = torch.zeros(Cn, device="cuda")
        ws = torch.empty((ops.colsum_workspace_bytes(Cn) // 4,), dtype=torch.float32, device="cuda")
        ops.colsum_bf16(m.to(torch.bfloat16).cuda(), rows, Cn, out, ws)
        assert torch.allclose(out.cpu(), m.sum(0), rtol=1e-4, atol=1e-3)
    # weight pack
    w = torch.randn(24, 40, 3, 3, generator=g)
    sc = torch.rand(24, generator=g) + 0.5
    wf, wd = pack_weights(ops, w, sc)
    ref = bf16_round(oihw_to_ohwi(w) * sc.view(-1, 1, 1, 1)).reshape(24, 9, 40)
    assert torch.equal(wf.float().cpu(), ref)
    assert torch.equal(wd.float().cpu(), ref.permute(2, 1, 0).contiguous())
    # sgd (megengine.optimizer.SGD form, oracle/model.py sgd_step)
    n = 10007
    wv = torch.randn(n, generator=g); vv = torch.randn(n, generator=g); gv = torch.randn(n, generator=g)
    wd_, vd, gd = wv.cuda(), vv.cuda(), gv.cuda()
    ops.sgd_momentum_step(wd_, vd, gd, lr=0.01, momentum=0.9, wd=1e-4, grad_scale=0.5)
    gg = gv * 0.5 + 1e-4 * wv
    vref = 0.9 * vv + gg
    assert torch.allclose(vd.cpu(), vref, rtol=1e-6, atol=1e-7)
    assert torch.allclose(wd_.cpu(), wv - 0.01 * vref, rtol=1e-6, atol=1e-7)


def test_weight_pack_multi_matches_single():
    """bd_weight_pack_multi (one launch for all trainable convs) == bd_weight_pack per conv, bit for bit."""
    from basedet_amd import ops
    torch.manual_seed(0)
    shapes = [(64, 9, 64), (256, 1, 1024), (40, 9, 256), (720, 9, 256), (16, 1, 256), (1024, 1, 72)]
    ent, refs = [], []
    for co, rs, ci in shapes:
        w = torch.randn(co, rs, ci, device="cuda")
        scale = torch.rand(co, device="cuda") + 0.5 if co % 3 == 1 else None
        wf = torch.zeros((co, rs, ci), dtype=torch.bfloat16, device="cuda")
        wd = torch.zeros((ci, rs, co), dtype=torch.bfloat16, device="cuda") if co != 16 else None
        rf = torch.empty_like(wf)
        rd = torch.empty((ci, rs, co), dtype=torch.bfloat16, device="cuda") if wd is not None else None
        ops.weight_pack(w, scale, rf, rd, co, rs, ci)
        ent.append((w, scale, wf, wd, co, rs, ci)); refs.append((rf, rd))
    table = ops.build_pack_table(ent, torch.device("cuda"))
    ops.weight_pack_multi(table)
    for (w, scale, wf, wd, *_), (rf, rd) in zip(ent, refs):
        assert torch.equal(wf, rf)
        if wd is not None:
            assert torch.equal(wd, rd)



@pytest.mark.parametrize("shape", [(5, 96, 160, 128, 256), (2, 100, 168, 64, 720), (3, 50, 84, 192, 264), (2, 40, 56, 136, 40), (4, 120, 200, 64, 64),
                                   (2, 64, 96, 200, 56)])
def test_patch_instances_agree_bitwise(shape):
    """Full-size grids (more workgroups than CUs, ragged channel tiles, K tails): the staggered 256-channel instance must give the
    SAME bits as the 128-channel instance (same accumulation order), forward with residual + ReLU and dgrad with add + mask."""
    ops = _ops()
    N, H, W, Cin, Cout = shape
    gin = ops.single(N, H, W)
    d = ops.conv_desc(gin, gin, Cin, Cout, 3, 3, 1, 1)
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(gin.pixels, Cin, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(Cout, 9, Cin, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    wt = (torch.randn(Cin, 9, Cout, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    b = torch.randn(Cout, device="cuda", generator=g)
    res = torch.randn(gin.pixels, Cout, device="cuda", generator=g).to(torch.bfloat16)
    gy = torch.randn(gin.pixels, Cout, device="cuda", generator=g).to(torch.bfloat16)
    addx = torch.randn(gin.pixels, Cin, device="cuda", generator=g).to(torch.bfloat16)
    gate = torch.relu(torch.randn(gin.pixels, Cin, device="cuda", generator=g)).to(torch.bfloat16)
    gate.view(torch.int16)[::7, ::3] = -32768            # -0.0
    gate[::5, 1::4] = -1.5
    outs = []
    for knob in (3 | 64 | 512, 3, 3 | 64 | 256):
        ops.set_route(patch3x3=knob)
        y = torch.full((gin.pixels, Cout), 3.0, device="cuda", dtype=torch.bfloat16)
        dx = torch.full((gin.pixels, Cin), 3.0, device="cuda", dtype=torch.bfloat16)
        ops.conv2d_fwd(d, x, w, b, y, add=res, flags=ops.EPI_RELU | ops.EPI_ADD_BEFORE)
        if Cin > 128:                                  # dgrad produces Cin channels; (…, 136, 40): a 40-channel reduction (bbox_pred), K < 64
            ops.conv2d_dgrad(d, gy, wt, dx, add=addx, mask=addx, flags=ops.EPI_ADD_BEFORE | ops.EPI_MASK)
        # the epilogue forms without a residual operand (their own code path in the 256-channel instance): forward bias / bias + ReLU,
        # dgrad plain / gated by a stored activation (zeros, negative zeros and negatives all close the gate)
        y2, y3 = torch.full_like(y, 3.0), torch.full_like(y, 3.0)
        dx2, dx3 = torch.full_like(dx, 3.0), torch.full_like(dx, 3.0)
        ops.conv2d_fwd(d, x, w, b, y2, flags=ops.EPI_RELU)
        ops.conv2d_fwd(d, x, w, b, y3)
        if Cin > 128:
            ops.conv2d_dgrad(d, gy, wt, dx2, mask=gate, flags=ops.EPI_MASK)
            ops.conv2d_dgrad(d, gy, wt, dx3)
        torch.cuda.synchronize()
        outs.append((y.clone(), dx.clone(), y2, y3, dx2, dx3))
    ops.set_route(patch3x3=3)
    for k in (1, 2):          # both staggered instances against the plain 128-channel kernel
        for a, b_ in zip(outs[0], outs[k]):
            assert torch.equal(a, b_)
    if Cin > 128:             # and the gate itself against its definition
        assert torch.equal(outs[1][4], torch.where(gate.float() > 0, outs[1][5], torch.zeros_like(outs[1][5])))


@pytest.mark.parametrize("case", [(2, 64, 72, 19, 27, 3, 1, 1), (2, 128, 64, 22, 30, 3, 2, 1), (3, 64, 136, 9, 14, 1, 1, 0),
                                  (2, 128, 136, 19, 27, 3, 1, 1), (1, 72, 264, 21, 9, 3, 1, 1)])      # the last two: the ring-staged kernel
@pytest.mark.parametrize("accumulate", [False, True])
def test_wgrad_bias_entry_point(case, accumulate):
    """bd_conv2d_wgrad_bias: weight gradient + bias gradient (column sums of g) in one call -- fused into the nine-tap kernel for the
    3x3 cases, followed by the column-sum pass for the 1x1 case; accumulate adds to both."""
    ops = _ops()
    N, Cin, Cout, H, W, R, stride, pad = case
    gen = torch.Generator().manual_seed(77 + Cin + Cout)
    x = bf16_round(torch.randn(N, Cin, H, W, generator=gen))
    gin = ops.single(N, H, W)
    gout = gin.conv_out(R, stride, pad)
    Ho, Wo = gout.H[0], gout.W[0]
    gy = bf16_round(torch.randn(N, Cout, Ho, Wo, generator=gen))
    d = ops.conv_desc(gin, gout, Cin, Cout, R, R, stride, pad)
    wr = torch.zeros(Cout, Cin, R, R, requires_grad=True)
    br = torch.zeros(Cout, requires_grad=True)
    TF.conv2d(x, wr, br, stride=stride, padding=pad).backward(gy)
    ws = torch.empty((ops.conv2d_wgrad_bias_workspace_bytes(d) // 4 + 4,), dtype=torch.float32, device="cuda")
    dw = torch.full((Cout, R, R, Cin), 0.5, dtype=torch.float32, device="cuda")
    db = torch.full((Cout,), 0.25, dtype=torch.float32, device="cuda")
    ops.conv2d_wgrad_bias(d, nchw_to_pm(x), nchw_to_pm(gy), dw, db, ws, accumulate=accumulate)
    base_w, base_b = (0.5, 0.25) if accumulate else (0.0, 0.0)
    assert rel_l2(dw.cpu() - base_w, oihw_to_ohwi(wr.grad)) < 2e-3
    assert rel_l2(db.cpu() - base_b, br.grad) < 1e-4


@pytest.mark.parametrize("chans", [(64, 72), (64, 200)])          # (64, 200): the ring-staged kernel (ones-row MFMA column sums)
def test_wgrad_bias_multilevel(chans):
    """Five pyramid levels in one descriptor: the fused column sums cover every level's pixels exactly once (ragged patches)."""
    ops = _ops()
    N, (C, Cout) = 2, chans
    Hs, Ws = [12, 6, 3, 2, 1], [20, 10, 5, 3, 2]
    gen = torch.Generator().manual_seed(11)
    geo = ops.Geom(N, Hs, Ws)
    d = ops.conv_desc(geo, geo, C, Cout, 3, 3, 1, 1)
    xp = bf16_round(torch.randn(geo.pixels, C, generator=gen)).to(torch.bfloat16).cuda()
    gy = bf16_round(torch.randn(geo.pixels, Cout, generator=gen))
    ws = torch.empty((ops.conv2d_wgrad_bias_workspace_bytes(d) // 4 + 4,), dtype=torch.float32, device="cuda")
    dw = torch.empty((Cout, 3, 3, C), dtype=torch.float32, device="cuda")
    db = torch.empty((Cout,), dtype=torch.float32, device="cuda")
    ops.conv2d_wgrad_bias(d, xp, gy.to(torch.bfloat16).cuda(), dw, db, ws)
    assert rel_l2(db.cpu(), gy.sum(0)) < 1e-4


@pytest.mark.parametrize("depth", [1, 0, 2, 3, 4, 6])
def test_dense_1x1_kernel_and_mask_bits(depth):
    """conv1x1.hip (every 1x1 / stride 1 launch over one dense level) in each variant (bd_conv_desc.route[0]: 0 = the generic kernel,
    1 = the default choice, 2 = 256^2 wherever legal, 3 = 128^2 only, 4 = the eight-wave 256-channel x 128-pixel tile wherever legal): forward with
    residual + ReLU and the data gradient with accumulate + mask against torch-CPU fp32; the bit-packed ReLU mask written by the
    forward launch equals (y > 0) bit for bit, and a data gradient gated by it equals the one gated by the bf16 activation."""
    ops = _ops()
    # 6: the 128^2 tile's LDS-DMA ring variant for every K that allows it (default: 512 <= K <= 1024 only)
    ops.set_route(dense1x1=depth)
    try:
        for (N, Cin, Cout, H, W) in ((2, 256, 64, 23, 37), (1, 64, 256, 50, 41), (2, 200, 192, 9, 13), (1, 1024, 256, 20, 21), (1, 32, 544, 7, 9),
                                     (1, 320, 448, 13, 19), (3, 512, 512, 17, 31)):
            g = torch.Generator().manual_seed(77 + Cin + Cout)
            x = bf16_round(torch.randn(N, Cin, H, W, generator=g))
            w = bf16_round(torch.randn(Cout, Cin, 1, 1, generator=g) / np.sqrt(Cin))
            bias = torch.randn(Cout, generator=g)
            res = bf16_round(torch.randn(N, Cout, H, W, generator=g))
            geo = ops.single(N, H, W)
            d = ops.conv_desc(geo, geo, Cin, Cout, 1, 1, 1, 0)
            wf, wd = pack_weights(ops, w)
            M = N * H * W
            y = torch.empty((M, Cout), dtype=torch.bfloat16, device="cuda")
            bits_ok = depth != 0 and ops.dense_1x1_bits_ok(d)
            ybits = torch.full((Cout // 32, M), -1, dtype=torch.int32, device="cuda") if bits_ok else None
            ops.conv2d_fwd(d, nchw_to_pm(x), wf, bias.cuda(), y, add=nchw_to_pm(res), flags=ops.EPI_RELU | ops.EPI_ADD_BEFORE, bits=ybits)
            ref = torch.relu(TF.conv2d(x, w, bias) + res)
            got = pm_to_nchw(y, N, H, W)
            assert rel_l2(got, ref) < 1e-2, (depth, Cin, Cout)
            if bits_ok:
                yb = (y.float() > 0).cpu().numpy().reshape(M, Cout // 32, 32)
                want = (yb.astype(np.uint64) << np.arange(32, dtype=np.uint64)).sum(-1).astype(np.uint32).T      # [Cout/32][M]
                assert np.array_equal(ybits.cpu().numpy().view(np.uint32), want)
            # data gradient of the transposed problem (K = Cout -> Cin), accumulate + mask
            gy = bf16_round(torch.randn(N, Cout, H, W, generator=g))
            acc0 = bf16_round(torch.randn(N, Cin, H, W, generator=g))
            act = torch.relu(bf16_round(torch.randn(N, Cin, H, W, generator=g)))           # the forward activation that gates dx
            xr = x.clone().requires_grad_(True)
            TF.conv2d(xr, w).backward(gy)
            refd = (xr.grad + acc0) * (act > 0)
            dx = nchw_to_pm(acc0).clone()
            ops.conv2d_dgrad(d, nchw_to_pm(gy), wd, dx, add=dx, mask=nchw_to_pm(act), flags=ops.EPI_ADD_BEFORE | ops.EPI_MASK)
            assert rel_l2(pm_to_nchw(dx, N, H, W), refd) < 1e-2, (depth, Cin, Cout)
            if bits_ok:
                ab = (nchw_to_pm(act).float() > 0).cpu().numpy().reshape(M, Cin // 32, 32)
                abits = (ab.astype(np.uint64) << np.arange(32, dtype=np.uint64)).sum(-1).astype(np.uint32).T.copy()
                dx2 = nchw_to_pm(acc0).clone()
                ops.conv2d_dgrad(d, nchw_to_pm(gy), wd, dx2, add=dx2, maskbits=torch.from_numpy(abits.view(np.int32)).cuda(),
                                 flags=ops.EPI_ADD_BEFORE)
                assert torch.equal(dx, dx2)
                # one-byte twins (e4m3 of the forward output, e5m2 of the data gradient): the same bytes from either tile
                if depth in (2, 3, 4, 6):
                    tw = {}
                    for dd in (2, 3, 4):
                        ops.set_route(dense1x1=dd)
                        y8 = torch.zeros((M, Cout), dtype=torch.uint8, device="cuda")
                        y2 = torch.empty_like(y)
                        ops.conv2d_fwd(d, nchw_to_pm(x), wf, bias.cuda(), y2, add=nchw_to_pm(res), flags=ops.EPI_RELU | ops.EPI_ADD_BEFORE,
                                       y8=y8, q_scale=0.5)
                        dx8 = torch.zeros((M, Cin), dtype=torch.uint8, device="cuda")
                        dx3 = nchw_to_pm(acc0).clone()
                        ops.conv2d_dgrad(d, nchw_to_pm(gy), wd, dx3, add=dx3, mask=nchw_to_pm(act), flags=ops.EPI_ADD_BEFORE | ops.EPI_MASK,
                                         dx8=dx8, q_scale=64.0)
                        tw[dd] = (y2.clone(), y8.clone(), dx3.clone(), dx8.clone())
                    ops.set_route(dense1x1=depth)
                    for a, b, c in zip(tw[2], tw[3], tw[4]):
                        assert torch.equal(a, b) and torch.equal(a, c)
                    assert torch.equal(tw[2][0], y) and torch.equal(tw[2][2], dx)
    finally:
        ops.set_route(dense1x1=1)


@pytest.mark.parametrize("case", [(4, 60, 70, 128, 512), (4, 60, 70, 512, 128), (2, 50, 84, 256, 1024), (2, 50, 84, 1024, 256), (16, 25, 42, 64, 256),
                                  (1, 33, 47, 192, 200), (3, 40, 50, 320, 320), (1, 11, 9, 2048, 512), (2, 37, 41, 128, 64)])
def test_conv1x1_ring_kernel_gives_the_dense_kernels_bits(case):
    """conv1x1_ring.hip (persistent workgroups, eight-stage LDS-DMA ring that runs across tile boundaries, epilogue operands requested a
    tile ahead, every wait an exact vmcnt; bd_conv_desc.route[0] mode 5: every launch it can take) against conv1x1_dense_kernel
    (bd_conv_desc.route[0] mode 3): the same bits for every epilogue the step uses -- K of 2 .. 32 ring steps (fewer / more than the ring is deep),
    several tiles per workgroup, ragged pixel and channel tiles."""
    ops = _ops()
    N, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(5 + Cin + Cout + H)
    M = N * H * W
    x = bf16_round(torch.randn(M, Cin, generator=g)).to(torch.bfloat16).cuda()
    w = bf16_round(torch.randn(Cout, Cin, 1, 1, generator=g) / np.sqrt(Cin))
    wf, wd = pack_weights(ops, w)
    bias = torch.randn(Cout, generator=g).cuda()
    res = bf16_round(torch.randn(M, Cout, generator=g)).to(torch.bfloat16).cuda()
    gy = bf16_round(torch.randn(M, Cout, generator=g)).to(torch.bfloat16).cuda()
    acc0 = bf16_round(torch.randn(M, Cin, generator=g)).to(torch.bfloat16).cuda()
    geo = ops.single(N, H, W)
    d = ops.conv_desc(geo, geo, Cin, Cout, 1, 1, 1, 0)
    bits_f = Cout % 32 == 0
    bits_d = Cin % 32 == 0
    gate = torch.randint(-2 ** 31, 2 ** 31 - 1, (max(Cin // 32, 1), M), dtype=torch.int32, generator=g).cuda()
    out = {}
    try:
        for mode in (3, 5):
            ops.set_route(dense1x1=mode)
            r = []
            # forward: residual + ReLU (+ gate bits out); bias only; residual added after the (absent) gate
            y = torch.full((M, Cout), 7.0, dtype=torch.bfloat16, device="cuda")
            yb = torch.full((Cout // 32, M), -1, dtype=torch.int32, device="cuda") if bits_f else None
            ops.conv2d_fwd(d, x, wf, bias, y, add=res, flags=ops.EPI_RELU | ops.EPI_ADD_BEFORE, bits=yb)
            r += [y, yb]
            y2 = torch.full((M, Cout), 7.0, dtype=torch.bfloat16, device="cuda")
            ops.conv2d_fwd(d, x, wf, bias, y2)
            r.append(y2)
            y3 = torch.full((M, Cout), 7.0, dtype=torch.bfloat16, device="cuda")
            ops.conv2d_fwd(d, x, wf, None, y3, add=res, flags=ops.EPI_ADD_AFTER)
            r.append(y3)
            # data gradient: plain; gated by bits; accumulated in place and gated by bits
            dx = torch.full((M, Cin), 7.0, dtype=torch.bfloat16, device="cuda")
            ops.conv2d_dgrad(d, gy, wd, dx)
            r.append(dx)
            if bits_d:
                dx2 = torch.full((M, Cin), 7.0, dtype=torch.bfloat16, device="cuda")
                ops.conv2d_dgrad(d, gy, wd, dx2, maskbits=gate, flags=0)
                dx3 = acc0.clone()
                ops.conv2d_dgrad(d, gy, wd, dx3, add=dx3, maskbits=gate, flags=ops.EPI_ADD_BEFORE)
                r += [dx2, dx3]
            torch.cuda.synchronize()
            out[mode] = r
    finally:
        ops.set_route(dense1x1=1)
    for k, (a, b) in enumerate(zip(out[3], out[5])):
        assert (a is None and b is None) or torch.equal(a, b), (case, k)
    # and against fp32 (the ring path by itself)
    ref = torch.relu(x.float().cpu() @ w.reshape(Cout, Cin).t() + bias.cpu() + res.float().cpu())
    assert rel_l2(out[5][0].float().cpu(), ref) < 1e-2


@pytest.mark.parametrize("R", [1, 3])
def test_strided_dgrad_sparse_accumulate(R):
    """BD_EPI_SPARSE: a stride-2 data gradient accumulating in place leaves the input pixels no tap reaches untouched (1x1: three of
    four), and equals the dense accumulate everywhere (the gate is idempotent: dx already holds gated values)."""
    ops = _ops()
    N, Cin, Cout, H, W = 2, 64, 128, 14, 22
    pad = R // 2
    g = torch.Generator().manual_seed(9 + R)
    w = bf16_round(torch.randn(Cout, Cin, R, R, generator=g) / np.sqrt(Cin * R * R))
    gin = ops.single(N, H, W)
    gout = gin.conv_out(R, 2, pad)
    d = ops.conv_desc(gin, gout, Cin, Cout, R, R, 2, pad)
    Ho, Wo = gout.H[0], gout.W[0]
    gy = bf16_round(torch.randn(N, Cout, Ho, Wo, generator=g))
    act = torch.relu(bf16_round(torch.randn(N, Cin, H, W, generator=g)))
    dx0 = bf16_round(torch.randn(N, Cin, H, W, generator=g)) * (act > 0)                 # what an earlier launch left: already gated
    wf, wd = pack_weights(ops, w)
    dense = nchw_to_pm(dx0).clone()
    ops.conv2d_dgrad(d, nchw_to_pm(gy), wd, dense, add=dense, mask=nchw_to_pm(act), flags=ops.EPI_ADD_BEFORE | ops.EPI_MASK)
    sparse = nchw_to_pm(dx0).clone()
    ops.conv2d_dgrad(d, nchw_to_pm(gy), wd, sparse, add=sparse, mask=nchw_to_pm(act), flags=ops.EPI_ADD_BEFORE | ops.EPI_MASK | ops.EPI_SPARSE)
    assert torch.equal(dense, sparse)
    xr = torch.zeros(N, Cin, H, W, requires_grad=True)
    TF.conv2d(xr, w, stride=2, padding=pad).backward(gy)
    ref = (xr.grad + dx0) * (act > 0)
    assert rel_l2(pm_to_nchw(sparse, N, H, W), ref) < 1e-2
    if R == 1:                        # untouched means untouched: poison the unreached pixels, they must survive
        poison = nchw_to_pm(dx0).clone().view(N, H, W, Cin)
        poison[:, 1::2] = 7.0
        poison[:, :, 1::2] = 7.0
        poison = poison.view(-1, Cin).contiguous()
        keep = poison.clone()
        ops.conv2d_dgrad(d, nchw_to_pm(gy), wd, poison, add=poison, mask=nchw_to_pm(act), flags=ops.EPI_ADD_BEFORE | ops.EPI_MASK | ops.EPI_SPARSE)
        pv, kv = poison.view(N, H, W, Cin), keep.view(N, H, W, Cin)
        assert torch.equal(pv[:, 1::2], kv[:, 1::2]) and torch.equal(pv[:, :, 1::2], kv[:, :, 1::2])
        assert torch.equal(pv[:, ::2, ::2], sparse.view(N, H, W, Cin)[:, ::2, ::2])


@pytest.mark.parametrize("shape,knob", [((16, 50, 84, 256, 256), 3), ((15, 100, 168, 256, 256), 3), ((16, 50, 84, 192, 200), 3),
                                        ((16, 200, 336, 64, 256), 3), ((3, 100, 168, 64, 720), 3)])
def test_pp_tail_split_same_bits(shape, knob):
    """conv3x3_pp.hip runs the last (grid mod CUs) pixel tiles of a launch on the 64-channel tile, in the last workgroups of the same grid
    (one workgroup per CU: a small remainder otherwise costs a whole round; res4's 3x3 at 16 x 50x84 = 312 tiles = 256 + 56; 15 x 100x168 =
    1032 = 4 x 256 + 8 with a ragged last tile; 200 channels: a ragged last channel tile), and the main tiles by PERSISTENT workgroups that
    walk up to 16 tiles each (16 x 200x336: 4 200 tiles = two workgroups per CU; 720 channels: three channel tiles, 255 workgroups).
    Same accumulation order: the launch must give the bits of the one-workgroup-per-tile, unsplit form (knob bits 13 + 14), forward and
    data gradient, plain and fused epilogues."""
    ops = _ops()
    N, H, W, Cin, Cout = shape
    gin = ops.single(N, H, W)
    d = ops.conv_desc(gin, gin, Cin, Cout, 3, 3, 1, 1)
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(gin.pixels, Cin, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(Cout, 9, Cin, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    wt = (torch.randn(Cin, 9, Cout, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    b = torch.randn(Cout, device="cuda", generator=g)
    res = torch.randn(gin.pixels, Cout, device="cuda", generator=g).to(torch.bfloat16)
    gy = torch.randn(gin.pixels, Cout, device="cuda", generator=g).to(torch.bfloat16)
    gate = torch.relu(torch.randn(gin.pixels, Cin, device="cuda", generator=g)).to(torch.bfloat16)
    outs = []
    for kb in (knob | 8192 | 16384, knob):
        ops.set_route(patch3x3=kb)
        y = torch.full((gin.pixels, Cout), 3.0, device="cuda", dtype=torch.bfloat16)
        y2 = torch.full_like(y, 3.0)
        dx = torch.full((gin.pixels, Cin), 3.0, device="cuda", dtype=torch.bfloat16)
        dx2 = torch.full_like(dx, 3.0)
        ops.conv2d_fwd(d, x, w, b, y, add=res, flags=ops.EPI_RELU | ops.EPI_ADD_BEFORE)
        ops.conv2d_fwd(d, x, w, b, y2, flags=ops.EPI_RELU)
        ops.conv2d_dgrad(d, gy, wt, dx, mask=gate, flags=ops.EPI_MASK)
        ops.conv2d_dgrad(d, gy, wt, dx2)
        torch.cuda.synchronize()
        outs.append((y, y2, dx, dx2))
    ops.set_route(patch3x3=3)
    for a, b_ in zip(*outs):
        assert torch.equal(a, b_)
    # and against fp32 on a strip that lies in the tail tiles (the last image)
    import torch.nn.functional as F
    xi = x.view(N, H, W, Cin)[-1:].permute(0, 3, 1, 2).float()
    wf = w.view(Cout, 3, 3, Cin).permute(0, 3, 1, 2).float()
    ref = torch.relu(F.conv2d(xi, wf, b, padding=1)).permute(0, 2, 3, 1).reshape(-1, Cout)
    got = outs[1][1].view(N, H * W, Cout)[-1].float()
    assert float((got - ref).norm() / ref.norm()) < 1e-2


def test_pp_persistent_launches_on_two_streams_same_bits():
    """Two streams launch the persistent 3x3 kernel (one workgroup per CU each, 161 KB of LDS: they cannot share a CU) against each other, with
    different shapes, twenty times: every result equals the single-stream one (no state outside a workgroup; the tail workgroups of one
    launch and the persistent ones of the other interleave on the CUs)."""
    ops = _ops()
    g = torch.Generator(device="cuda").manual_seed(23)
    cases = []
    for (N, H, W, Cin, Cout) in [(16, 50, 84, 256, 256), (6, 100, 168, 256, 256)]:
        gin = ops.single(N, H, W)
        d = ops.conv_desc(gin, gin, Cin, Cout, 3, 3, 1, 1)
        x = torch.randn(gin.pixels, Cin, device="cuda", generator=g).to(torch.bfloat16)
        w = (torch.randn(Cout, 9, Cin, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
        b = torch.randn(Cout, device="cuda", generator=g)
        ref = torch.empty((gin.pixels, Cout), device="cuda", dtype=torch.bfloat16)
        ops.conv2d_fwd(d, x, w, b, ref, flags=ops.EPI_RELU)
        cases.append((d, x, w, b, ref, gin.pixels, Cout))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [[], []]
    for it in range(20):
        for k, st in enumerate(streams):
            d, x, w, b, ref, pix, Cout = cases[k]
            with torch.cuda.stream(st):
                y = torch.full((pix, Cout), 3.0, device="cuda", dtype=torch.bfloat16)
                ops.conv2d_fwd(d, x, w, b, y, flags=ops.EPI_RELU)
                outs[k].append(y)
    for st in streams:
        st.synchronize()
    for k in range(2):
        for y in outs[k]:
            assert torch.equal(y, cases[k][4])


@pytest.mark.parametrize("case", [(2, 256, 256, 50, 84), (1, 128, 720, 37, 41), (3, 200, 136, 11, 19), (2, 64, 96, 8, 8), (1, 64, 128, 3, 5)])
def test_wgrad_ring_kernel_agrees_with_the_register_staged_kernel(case):
    """conv_wgrad3x3_ring.hip (LDS-DMA ring, 64 ci x 128 co tile, persistent workgroups) against conv_wgrad3x3.hip (bit 2 of
    bd_conv_desc.route[2] routes its shapes back there): the same products summed in another order -> fp32 rounding only; the
    bias column sums likewise; and two launches of the ring kernel are bit-identical (fixed-order reduce)."""
    ops = _ops()
    N, Cin, Cout, H, W = case
    gen = torch.Generator().manual_seed(5 + Cin + Cout)
    geo = ops.single(N, H, W)
    d = ops.conv_desc(geo, geo, Cin, Cout, 3, 3, 1, 1)
    x = bf16_round(torch.randn(geo.pixels, Cin, generator=gen)).to(torch.bfloat16).cuda()
    gy = bf16_round(torch.randn(geo.pixels, Cout, generator=gen)).to(torch.bfloat16).cuda()
    ws = torch.empty((ops.conv2d_wgrad_bias_workspace_bytes(d) // 4 + 4,), dtype=torch.float32, device="cuda")
    outs = []
    for knob in (1, 5, 1):
        ops.set_route(wgrad=knob)
        dw = torch.full((Cout, 3, 3, Cin), 3.0, dtype=torch.float32, device="cuda")
        db = torch.full((Cout,), 3.0, dtype=torch.float32, device="cuda")
        ops.conv2d_wgrad_bias(d, x, gy, dw, db, ws)
        torch.cuda.synchronize()
        outs.append((dw.cpu(), db.cpu()))
    ops.set_route(wgrad=1)
    assert rel_l2(outs[0][0], outs[1][0]) < 2e-6 and rel_l2(outs[0][1], outs[1][1]) < 2e-6
    assert torch.equal(outs[0][0], outs[2][0]) and torch.equal(outs[0][1], outs[2][1])
    assert rel_l2(outs[0][1], gy.float().sum(0).cpu()) < 1e-4


@pytest.mark.parametrize("case", [(2, 1024, 256, 50, 84), (2, 128, 512, 40, 56), (1, 512, 2048, 13, 21), (2, 256, 64, 30, 40), (1, 72, 200, 9, 11),
                                  (1, 64, 64, 2, 3)])
def test_wgrad1x1_ring_kernel_agrees_with_the_register_staged_kernel(case):
    """conv_wgrad1x1_ring.hip (LDS-DMA ring, persistent workgroups) against conv_wgrad1x1.hip (bit 2 of bd_conv_desc.route[2] routes
    its shapes back there) and against the fp32 definition dW = G^T X: same products in another order -> fp32 rounding only; two launches
    of the ring kernel are bit-identical; row scale and accumulate go through the ring reduce."""
    ops = _ops()
    N, Cin, Cout, H, W = case
    gen = torch.Generator().manual_seed(9 + Cin + Cout)
    geo = ops.single(N, H, W)
    d = ops.conv_desc(geo, geo, Cin, Cout, 1, 1, 1, 0)
    x = bf16_round(torch.randn(geo.pixels, Cin, generator=gen))
    gy = bf16_round(torch.randn(geo.pixels, Cout, generator=gen))
    ref = (gy.double().t() @ x.double()).float().view(Cout, 1, 1, Cin)
    xd, gd = x.to(torch.bfloat16).cuda(), gy.to(torch.bfloat16).cuda()
    ws = torch.empty((ops.conv2d_wgrad_workspace_bytes(d) // 4 + 4,), dtype=torch.float32, device="cuda")
    scale = (torch.rand(Cout, generator=gen) + 0.5).cuda()
    outs = []
    for knob in (1, 5, 1):
        ops.set_route(wgrad=knob)
        dw = torch.full((Cout, 1, 1, Cin), 3.0, dtype=torch.float32, device="cuda")
        ops.conv2d_wgrad(d, xd, gd, dw, ws)
        first = dw.cpu()
        ops.conv2d_wgrad(d, xd, gd, dw, ws, row_scale=scale, accumulate=True)
        torch.cuda.synchronize()
        outs.append((first, dw.cpu()))
    ops.set_route(wgrad=1)
    assert rel_l2(outs[0][0], ref) < 1e-5
    assert rel_l2(outs[0][0], outs[1][0]) < 2e-6
    assert torch.equal(outs[0][0], outs[2][0]) and torch.equal(outs[0][1], outs[2][1])
    assert rel_l2(outs[0][1], ref * (1 + scale.cpu().view(-1, 1, 1, 1))) < 1e-5


def test_wgrad_queue_reduces_several_layers_in_one_launch_bit_identically():
    """bd_wgrad_queue_*: the partial-sum kernels of five layers (ring 3x3 with bias, ring 1x1, register-staged stride-2 3x3, generic 7x7-like
    filter, narrow 3x3 with bias) run back to back with their own workspaces, ONE flush reduces them; results are bit-identical to the
    per-layer entry points (same fixed-order sums), accumulate and row scale included, and nothing is pending afterwards."""
    ops = _ops()
    gen = torch.Generator().manual_seed(21)
    layers = [  # N, Cin, Cout, H, W, R, stride, pad, bias
        (2, 128, 136, 19, 27, 3, 1, 1, True), (2, 256, 128, 30, 40, 1, 1, 0, False), (2, 64, 192, 27, 31, 3, 2, 1, False),
        (1, 64, 64, 12, 14, 5, 1, 2, False), (2, 64, 72, 19, 27, 3, 1, 1, True)]
    q = ops.WgradQueue()
    items = []
    for (N, Cin, Cout, H, W, R, stride, pad, bias) in layers:
        gin = ops.single(N, H, W)
        gout = gin.conv_out(R, stride, pad)
        d = ops.conv_desc(gin, gout, Cin, Cout, R, R, stride, pad)
        x = bf16_round(torch.randn(gin.pixels, Cin, generator=gen)).to(torch.bfloat16).cuda()
        gy = bf16_round(torch.randn(gout.pixels, Cout, generator=gen)).to(torch.bfloat16).cuda()
        scale = (torch.rand(Cout, generator=gen) + 0.5).cuda()
        nbytes = ops.conv2d_wgrad_bias_workspace_bytes(d) if bias else ops.conv2d_wgrad_workspace_bytes(d)
        ws = torch.empty((nbytes // 4 + 4,), dtype=torch.float32, device="cuda")
        ws_ref = torch.empty_like(ws)
        dw_ref = torch.full((Cout, R, R, Cin), 2.0, dtype=torch.float32, device="cuda")
        db_ref = torch.full((Cout,), 2.0, dtype=torch.float32, device="cuda") if bias else None
        if bias:
            ops.conv2d_wgrad_bias(d, x, gy, dw_ref, db_ref, ws_ref, row_scale=scale, accumulate=True)
        else:
            ops.conv2d_wgrad(d, x, gy, dw_ref, ws_ref, row_scale=scale, accumulate=True)
        dw = torch.full((Cout, R, R, Cin), 2.0, dtype=torch.float32, device="cuda")
        db = torch.full((Cout,), 2.0, dtype=torch.float32, device="cuda") if bias else None
        q.wgrad(d, x, gy, dw, db, ws, row_scale=scale, accumulate=True)
        items.append((dw, db, dw_ref, db_ref, ws, x, gy, scale))
    assert q.pending() >= len(layers)
    before = [it[0].clone() for it in items]
    torch.cuda.synchronize()
    assert all(bool((b == 2.0).all()) for b in before)           # nothing is reduced before the flush
    q.flush()
    torch.cuda.synchronize()
    assert q.pending() == 0
    for i, (dw, db, dw_ref, db_ref, *_rest) in enumerate(items):
        assert torch.equal(dw, dw_ref), i
        if db is not None:
            assert torch.equal(db, db_ref), i
    q.flush()                                                    # an empty flush is a no-op
    q.close()


def test_conv_last_kernel_names_the_dispatched_kernel_for_the_bench_descriptors():
    """bd_conv_last_kernel() is written by the launch sites themselves; bench.py attributes a launch's time and roofline row by it.  This
    pins the names for the descriptors of the RetinaNet-R50 step at bench size (16 x 800 x 1344): a dispatch change shows up HERE, not
    as a silently mislabelled roofline row (round 4 mirrored the dispatch in Python)."""
    ops = _ops()
    N = 16
    last = lambda: ops.L().bd_conv_last_kernel().decode()
    bf = dict(dtype=torch.bfloat16, device="cuda")

    def run(gin, gout, cin, cout, k, stride, pad, kinds):
        d = ops.conv_desc(gin, gout, cin, cout, k, k, stride, pad)
        x = torch.zeros((gin.pixels, cin), **bf)
        y = torch.zeros((gout.pixels, cout), **bf)
        wf = torch.zeros((cout, k * k, cin), **bf)
        wd = torch.zeros((cin, k * k, cout), **bf)
        out = {}
        if "fwd" in kinds:
            ops.conv2d_fwd(d, x, wf, None, y)
            out["fwd"] = last()
        if "dgrad" in kinds:
            ops.conv2d_dgrad(d, y, wd, x)
            out["dgrad"] = last()
        if "wgrad" in kinds:
            ws = torch.empty((ops.conv2d_wgrad_workspace_bytes(d) // 4 + 4,), dtype=torch.float32, device="cuda")
            dw = torch.empty((cout, k, k, cin), dtype=torch.float32, device="cuda")
            ops.conv2d_wgrad(d, x, y, dw, ws)
            out["wgrad"] = last()
        torch.cuda.synchronize()
        return out

    all3 = ("fwd", "dgrad", "wgrad")
    pyr = ops.Geom(N, [100, 50, 25, 13, 7], [168, 84, 42, 21, 11])
    assert run(pyr, pyr, 256, 256, 3, 1, 1, all3) == {"fwd": "conv3x3_pp_kernel", "dgrad": "conv3x3_pp_kernel", "wgrad": "conv_wgrad3x3_ring_kernel"}
    g3, g4, g5 = ops.single(N, 100, 168), ops.single(N, 50, 84), ops.single(N, 25, 42)
    g2 = ops.single(N, 200, 336)
    # res3 conv1 (512 -> 128 at 100x168): the ring kernel; res5 conv1 (2048 -> 512 at 25x42): the K-sliced dense kernel family
    assert run(g3, g3, 512, 128, 1, 1, 0, all3) == {"fwd": "conv1x1_ring_kernel", "dgrad": "conv1x1_ring_kernel", "wgrad": "conv_wgrad1x1_ring_kernel"}
    r5 = run(g5, g5, 2048, 512, 1, 1, 0, all3)
    assert r5["fwd"] in ("conv1x1_dense_kernel", "conv1x1_gemm_kernel") and r5["wgrad"] == "conv_wgrad1x1_ring_kernel", r5
    # the stride-2 3x3 of res3.0 and the stride-2 shortcut of res4.0
    s2 = run(g2, g3, 128, 128, 3, 2, 1, all3)
    assert s2 == {"fwd": "conv_igemm_kernel<32>", "dgrad": "conv_igemm_kernel<32>", "wgrad": "conv_wgrad3x3_kernel"}, s2
    sc = run(g3, g4, 512, 1024, 1, 2, 0, ("fwd", "wgrad"))
    assert sc == {"fwd": "conv1x1_dense_kernel", "wgrad": "conv_wgrad1x1_kernel"}, sc
    # narrow 3x3 (res2-sized conv2 of a trainable layer1 would be 64 -> 64): the 64-channel staggered tile
    assert run(g3, g3, 64, 64, 3, 1, 1, ("fwd",)) == {"fwd": "conv3x3_pp128_kernel"}


@pytest.mark.parametrize("M", [16 * 700 + 5, 16 * 256 * 7 + 16, 37])
def test_thin_1x1_backward_in_one_pass(M):
    """bd_conv1x1_thin_bwd (the RPN prediction layer's backward: rpn.py:60-68 under autograd, 256 -> 3 + 12 channels padded to 16):
    dx = (x > 0) * g W, dW = g^T x, db = sum g against float64 on the same bf16-rounded operands -- dx within one bf16 rounding of the
    exact value (rel-L2 <= 3e-3), dW / db at fp32 summation accuracy (<= 1e-5) -- for pixel counts that are not a multiple of the
    16-pixel group, fewer groups than workgroups, and several groups per workgroup; two launches give identical bits."""
    ops = _ops()
    rng = np.random.default_rng(M)
    Cin, Cout, real = 256, 16, 15
    x = np.maximum(rng.normal(0, 1, (M, Cin)), 0).astype(np.float32)          # a ReLU output: about half zeros
    x[rng.random((M, Cin)) < 0.05] = 0.0
    g = rng.normal(0, 1, (M, Cout)).astype(np.float32)
    g[:, real:] = 0
    w = rng.normal(0, 0.05, (Cout, Cin)).astype(np.float32)
    w[real:] = 0
    xb, gb, wb = (bf16_round(torch.from_numpy(a)).numpy() for a in (x, g, w))
    xd = torch.from_numpy(xb).to(torch.bfloat16).cuda()
    gd = torch.from_numpy(gb).to(torch.bfloat16).cuda()
    wd = torch.from_numpy(w).cuda()
    ws = torch.empty((ops.conv1x1_thin_bwd_workspace_bytes(),), dtype=torch.uint8, device="cuda")
    outs = []
    for _ in range(2):
        dx = torch.full((M, Cin), 9.0, dtype=torch.bfloat16, device="cuda")
        dw = torch.full((Cout, Cin), 9.0, dtype=torch.float32, device="cuda")
        db = torch.full((Cout,), 9.0, dtype=torch.float32, device="cuda")
        ops.conv1x1_thin_bwd(xd, gd, wd, M, Cin, Cout, dx, dw, db, real, ws)
        torch.cuda.synchronize()
        outs.append((dx.clone(), dw.clone(), db.clone()))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    # forward of the same layer (bd_conv1x1_thin_fwd)
    bias = rng.normal(0, 0.1, (Cout,)).astype(np.float32)
    y = torch.full((M, Cout), 9.0, dtype=torch.bfloat16, device="cuda")
    ops.conv1x1_thin_fwd(xd, wd, torch.from_numpy(bias).cuda(), M, Cin, Cout, y)
    y_ref = xb.astype(np.float64) @ wb.astype(np.float64).T + bias
    assert rel_l2(y.float().cpu(), torch.from_numpy(y_ref)) <= 3e-3
    dx_ref = (gb.astype(np.float64) @ wb.astype(np.float64)) * (xb > 0)
    dw_ref = gb.astype(np.float64).T @ xb.astype(np.float64)
    db_ref = gb.astype(np.float64).sum(0)
    got_dx = outs[0][0].float().cpu().numpy()
    assert rel_l2(torch.from_numpy(got_dx), torch.from_numpy(dx_ref)) <= 3e-3
    assert np.all(got_dx[xb == 0] == 0)
    assert rel_l2(outs[0][1].cpu(), torch.from_numpy(dw_ref)) <= 1e-5
    assert rel_l2(outs[0][2].cpu(), torch.from_numpy(db_ref)) <= 1e-5
    assert np.all(outs[0][1].cpu().numpy()[real:] == 0) and np.all(outs[0][2].cpu().numpy()[real:] == 0)

