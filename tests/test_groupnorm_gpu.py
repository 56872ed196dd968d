"""GroupNorm(32, 256) + ReLU of the FCOS PointHead towers (basedet/layers/head/point_head.py:47-58) on the HIP path against
torch.nn.functional.group_norm on the CPU in fp32, per (image, pyramid level): forward value and statistics, and the backward
(dy, dgamma, dbeta) with the ReLU gate recomputed from y (round 5: bd_groupnorm_bwd no longer reads z).  Tolerances: bf16 output
rounding (rel-L2 <= 4e-3 forward, 6e-3 backward); two launches must agree bit for bit."""
import numpy as np
import pytest
import torch
import torch.nn.functional as TF

from tests.util import rel_l2

pytestmark = pytest.mark.gpu
C = 256


def _reference(y, dz, gamma, beta, geom, relu=True):
    """y, dz: fp32 (pixels, C) on the CPU in the pixel-major multi-level layout."""
    N = geom.N
    yv = y.view(N, geom.pix_per_img, C)
    dzv = dz.view(N, geom.pix_per_img, C)
    z = torch.zeros_like(yv)
    dy = torch.zeros_like(yv)
    g = gamma.clone().requires_grad_(True)
    b = beta.clone().requires_grad_(True)
    dg = torch.zeros(C)
    db = torch.zeros(C)
    for i in range(geom.nlev):
        o, n = geom.off[i], geom.H[i] * geom.W[i]
        x = yv[:, o:o + n].permute(0, 2, 1).contiguous().requires_grad_(True)          # (N, C, n)
        out = TF.group_norm(x, 32, g, b, eps=1e-5)
        if relu:
            out = TF.relu(out)
        z[:, o:o + n] = out.detach().permute(0, 2, 1)
        gx, gg, gb = torch.autograd.grad(out, (x, g, b), dzv[:, o:o + n].permute(0, 2, 1).contiguous())
        dy[:, o:o + n] = gx.permute(0, 2, 1)
        dg += gg
        db += gb
    return z.reshape(-1, C), dy.reshape(-1, C), dg, db


@pytest.mark.parametrize("N,sizes", [(3, [(25, 42), (13, 21), (7, 11)]), (9, [(20, 30), (10, 15), (5, 8), (3, 4), (2, 2)])])
def test_groupnorm_fwd_bwd_matches_torch_and_is_reproducible(N, sizes):
    from basedet_amd import ops
    geom = ops.Geom(N, [h for h, _ in sizes], [w for _, w in sizes])
    gen = torch.Generator().manual_seed(17 + N)
    y = (torch.randn(geom.pixels, C, generator=gen) * 1.5 + 0.3).to(torch.bfloat16)
    dz = torch.randn(geom.pixels, C, generator=gen).to(torch.bfloat16)
    gamma = torch.rand(C, generator=gen) + 0.5
    beta = torch.randn(C, generator=gen) * 0.3
    z_ref, dy_ref, dg_ref, db_ref = _reference(y.float(), dz.float(), gamma, beta, geom)
    yd, dzd, gd, bd = y.cuda(), dz.cuda(), gamma.cuda(), beta.cuda()
    ws = torch.empty((ops.groupnorm_workspace_bytes(N, geom.nlev, C, geom.pix_per_img) // 4 + 16,), dtype=torch.float32, device="cuda")
    outs = []
    for rep in range(2):          # (two launches: the reductions are fixed-order -- the image-chunked schedules of rounds 4-5 are gone)
        stats = torch.empty((N, geom.nlev, 32, 2), dtype=torch.float32, device="cuda")
        z = torch.empty_like(yd)
        dy = torch.empty_like(yd)
        dg = torch.full((C,), 3.0, device="cuda")
        db = torch.full((C,), 3.0, device="cuda")
        ops.groupnorm_fwd(yd, gd, bd, geom, C, 1e-5, True, stats, z, ws)
        ops.groupnorm_bwd(dzd, yd, gd, bd, stats, geom, C, True, dy, dg, db, ws)
        torch.cuda.synchronize()
        outs.append((z.clone(), dy.clone(), dg.clone(), db.clone(), stats.clone()))
    z, dy, dg, db, stats = outs[0]
    assert rel_l2(z.float().cpu(), z_ref) < 4e-3
    assert rel_l2(dy.float().cpu(), dy_ref) < 6e-3
    assert rel_l2(dg.cpu(), dg_ref) < 2e-3 and rel_l2(db.cpu(), db_ref) < 2e-3
    # statistics of the first level of image 0 against the definition
    n0 = geom.H[0] * geom.W[0]
    x0 = y.float()[:n0].view(n0, 32, 8)
    mean = x0.mean(dim=(0, 2))
    rstd = 1.0 / torch.sqrt(x0.var(dim=(0, 2), unbiased=False) + 1e-5)
    assert torch.allclose(stats[0, 0, :, 0].cpu(), mean, rtol=1e-4, atol=1e-5)
    assert torch.allclose(stats[0, 0, :, 1].cpu(), rstd, rtol=1e-4, atol=1e-5)
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert torch.equal(a, b)
    # accumulate adds on top
    dg2, db2 = dg.clone(), db.clone()
    ops.groupnorm_bwd(dzd, yd, gd, bd, stats, geom, C, True, dy, dg2, db2, ws, accumulate=True)
    assert torch.allclose(dg2, 2 * dg, rtol=1e-6) and torch.allclose(db2, 2 * db, rtol=1e-6)


@pytest.mark.parametrize("N,sizes", [(3, [(25, 42), (13, 21), (7, 11)]), (16, [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)])])
def test_groupnorm_statistics_from_the_convolution_epilogue(N, sizes):
    """Round 6: conv -> GroupNorm -> ReLU with the statistics pass fused into the convolution (bd_conv2d_fwd_gnstats + bd_groupnorm_fwd_parts)
    against the separate form (bd_conv2d_fwd + bd_groupnorm_fwd).  The convolution's output is the same bits; the statistics -- summed from the
    fp32 results before the bf16 rounding instead of from the rounded tensor -- agree to 2e-3 (mean: of the group's standard deviation; rstd:
    relative); z agrees to a bf16 rounding step of |z| + 1; mean and rstd also match the definition on the fp32 convolution result; two
    launches agree bit for bit.  The second case is the benchmark's pyramid (1 536 tiles, six per workgroup, patches that overhang the image)."""
    from basedet_amd import ops
    from tests.util import pack_weights
    geom = ops.Geom(N, [h for h, _ in sizes], [w for _, w in sizes])
    gen = torch.Generator().manual_seed(23 + N)
    x = torch.randn(geom.pixels, C, generator=gen).to(torch.bfloat16).cuda()
    w = (torch.randn(C, C, 3, 3, generator=gen) / np.sqrt(9 * C)).to(torch.bfloat16).float()
    bias = (torch.randn(C, generator=gen) * 0.5).cuda()
    gamma = (torch.rand(C, generator=gen) + 0.5).cuda()
    beta = (torch.randn(C, generator=gen) * 0.3).cuda()
    wf, _ = pack_weights(ops, w)
    d = ops.conv_desc(geom, geom, C, C, 3, 3, 1, 1)
    ws = torch.empty((ops.groupnorm_workspace_bytes(N, geom.nlev, C, geom.pix_per_img) // 4 + 16,), dtype=torch.float32, device="cuda")
    y0 = torch.empty((geom.pixels, C), dtype=torch.bfloat16, device="cuda")
    ops.conv2d_fwd(d, x, wf, bias, y0)
    st0 = torch.empty((N, geom.nlev, 32, 2), dtype=torch.float32, device="cuda")
    z0 = torch.empty_like(y0)
    ops.groupnorm_fwd(y0, gamma, beta, geom, C, 1e-5, True, st0, z0, ws)
    outs = []
    for rep in range(2):
        part = torch.full((ops.conv2d_fwd_gnstats_bytes(d) // 4,), float("nan"), dtype=torch.float32, device="cuda")
        y1 = torch.full((geom.pixels, C), float("nan"), dtype=torch.bfloat16, device="cuda")
        ops.conv2d_fwd_gnstats(d, x, wf, bias, y1, part)
        assert ops.L().bd_conv_last_kernel().decode() == "conv3x3_pp_kernel"
        st1 = torch.full((N, geom.nlev, 32, 2), float("nan"), dtype=torch.float32, device="cuda")
        z1 = torch.full((geom.pixels, C), float("nan"), dtype=torch.bfloat16, device="cuda")
        ops.groupnorm_fwd_parts(d, y1, part, gamma, beta, 1e-5, True, st1, z1)
        torch.cuda.synchronize()
        outs.append((y1, st1, z1, part))
    y1, st1, z1, part = outs[0]
    assert torch.equal(y1, y0), "the fused launch's convolution output differs"
    assert bool(torch.isfinite(part).all()) and bool(torch.isfinite(st1).all())
    std0 = 1.0 / st0[..., 1]
    assert float(((st1[..., 0] - st0[..., 0]).abs() / std0).max()) < 2e-3
    assert float(((st1[..., 1] - st0[..., 1]).abs() / st0[..., 1]).max()) < 2e-3
    dz = (z1.float() - z0.float()).abs()
    assert bool((dz <= (z0.float().abs() + 1.0) * 2.0 ** -7).all()), float(dz.max())
    assert rel_l2(z1.float().cpu(), z0.float().cpu()) < 2e-3
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    # against the definition, on the first level of image 0 (fp32 convolution of the same bf16 operands)
    H, W = sizes[0]
    xi = x[: H * W].float().cpu().view(1, H, W, C).permute(0, 3, 1, 2)
    yi = TF.conv2d(xi, w, bias.cpu(), padding=1)[0].permute(1, 2, 0).reshape(H * W, 32, 8)
    mean = yi.mean(dim=(0, 2))
    rstd = 1.0 / torch.sqrt(yi.var(dim=(0, 2), unbiased=False) + 1e-5)
    assert torch.allclose(st1[0, 0, :, 0].cpu(), mean, rtol=1e-3, atol=1e-3)
    assert torch.allclose(st1[0, 0, :, 1].cpu(), rstd, rtol=1e-3, atol=1e-4)
