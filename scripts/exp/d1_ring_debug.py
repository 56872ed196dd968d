"""Where does conv1x1_ring_kernel differ from conv1x1_dense_kernel?  (debug aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from basedet_amd import ops
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.util import pack_weights, bf16_round

for (N, H, W, Cin, Cout) in ((2, 23, 37, 256, 64), (1, 20, 21, 1024, 256), (4, 60, 70, 128, 512)):
    g = torch.Generator().manual_seed(1)
    M = N * H * W
    x = bf16_round(torch.randn(M, Cin, generator=g)).to(torch.bfloat16).cuda()
    w = bf16_round(torch.randn(Cout, Cin, 1, 1, generator=g) / np.sqrt(Cin))
    wf, wd = pack_weights(ops, w)
    bias = torch.randn(Cout, generator=g).cuda()
    res = bf16_round(torch.randn(M, Cout, generator=g)).to(torch.bfloat16).cuda()
    geo = ops.single(N, H, W)
    d = ops.conv_desc(geo, geo, Cin, Cout, 1, 1, 1, 0)
    out = {}
    for name, kw in (("plain", dict()), ("bias", dict(bias=bias)), ("add", dict(add=res, flags=ops.EPI_ADD_BEFORE)), ("relu", dict(flags=ops.EPI_RELU))):
        for mode in (3, 1):
            ops.L().bd_conv_set_dense1x1(mode)
            y = torch.full((M, Cout), 7.0, dtype=torch.bfloat16, device="cuda")
            ops.conv2d_fwd(d, x, wf, kw.get("bias"), y, add=kw.get("add"), flags=kw.get("flags", 0))
            torch.cuda.synchronize()
            out[mode] = y.float().cpu().numpy()
        bad = out[3] != out[1]
        print(f"{Cin}->{Cout} M={M} [{name}]: {bad.mean() * 100:.2f} % differ; max |diff| {np.abs(out[3] - out[1]).max():.4f}")
        if bad.any():
            pm, ch = np.nonzero(bad)
            print("   by pixel tile:", np.bincount(pm // 128, minlength=(M + 127) // 128)[:20])
            print("   by pixel in tile (16-blocks):", np.bincount((pm % 128) // 16, minlength=8))
            print("   by channel (8-blocks):", np.bincount(ch // 8, minlength=Cout // 8)[:32])
            i = np.argmax(np.abs(out[3] - out[1]))
            print("   e.g.", pm[0], ch[0], out[3][pm[0], ch[0]], out[1][pm[0], ch[0]], " worst", out[3].flat[i], out[1].flat[i])
ops.L().bd_conv_set_dense1x1(1)
