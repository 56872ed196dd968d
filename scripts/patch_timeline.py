"""Timeline of one workgroup of the 3x3 patch kernel (s_memtime stamps): python scripts/patch_timeline.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from basedet_amd import ops
lib = ops.L()
lib.bd_conv3x3_set_debug.argtypes = [ctypes.c_void_p]
for (N, H, W, Cin, Cout) in ((16, 100, 168, 256, 256), (16, 100, 168, 128, 128)):
    gin = ops.single(N, H, W); d = ops.conv_desc(gin, gin, Cin, Cout, 3, 3, 1, 1)
    x = torch.randn(gin.pixels, Cin, device="cuda").to(torch.bfloat16)
    w = (torch.randn(Cout, 9, Cin, device="cuda") * 0.02).to(torch.bfloat16)
    y = torch.empty(gin.pixels, Cout, device="cuda", dtype=torch.bfloat16)
    dbg = torch.zeros(8 * 80, dtype=torch.int64, device="cuda")
    for _ in range(3):
        ops.conv2d_fwd(d, x, w, None, y, flags=ops.EPI_RELU)
    lib.bd_conv3x3_set_debug(ctypes.c_void_p(dbg.data_ptr()))
    ops.conv2d_fwd(d, x, w, None, y, flags=ops.EPI_RELU)
    torch.cuda.synchronize()
    lib.bd_conv3x3_set_debug(ctypes.c_void_p(0))
    v = dbg.cpu().view(8, 80).tolist()
    t0 = min(v[w][1] for w in range(8))
    print(f"Cin={Cin}: per wave (shader cycles since the first wave started); stamps: start, prologue-end, then per step [compute issued, barrier passed(, x swapped)] ..., end")
    for w in range(8):
        n = v[w][0]; st = [x - t0 for x in v[w][1:1 + n]]
        print(f"wave {w}: start {st[0]:5d} | deltas", [st[i + 1] - st[i] for i in range(n - 1)][:12], "| end", st[-1])
