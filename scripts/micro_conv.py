import sys, torch, time
sys.path.insert(0, '.')
from basedet_amd import ops
def bench(N, H, W, Cin, Cout, R=3, stride=1, pad=1, mode="fwd", iters=10):
    gin = ops.single(N, H, W); gout = gin.conv_out(R, stride, pad)
    d = ops.conv_desc(gin, gout, Cin, Cout, R, R, stride, pad)
    x = torch.randn(gin.pixels, Cin, device="cuda").to(torch.bfloat16)
    w = (torch.randn(Cout, R * R, Cin, device="cuda") * 0.02).to(torch.bfloat16)
    wt = (torch.randn(Cin, R * R, Cout, device="cuda") * 0.02).to(torch.bfloat16)
    y = torch.empty(gout.pixels, Cout, device="cuda", dtype=torch.bfloat16)
    g = torch.randn(gout.pixels, Cout, device="cuda").to(torch.bfloat16)
    dx = torch.empty_like(x)
    ws = torch.empty(ops.conv2d_wgrad_workspace_bytes(d) // 4 + 16, device="cuda")
    dw = torch.empty(Cout, R, R, Cin, device="cuda")
    def run():
        if mode == "fwd": ops.conv2d_fwd(d, x, w, None, y, flags=ops.EPI_RELU)
        elif mode == "dgrad": ops.conv2d_dgrad(d, g, wt, dx)
        else: ops.conv2d_wgrad(d, x, g, dw, ws)
    for _ in range(3): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): run()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    fl = 2.0 * gout.pixels * Cin * Cout * R * R
    print(f"{mode:6s} N={N} {H}x{W} Cin={Cin:5d} Cout={Cout:5d} R={R} s={stride}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s", flush=True)
if __name__ == "__main__":
    for cin in (64, 128, 256, 512, 1024, 2048):
        bench(16, 100, 168, cin, 256)
    for cout in (128, 256, 512, 1024):
        bench(16, 100, 168, 256, cout)
    for mode in ("dgrad", "wgrad"):
        for cin in (256, 1024):
            bench(16, 100, 168, cin, 256, mode=mode)
