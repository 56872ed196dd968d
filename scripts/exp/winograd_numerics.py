"""VERDICT round 4, item 8 (gated experiment): numerics gate of a Winograd F(2x2, 3x3) forward for one head-tower convolution
(256 -> 256, 3x3 / stride 1 / pad 1), evaluated on the CPU BEFORE any kernel is written.

What a gfx950 kernel would compute: input tiles d (4x4) -> V = B^T d B in fp32, rounded to bf16 for the MFMA; weights g (3x3) -> U = G g G^T
in fp32, rounded to bf16; 16 GEMMs M = U . V with fp32 accumulation over the 256 input channels; Y = A^T M A in fp32; bf16 output.
Gate (VERDICT): rel-L2 <= 5e-3 vs the fp32 convolution of the same bf16 operands (the direct kernel: ~2e-3, the bf16 output rounding).
Run: python scripts/exp/winograd_numerics.py  -> profiles/r05_winograd_probe.txt"""
import sys

import numpy as np
import torch
import torch.nn.functional as TF


def bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def winograd_f2x2(x, w, round_uv=True):
    """x: (N, C, H, W) fp32 (bf16-representable), w: (K, C, 3, 3); H, W even.  Returns (N, K, H, W) fp32 before the output rounding."""
    N, C, H, W = x.shape
    K = w.shape[0]
    xp = TF.pad(x, (1, 1, 1, 1))
    # tiles: 4x4 windows at stride 2
    t = xp.unfold(2, 4, 2).unfold(3, 4, 2)                    # (N, C, H/2, W/2, 4, 4)
    V = torch.einsum("ij,nchwjk,lk->nchwil", BT, t, BT)        # B^T d B
    U = torch.einsum("ij,kcjl,ml->kcim", G, w, G)              # G g G^T  (K, C, 4, 4)
    if round_uv:
        V, U = bf(V), bf(U)
    M = torch.einsum("kcij,nchwij->nkhwij", U.double(), V.double()).float()      # fp32-like accumulation (double: no order effects)
    Y = torch.einsum("ij,nkhwjl,ml->nkhwim", AT, M, AT)        # (N, K, H/2, W/2, 2, 2)
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(N, K, H, W)


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def main():
    torch.manual_seed(0)
    out = []
    for name, xs, ws in (("randn activations (post-ReLU: half zeros), He-init weights", 1.0, None),
                         ("activations with a DC offset of 2 sigma (what a biased ReLU layer feeds the next one)", 1.0, "dc")):
        N, C, K, H, W = 1, 256, 256, 50, 84
        x = torch.randn(N, C, H, W)
        if ws == "dc":
            x = x + 2.0
        x = bf(torch.relu(x) * xs)
        w = bf(torch.randn(K, C, 3, 3) * np.sqrt(2.0 / (9 * C)))
        ref = TF.conv2d(x.double(), w.double(), padding=1).float()
        direct = bf(ref)                                        # what the direct kernel stores: fp32 accumulate, one bf16 rounding
        y = winograd_f2x2(x, w, round_uv=True)
        y_exact = winograd_f2x2(x, w, round_uv=False)
        out.append(f"{name}:\n"
                   f"  direct kernel (fp32 accumulate, bf16 store)          rel-L2 {rel(direct, ref):.2e}\n"
                   f"  Winograd F(2x2,3x3), fp32 transforms, no rounding    rel-L2 {rel(y_exact, ref):.2e}\n"
                   f"  Winograd, U and V rounded to bf16 for the MFMA       rel-L2 {rel(y, ref):.2e}   (+ bf16 store: {rel(bf(y), ref):.2e})\n")
    gate = 5e-3
    text = ("# Winograd F(2x2, 3x3) forward for a 256 -> 256 head-tower convolution: numerics gate, CPU simulation (scripts/exp/winograd_numerics.py)\n"
            f"# gate: rel-L2 <= {gate:.0e} vs the fp32 convolution of the same bf16 operands; the layer-by-layer bound of the parity tests is 2e-2, of which\n"
            "# 1.1e-2 is already used by the direct kernels at the deepest tower layer (tests/test_fullsize_parity_gpu.py)\n" + "".join(out))
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
