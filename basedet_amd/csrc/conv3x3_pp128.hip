// 3x3 / stride-1 / pad-1 convolution forward and data-gradient: the staggered patch kernel of conv3x3_pp.hip with a 128-channel tile.
//
// For layers with <= 128 output channels (res3) and for grids too small for the 256-channel tile to fill 256 CUs (res4 at 50x84).
// Same structure -- four 4x16 patches per workgroup, padded activation image re-staged once per K block, weight ring filled by
// buffer LDS-DMA, two wave groups (the two 64-channel halves) one barrier apart, tied-accumulator MFMAs -- with:
//   * wave tile 64 x 64 (acc[4][4]); one tap (K = 64) = two phases of 16 MFMAs (K halves): 8 ds_read_b128 per phase;
//   * weight ring of FOUR 16 KB tap slots, tap T's two pieces per wave issued in phases (T-3, 1) and (T-2, 0) and retired by
//     vmcnt(2) in phase (T-1, 1): nine taps per K block do not divide the ring, so the slot index (cb + t) & 3 is a run-time scalar.
// LDS 126 KB, one workgroup per CU.
#include "conv3x3_pp128_body.h"

namespace {
using namespace pp128;

struct PSeg { int patch_start, H, W, pw, src_off, dst_off; };

struct PParams {
    const bf16_raw* src;
    const bf16_raw* w;       // [CO][9][CK]
    const float* bias;
    const bf16_raw* add;
    const bf16_raw* mask;
    bf16_raw* dst;
    int CK, CO, flags, nseg;
    int src_ppi, dst_ppi;
    unsigned src_bytes;
    int patches_per_img, total_patches, n_tiles;
    int patch_begin;         // first patch of this launch; total_patches = end of the range (conv3x3_pp.hip hands its tail tiles over)
    PSeg seg[MAX_SEG];
};


template <int MODE, int TCO>
__global__ __launch_bounds__(512, 1) void conv3x3_pp128_kernel(const PParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    pp128::body<MODE, TCO>(p, smem, (int)blockIdx.x, (int)gridDim.x, p.patch_begin, p.total_patches, p.n_tiles);
}

}  // namespace

// 0 = launched, 1 = shape not handled here (caller falls back to conv3x3.hip)
extern BD_KNOB int g_patch_pp128;     // 0 = the 64-channel tile for Cout <= 64 only (default), 1 = every shape, -1 = never
// patch range [patch_begin, patch_end) (patch_end < 0: all) on the tco-channel tile (0: by shape)
int bd_conv3x3_pp128_launch_range(const bd_conv_desc* d, int mode, const void* src, const void* w, const float* bias, const void* add,
                                  const void* mask, void* dst, int flags, hipStream_t stream, int patch_begin, int patch_end, int tco) {
    PParams p{};
    p.CK = mode == 0 ? d->Cin : d->Cout;
    p.CO = mode == 0 ? d->Cout : d->Cin;
    p.src_ppi = mode == 0 ? d->in_pix_per_img : d->out_pix_per_img;
    p.dst_ppi = mode == 0 ? d->out_pix_per_img : d->in_pix_per_img;
    if (p.CK % 8 != 0 || p.CO % 8 != 0) return 1;
    if ((long long)d->N * p.src_ppi * p.CK * 2 >= 0x7fffffffll || (long long)d->N * p.dst_ppi >= 0x7fffffffll ||
        (long long)p.CO * 9 * p.CK >= 0x7fffffffll) return 1;
    p.src = (const bf16_raw*)src; p.w = (const bf16_raw*)w; p.bias = bias;
    p.add = (const bf16_raw*)add; p.mask = (const bf16_raw*)mask; p.dst = (bf16_raw*)dst;
    p.flags = flags; p.nseg = d->nseg;
    p.src_bytes = (unsigned)((long long)d->N * p.src_ppi * p.CK * 2);
    int ps = 0;
    for (int s = 0; s < d->nseg; ++s) {
        PSeg& sg = p.seg[s];
        sg.patch_start = ps; sg.H = d->Ho[s]; sg.W = d->Wo[s]; sg.pw = cdiv(d->Wo[s], PW);
        sg.src_off = mode == 0 ? d->in_off[s] : d->out_off[s];
        sg.dst_off = mode == 0 ? d->out_off[s] : d->in_off[s];
        ps += cdiv(d->Ho[s], PH) * sg.pw;
    }
    p.patches_per_img = ps;
    p.total_patches = patch_end >= 0 ? patch_end : ps * d->N;
    p.patch_begin = patch_begin;
    const bool small = tco ? tco == 64 : p.CO <= 64;            // 64-channel tile: no matrix work on absent channels (res2 3x3, bbox_pred)
    if (!tco && (g_patch_pp128 < 0 || (g_patch_pp128 == 0 && !small))) return 1;
    p.n_tiles = cdiv(p.CO, small ? 64 : 128);
    const int grid = cdiv(p.total_patches - p.patch_begin, NPATCH) * p.n_tiles;
    if (grid <= 0) return 0;
    BD_ONCE_PER_DEVICE(
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_pp128_kernel<0, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, Tile<128>::LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_pp128_kernel<1, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, Tile<128>::LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_pp128_kernel<0, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, Tile<64>::LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_pp128_kernel<1, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, Tile<64>::LDS_BYTES));
    if (!tco) bd_note_kernel("conv3x3_pp128_kernel");          // (tco != 0: the tail of a conv3x3_pp_kernel launch, named there)
    if (small) {
        if (mode == 0) hipLaunchKernelGGL((conv3x3_pp128_kernel<0, 64>), dim3(grid), dim3(512), Tile<64>::LDS_BYTES, stream, p);
        else hipLaunchKernelGGL((conv3x3_pp128_kernel<1, 64>), dim3(grid), dim3(512), Tile<64>::LDS_BYTES, stream, p);
    } else {
        if (mode == 0) hipLaunchKernelGGL((conv3x3_pp128_kernel<0, 128>), dim3(grid), dim3(512), Tile<128>::LDS_BYTES, stream, p);
        else hipLaunchKernelGGL((conv3x3_pp128_kernel<1, 128>), dim3(grid), dim3(512), Tile<128>::LDS_BYTES, stream, p);
    }
    return 0;
}

int bd_conv3x3_pp128_launch(const bd_conv_desc* d, int mode, const void* src, const void* w, const float* bias, const void* add,
                            const void* mask, void* dst, int flags, hipStream_t stream) {
    return bd_conv3x3_pp128_launch_range(d, mode, src, w, bias, add, mask, dst, flags, stream, 0, -1, 0);
}
