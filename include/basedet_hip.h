/*
 * basedet_hip.h -- flat C ABI of libbasedet_hip.so (MI355X / gfx950).
 *
 * The reference (megvii-research/basedet) has no FFI/plugin ABI of its own: its hot path sits behind two
 * Python surfaces (basedet.layers / basedet.structures operator signatures, and the BaseNet/Solver protocol)
 * and all arithmetic is delegated to MegEngine.  Each entry point below therefore cites the reference
 * *Python call site* (file:line under basedet/) whose MegEngine kernels it replaces.
 *
 * Conventions
 *   - every function returns 0 (BD_OK) or a negative BD_E* code; bd_last_error_string() gives a thread-local message;
 *     no C++ exception crosses the ABI.
 *   - all pointers are caller-owned DEVICE memory unless the name ends in _host; the library never allocates or
 *     frees user tensors.  Scratch comes from a caller-provided workspace (query with *_workspace_bytes).
 *   - every call is asynchronous and ordered on `stream` (a hipStream_t passed as void*); no host sync inside.
 *   - activations are NHWC bf16 ("pixel-major": element (n, y, x, c) at ((n*pix_per_img + off + y*W + x)*C + c)),
 *     master weights/gradients are fp32 [Cout][R][S][Cin]; packed bf16 weight copies are made by bd_weight_pack.
 *   - integer/index results (labels, match indices, NMS keep lists, anchors) are bit-exact w.r.t. oracle/box_ops.py;
 *     these kernels are compiled with -ffp-contract=off and IEEE division.
 */
#ifndef BASEDET_HIP_H
#define BASEDET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BD_OK 0
#define BD_EINVAL (-1)     /* bad argument / unsupported shape */
#define BD_ELAUNCH (-2)    /* HIP launch failure */
#define BD_EWORKSPACE (-3) /* workspace too small */

#define BD_MAX_SEGS 8

/* epilogue flags for bd_conv2d_fwd / bd_conv2d_dgrad */
#define BD_EPI_RELU 1            /* y = max(y, 0)                                         */
#define BD_EPI_ADD_BEFORE 2      /* y = acc + bias + add[idx]   (residual add, then relu/mask) */
#define BD_EPI_ADD_AFTER 4       /* y = mask(acc + bias) + add[idx]                        */
#define BD_EPI_MASK 8            /* y = mask[idx] > 0 ? y : 0   (ReLU backward from the stored forward output) */
#define BD_EPI_SPARSE 16         /* strided data gradient accumulating in place (BD_EPI_ADD_BEFORE, add == dx): input pixels that no
                                  * filter tap reaches (3 of 4 for a 1x1 / stride-2 shortcut) are left untouched -- the caller states that
                                  * dx there already holds its final (masked) value */

typedef void* bd_stream_t; /* hipStream_t */

/* Geometry of one convolution over up to BD_MAX_SEGS pyramid levels that share weights
 * (RetinaNetHead applies the same convs to P3..P7: layers/head/retina_head.py:103-112).
 * A single-level conv uses nseg = 1, in_off = out_off = 0, *_pix_per_img = H*W. */
typedef struct bd_conv_desc {
    int32_t N;                 /* images */
    int32_t Cin, Cout;
    int32_t R, S;              /* kernel height / width */
    int32_t stride, pad;
    int32_t nseg;
    int32_t Hi[BD_MAX_SEGS], Wi[BD_MAX_SEGS];   /* input  size per level */
    int32_t Ho[BD_MAX_SEGS], Wo[BD_MAX_SEGS];   /* output size per level */
    int32_t in_off[BD_MAX_SEGS], out_off[BD_MAX_SEGS]; /* pixel offset of the level inside one image */
    int32_t in_pix_per_img, out_pix_per_img;    /* pixel stride between images */
    /* Kernel ROUTING of this call, for parity tests and A/B harnesses (production callers leave all of it 0 = the library's choice).
     * Round 6: these words replace the process-global bd_*_set_* knobs of rounds 1-5 -- the library keeps NO routing state between calls.
     * A word holds (route + 1):
     *   route[0]  dense 1x1 launches (1x1 / stride 1 over one dense level), mode 0 .. 6: 1 (default) = conv1x1_ring.hip takes the short-K
     *             launches into >= 256 channels, conv1x1.hip the rest; 0 = the generic kernel (the *_bits / _ex entry points then return
     *             BD_EINVAL); 2 = conv1x1.hip's 256 x 256 tile, 3 = its 128 x 128 tile only, 4 = its eight-wave 256-channel x 128-pixel
     *             tile wherever legal, 5 = as 1 with conv1x1_ring.hip for every launch it can take, 6 = as 3 with the 128 x 128 tile's
     *             LDS-DMA ring variant for every K that allows it (all: the same results bit for bit); bd_conv1x1_fp8 honours 1 / 3 / 5;
     *   route[1]  3x3 / generic path, a bit mask (default 3): see "route[1] bits" below;
     *   route[2]  weight gradients, a bit mask (default 1): bit 0 = operand transposes through ds_read_b64_tr_b16 (clear: scalar 16-bit
     *             LDS reads, the slow reference path that validates the transposing read); bit 1 set = no nine-tap / 1x1 kernels (generic
     *             weight-gradient kernel); bit 2 set = no ring-staged kernels (conv_wgrad3x3_ring.hip, conv_wgrad1x1_ring.hip);
     *   route[3]  fp8 forward (bd_conv2d_fwd_fp8*): 1 (default) = conv3x3_pp8.hip where the shape allows, 0 = the generic per-tap kernel.
     * sr_seed != 0: the e5m2 twins this call writes (the dx8 outputs of bd_conv2d_dgrad_ex / bd_conv2d_dgrad_fp8 / bd_conv1x1_fp8 mode 1)
     * are rounded STOCHASTICALLY -- a hash of (seed, element index) below the kept bits -- instead of to nearest (a training
     * hyper-parameter of the fp8 configuration, not a route). */
    int32_t route[4];
    uint32_t sr_seed;
} bd_conv_desc;

const char* bd_last_error_string(void);
int bd_version(void);
/* Name of the device kernel (as rocprofv3 prints it, template arguments dropped) that the calling thread's LAST convolution entry point
 * (bd_conv2d_fwd* / _dgrad* / _wgrad* / bd_conv1x1_fp8 / bd_conv2d_*_fp8) dispatched to; "" before the first call.  Written by the launch
 * sites themselves, so a measurement harness attributes time to the kernel that really ran (tools/benchmark.py:125-140 has no
 * counterpart: MegEngine's profiler names its own kernels). */
const char* bd_conv_last_kernel(void);
/* Measurement probe, SYNCHRONOUS (not an operator): the dense bf16 TFLOP/s this device's matrix pipes sustain on the register-level
 * pattern of the convolution kernels (rotating random fragments, 32 accumulators of v_mfma_f32_16x16x32_bf16, no LDS / memory) after
 * `seconds` (<= 30) of back-to-back launches on `stream`; clock_mhz_out (may be NULL) = the in-kernel clock held under that load.
 * bench.py's roofline.peak_measured (the harness it mirrors: tools/benchmark.py:125-140). */
int bd_probe_mfma_rate(double seconds, double* tflops_out, double* clock_mhz_out, bd_stream_t stream);

/* Measurement probe, SYNCHRONOUS: the clock the chip HELD inside `kernel` ("conv3x3_pp_kernel", "conv_wgrad3x3_ring_kernel": the two
 * MFMA-bound kernels that dominate a step), averaged over all its launches since the last reset -- workgroup 0 of every launch adds the
 * shader-cycle and 100 MHz differences of its lifetime to two device counters.  reset != 0 clears the sums afterwards.  busy_ms_out (may
 * be NULL): the summed lifetime of those workgroups.  bench.py's roofline.clock_mhz / frac_in_cycles (tools/benchmark.py:125-140 has no
 * counterpart). */
int bd_probe_kernel_clock(const char* kernel, int reset, double* clock_mhz_out, double* busy_ms_out);

/* ---------------------------------------------------------------------------------------------------------
 * Dense path: convolution forward / data-gradient / weight-gradient (MFMA implicit GEMM).
 * Replaces M.Conv2d + norm + F.relu + residual add at models/cls/resnet.py:28-52,70-113,142-146,
 * layers/backbone/fpn_backbone.py:61-76,123-160,198-204, layers/head/retina_head.py:49-70,103-112 and
 * their MegEngine autodiff backward (GradManager.backward, solver/default_solver.py:118-124).
 * ------------------------------------------------------------------------------------------------------- */

/* y = epi(conv(x, w) + bias [+ add]); w_packed = [Cout][R][S][Cin] bf16 (bd_weight_pack fwd layout).
 * Requires Cin % 8 == 0 and Cout % 8 == 0. bias may be NULL. add/mask per flags (same shape as y). */
int bd_conv2d_fwd(const bd_conv_desc* d, const void* x, const void* w_packed, const float* bias,
                  const void* add, void* y, int flags, bd_stream_t stream);

/* dx = epi(conv_transpose(g, w) [+ add]) ; w_packed_t = [Cin][R][S][Cout] bf16 (bd_weight_pack dgrad layout).
 * g has the conv's OUTPUT geometry, dx/add/mask the INPUT geometry. Requires Cout % 8 == 0, Cin % 8 == 0. */
int bd_conv2d_dgrad(const bd_conv_desc* d, const void* g, const void* w_packed_t, const void* add,
                    const void* mask, void* dx, int flags, bd_stream_t stream);

/* Bit-packed ReLU masks for the HBM-bound 1x1 layers.  A forward launch with BD_EPI_RELU may also write ybits: one bit per output
 * element (y > 0), uint32 [Cout/32][M] (M = N * pixels of the level; word (g, m) holds channels 32g .. 32g+31 of pixel m, bit b =
 * channel 32g + b).  A data-gradient launch reads such a tensor (of ITS output geometry, i.e. the conv's input) instead of the bf16
 * activation as its BD_EPI_MASK operand: a 16x smaller stream (models/cls/resnet.py:70-113: the block input's ReLU gate in conv1's
 * backward).  Only the dense 1x1 kernel (1x1 / stride 1 / pad 0 over one dense level, channels % 32 == 0, tensors < 2 GB) serves
 * these two; any other descriptor returns BD_EINVAL. */
int bd_conv2d_fwd_bits(const bd_conv_desc* d, const void* x, const void* w_packed, const float* bias, const void* add, void* y,
                       uint32_t* ybits, int flags, bd_stream_t stream);
int bd_conv2d_dgrad_bits(const bd_conv_desc* d, const void* g, const void* w_packed_t, const void* add, const uint32_t* maskbits,
                         void* dx, int flags, bd_stream_t stream);
/* bd_conv2d_fwd with both optional side outputs of the dense 1x1 kernel: ybits (may be NULL) as above, and y8 (may be NULL) =
 * e4m3(clamp(y * q_scale)), the input of a following fp8 3x3 convolution (bd_conv2d_fwd_fp8: saves its cast pass). */
int bd_conv2d_fwd_ex(const bd_conv_desc* d, const void* x, const void* w_packed, const float* bias, const void* add, void* y,
                     uint32_t* ybits, void* y8, float q_scale, int flags, bd_stream_t stream);
/* bd_conv2d_dgrad of the dense 1x1 kernel with its optional side input / output: maskbits (may be NULL: then `mask` is the bf16 form)
 * and dx8 (may be NULL) = e5m2(clamp(dx * q_scale)), the gradient operand of a following bd_conv2d_dgrad_fp8. */
int bd_conv2d_dgrad_ex(const bd_conv_desc* d, const void* g, const void* w_packed_t, const void* add, const void* mask,
                       const uint32_t* maskbits, void* dx, void* dx8, float q_scale, int flags, bd_stream_t stream);

/* dw[Cout][R][S][Cin] (fp32) = sum over pixels g^T x, times row_scale[Cout] (NULL = 1); split over pixels with
 * fp32 partial slabs in ws, reduced in a fixed order (bitwise reproducible).  accumulate != 0 adds to dw. */
size_t bd_conv2d_wgrad_workspace_bytes(const bd_conv_desc* d);
int bd_conv2d_wgrad(const bd_conv_desc* d, const void* x, const void* g, const float* row_scale, float* dw,
                    int accumulate, void* ws, size_t ws_bytes, bd_stream_t stream);

/* bd_conv2d_wgrad plus the bias gradient dbias[Cout] = sum over all output pixels of g (the reference's autodiff of `conv + bias`).
 * The nine-tap 3x3 kernel sums the columns of g while it stages them (no second pass over g); the other kernels are followed by
 * bd_colsum_bf16 per pyramid level.  accumulate applies to dw and dbias alike.  Fixed-order reductions (reproducible). */
size_t bd_conv2d_wgrad_bias_workspace_bytes(const bd_conv_desc* d);
int bd_conv2d_wgrad_bias(const bd_conv_desc* d, const void* x, const void* g, const float* row_scale, float* dw, float* dbias,
                         int accumulate, void* ws, size_t ws_bytes, bd_stream_t stream);

/* Deferred weight-gradient reduces (round 4).  bd_conv2d_wgrad / _bias run a layer's partial-sum kernel and then its fixed-order reduce:
 * two to three launches per layer, 60 reduce launches per RetinaNet-R50 step.  With a queue the caller runs the partial-sum kernels of
 * several layers back to back (bd_conv2d_wgrad_queued: same arguments and results as bd_conv2d_wgrad_bias, dbias may be NULL) and ONE
 * launch (bd_wgrad_queue_flush) reduces all of them, in the same fixed order as the per-layer reduce (bitwise identical results).
 * Each queued layer needs its OWN workspace, left untouched until the flush; dw / dbias are valid after the flush, in `stream` order.
 * Replaces the same call sites as bd_conv2d_wgrad (MegEngine's autodiff of the convolutions, solver/default_solver.py:118-124). */
typedef struct bd_wgrad_queue* bd_wgrad_queue_t;
int bd_wgrad_queue_create(bd_wgrad_queue_t* out);
int bd_wgrad_queue_destroy(bd_wgrad_queue_t q);
int bd_wgrad_queue_pending(bd_wgrad_queue_t q);     /* reduce descriptors waiting for a flush */
int bd_conv2d_wgrad_queued(bd_wgrad_queue_t q, const bd_conv_desc* d, const void* x, const void* g, const float* row_scale, float* dw,
                           float* dbias, int accumulate, void* ws, size_t ws_bytes, bd_stream_t stream);
int bd_wgrad_queue_flush(bd_wgrad_queue_t q, bd_stream_t stream);

/* bd_conv_desc.route[1] bits (the word holds mask + 1; default mask 3): bit 0 = 3x3/stride-1 forward and dgrad use the patch kernel
 * (conv3x3.hip); bit 1 = BK=32 tiles for 1x1 convs in the generic kernel; bit 2 = unused.
 * 0 = everything through the generic per-tap implicit GEMM (conv_igemm.hip).
 * Ablation bits (set = feature OFF unless noted): bit 3 = register-staged instead of LDS-DMA weights in the patch kernel;
 * bit 4 = (set = ON; -DBD_AB_SKIP diagnostic builds ONLY -- the shipped library answers BD_EINVAL) timing A/B: every 3x3 / stride-2 forward
 * and data-gradient launch returns at once, its output keeps whatever it held (how much of a step the strided launches are); bit 5 = no early epilogue-operand prefetch in the generic kernel;
 * bit 6 = no staggered 256-channel patch instance (conv3x3_pp.hip; by default taken for every Cout > 128, Cin % 8 == 0 shape);
 * bit 7 = take it only where its grid fills the chip better than the 128-channel instance's (makespan estimate);
 * bit 8 = (set = ON) staggered 128-channel patch instance (conv3x3_pp128.hip) for every other 3x3/stride-1 shape (its 64-channel
 * tile serves Cout <= 64 by default); bit 9 = none of the conv3x3_pp128.hip instances; bit 10 = BK=64 tiles for the stride-2 3x3
 * launches of the generic kernel (by default BK=32 when the grid has >= 512 tiles: more workgroups per CU for their short K loops);
 * bit 11 = 64-bit pointer staging in the generic kernel instead of range-checked buffer loads (taken by default when the source tensor
 * and the packed weights are < 2 GB each); bit 12 = the 1x1 / stride-2 launches over one dense level (forward, and the data gradient
 * with BD_EPI_SPARSE) stay on the generic kernel instead of the dense 1x1 kernel with strided rows;
 * bit 13 = no tail split in conv3x3_pp.hip (by default a launch of 256 k + r pixel tiles with 0 < r <= CUs / 4 runs its last r tiles on
 * the 64-channel tile of conv3x3_pp128_body.h, in 4 r short workgroups at the end of the same grid, instead of paying a whole round of
 * 256-channel workgroups for r tiles; same bits);
 * bit 14 = one workgroup per tile in conv3x3_pp.hip instead of the persistent grid (by default one workgroup per CU walks up to 16 tiles
 * and requests the next tile's operands inside the current tile's last K block; same bits);
 * bits 15, 16 = unused (16 routed to round 5's conv_igemm_wide.hip, deleted in round 6: BD_EINVAL). */


/* ResNet stem: 7x7/2 pad 3 conv (3->64) + folded FrozenBN + ReLU (models/cls/resnet.py:142-146,238-240).
 * x_halo: bf16 [N][H+6][W+8][4] as written by bd_pad_normalize (zero halo, channel 3 = 0);
 * w_stem: bf16 [64][7][8][4] as written by bd_stem_weight_pack; y: bf16 NHWC [N][H/2][W/2][64]. */
int bd_stem_conv7x7_fwd(int N, int H, int W, const void* x_halo, const void* w_stem, const float* bias,
                        void* y, bd_stream_t stream);
/* The same stem followed by M.MaxPool2d(3, 2, 1) (models/cls/resnet.py:146,238-241) in one pass: y_pool = bf16 NHWC
 * [N][(H/2-1)/2+1][(W/2-1)/2+1][64], bit-identical to bd_stem_conv7x7_fwd + bd_maxpool3x3s2_fwd; the half-resolution
 * stem output is never written (the stem is frozen: solver/default_solver.py:83-94, so nothing else reads it). */
int bd_stem_pool_fwd(int N, int H, int W, const void* x_halo, const void* w_stem, const float* bias, void* y_pool,
                     bd_stream_t stream);
int bd_stem_weight_pack(const float* w /*[64][7][7][3] fp32*/, const float* row_scale, void* w_stem,
                        bd_stream_t stream);

/* fp32 master [Cout][R*S][Cin] (* row_scale[Cout] if non-NULL) -> bf16 fwd layout [Cout][RS][Cin] and
 * bf16 dgrad layout [Cin][RS][Cout] (either output may be NULL). */
int bd_weight_pack(const float* w, const float* row_scale, void* w_fwd, void* w_dgrad, int Cout, int RS,
                   int Cin, bd_stream_t stream);

/* The same packing for many convolutions in one launch (the refresh after every optimizer step).  descs_dev: DEVICE array of
 * n descriptors sorted by block_start, block_start = running sum of bd_weight_pack_blocks(Cout, RS, Cin);
 * total_blocks = the final sum.  w_fwd / w_dgrad / row_scale may be NULL per descriptor. */
typedef struct bd_pack_desc {
    const float* w;
    const float* row_scale;
    void* w_fwd;
    void* w_dgrad;
    int32_t Cout, RS, Cin, block_start;
} bd_pack_desc;
int bd_weight_pack_blocks(int Cout, int RS, int Cin);
int bd_weight_pack_multi(const bd_pack_desc* descs_dev, int n, int total_blocks, bd_stream_t stream);

/* bias gradient: column sums over the pixel rows {n*pix_per_img + off + i : n < N, i < cnt} of a bf16
 * [.][C] activation gradient (one pyramid level, or the whole tensor with N=1, off=0) into fp32 out[C];
 * accumulate != 0 adds.  C % 8 == 0, C <= 2048; two-stage fixed-order reduction through ws (reproducible). */
size_t bd_colsum_workspace_bytes(int C);
int bd_colsum_bf16(const void* g, int N, int64_t pix_per_img, int64_t off, int64_t cnt, int C, float* out,
                   int accumulate, void* ws, size_t ws_bytes, bd_stream_t stream);

/* A FROZEN bottleneck block in one launch (models/cls/resnet.py:70-113 Bottleneck.forward; frozen = stem + layer1 under
 * BACKBONE.FREEZE_AT = 2, solver/default_solver.py:83-94: forward only, nobody reads the mid tensors again):
 *   y = relu( conv3(relu(conv2_3x3(relu(conv1(x) + b1)) + b2)) + b3 + (wd ? conv_d(x) + bd : x) )
 * x: bf16 NHWC [N][H][W][Cin] (one dense level), y: bf16 [N][H][W][Cout]; weights as written by bd_weight_pack (bf16 [Cout][RS][Cin],
 * FrozenBN scale folded in), shifts fp32.  Supported: stride 1, Cmid = 64, Cout = 256, and Cin = 64 with a 1x1 downsample (layer1.0)
 * or Cin = 256 with the identity skip (layer1.1, layer1.2) -- bd_bottleneck_fwd_supported says so; other shapes take the three
 * bd_conv2d_fwd launches.  The mid tensors are rounded to bf16 exactly as those launches store them. */
int bd_bottleneck_fwd_supported(int N, int H, int W, int Cin, int Cmid, int Cout, int has_downsample);
int bd_bottleneck_fwd(int N, int H, int W, int Cin, int Cmid, int Cout, const void* x, const void* w1, const float* b1, const void* w2,
                      const float* b2, const void* w3, const float* b3, const void* wd, const float* bd, void* y, bd_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Memory-bound image ops.
 * ------------------------------------------------------------------------------------------------------- */

/* layers/common/pre_processing.py:11-49 (data_to_input/get_padded_tensor): zero-pad (top-left aligned) to
 * (Hp, Wp), THEN (x-mean)/std, so the pad region holds -mean/std.  in: fp32 NCHW [N][3][H][W].
 * out: bf16 [N][Hp+6][Wp+8][4] with a zero halo of 3 rows / 4 columns and channel 3 = 0 (stem layout). */
int bd_pad_normalize(const float* in, int N, int H, int W, int Hp, int Wp, const float* mean3,
                     const float* std3, void* out, bd_stream_t stream);
/* layers/common/pre_processing.py:13 `Tensor(image)`: the host batch the loader hands over (float64 from utils/dummy.py:60, float32 or
 * uint8 from a reader) becomes the fp32 device tensor bd_pad_normalize reads.  Host-side transport, no arithmetic beyond the exact
 * widening / narrowing cast (float64 -> fp32 rounds to nearest, as numpy's astype does): `threads` workers (0 = a default from the
 * core count) convert `chunk_elems`-element chunks (0 = 1 Mi) into a pinned staging buffer owned by the handle, and every chunk leaves
 * with its own hipMemcpyAsync on `stream` as soon as it is converted -- conversion of chunk k + 1 under the DMA of chunk k.  submit
 * returns when every chunk has been ENQUEUED; dst_dev is valid in stream order.  A second submit first waits until the previous one's
 * copies have drained the staging buffer.  One handle per process / device; not re-entrant for one handle. */
typedef struct bd_h2d* bd_h2d_t;
#define BD_HOST_F64 0
#define BD_HOST_F32 1
#define BD_HOST_U8 2
int bd_h2d_create(bd_h2d_t* out, int device, int threads);
int bd_h2d_threads(bd_h2d_t h);
int bd_h2d_submit(bd_h2d_t h, const void* src_host, int src_dtype, int64_t n, float* dst_dev, int64_t chunk_elems, bd_stream_t stream);
int bd_h2d_destroy(bd_h2d_t h);

/* same arithmetic, fp32 NCHW output [N][3][Hp][Wp] (parity checks of the reference semantics). */
int bd_pad_normalize_nchw(const float* in, int N, int H, int W, int Hp, int Wp, const float* mean3,
                          const float* std3, float* out, bd_stream_t stream);

/* M.MaxPool2d(3, 2, 1) (models/cls/resnet.py:146) on NHWC bf16. */
int bd_maxpool3x3s2_fwd(const void* x, int N, int H, int W, int C, void* y, bd_stream_t stream);

/* FPN top-down merge (fpn_backbone.py:143-148): lateral[n,y,x,c] += bilinear_up2x(top)[n,y,x,c]
 * (align_corners=False, border-clamped). top: [N][H][W][C]; lateral: [N][2H][2W][C] with pixel addressing
 * (n*lat_pix_per_img + lat_off + y*2W + x). */
int bd_upsample2x_add_fwd(const void* top, int64_t top_pix_per_img, int64_t top_off, void* lateral,
                          int64_t lat_pix_per_img, int64_t lat_off, int N, int H, int W, int C,
                          bd_stream_t stream);
/* gradient of the above w.r.t. top: dtop[n,y,x,c] (+)= sum_w dlat * w. accumulate != 0 adds into dtop. */
int bd_upsample2x_add_bwd(const void* dlat, int64_t lat_pix_per_img, int64_t lat_off, void* dtop,
                          int64_t top_pix_per_img, int64_t top_off, int N, int H, int W, int C,
                          int accumulate, bd_stream_t stream);

/* elementwise helpers on bf16 buffers of n elements (n % 8 == 0) */
int bd_relu_bf16(const void* x, void* y, int64_t n, bd_stream_t stream);
/* y = (mask > 0 ? g : 0) [+ add] */
int bd_relu_bwd_bf16(const void* g, const void* mask, const void* add, void* y, int64_t n, bd_stream_t stream);
int bd_add_bf16(const void* a, const void* b, void* y, int64_t n, bd_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Box operators (basedet.layers / basedet.structures).
 * ------------------------------------------------------------------------------------------------------- */

/* DefaultAnchorGenerator (layers/common/anchor_generator.py:111-122): out[(i*W+j)*A+a] =
 * (x,y,x,y)+base[a], x = offset*stride + j*stride.  base: fp32 [A][4] (host-computed in float64, :99-109). */
int bd_anchors_generate(int H, int W, int stride, float offset, const float* base, int A, float* out,
                        bd_stream_t stream);
/* AnchorPointGenerator (anchor_generator.py:152-165): out[(i*W+j)*A+a] = (x, y). */
int bd_points_generate(int H, int W, int stride, float offset, int A, float* out, bd_stream_t stream);

/* structures/op_patch.py:33-97 (IOU), :170-227 (IOA), structures/boxes.py:74-95 (giou), :114-130
 * (intersection).  mode: 0 iou, 1 ioa, 2 intersection, 3 giou.  b1 [m][4], b2 [n][4], out [m][n], fp32. */
int bd_box_pairwise(const float* b1, int m, const float* b2, int n, int mode, float* out, bd_stream_t stream);

/* BoxCoder.encode/decode (structures/boxcoder.py:61-98) with mean/std[4]; rows independent. */
int bd_box_encode(const float* anchors, const float* gt, int64_t n, const float* mean4_host,
                  const float* std4_host, float* out, bd_stream_t stream);
int bd_box_decode(const float* anchors, const float* deltas, int64_t n, const float* mean4_host,
                  const float* std4_host, float* out, bd_stream_t stream);

/* RetinaNet.get_ground_truth (models/det/retinanet.py:211-232) for a whole batch in two launches:
 * IoU (op_patch.py:33-97) -> Matcher (layers/common/matcher.py:31-51, thresholds lo/hi, low-quality rule)
 * -> labels[fg] = class -> BoxCoder.encode of the matched gt for every anchor.
 * gt_boxes [N][Gmax][5] (x1,y1,x2,y2,class 1-based), num_gt [N] int32, anchors [A][4].
 * labels/match_idx [N][A] int32, offsets [N][A][4] fp32, num_fg: int32[1] (sum over the batch, zeroed inside).
 * ws: N*Gmax floats (per-gt max IoU). argmax tie-break: lowest gt index. */
int bd_retina_assign_encode(const float* anchors, int A, const float* gt_boxes, const int32_t* num_gt, int N,
                            int Gmax, float thr_lo, float thr_hi, int allow_low_quality,
                            const float* mean4_host, const float* std4_host, int32_t* labels,
                            int32_t* match_idx, float* offsets, int32_t* num_fg, void* ws, size_t ws_bytes,
                            bd_stream_t stream);

/* FCOS.get_ground_truth (models/det/fcos.py:222-293). points [P][2]; level_of_point via lvl_start[L+1];
 * soi [L][2]; strides [L] (host arrays). labels [N][P] int32, offsets [N][P][4], ctrness [N][P];
 * stats[0] = num_fg, stats[1] = sum of centerness over fg (fp32, zeroed inside). */
int bd_fcos_assign(const float* points, int P, const int32_t* lvl_start_host, const float* soi_host,
                   const int32_t* strides_host, int L, float radius, const float* gt_boxes,
                   const int32_t* num_gt, int N, int Gmax, int32_t* labels, float* offsets, float* ctrness,
                   float* stats, bd_stream_t stream);

/* ATSS.get_ground_truth (models/det/atss.py:17-86): per gt and level the `topk` points closest to the gt centre (ties: lower
 * index), IoU with the point-centred square anchors of side stride*anchor_scale, positives = candidates with IoU >= mean + std
 * (population std, index-order fp32 sums) whose centre lies inside the gt; a point claimed by several gts goes to the highest
 * IoU (ties: lowest gt index).  Outputs as bd_fcos_assign.  ws: bd_atss_assign_workspace_bytes. */
size_t bd_atss_assign_workspace_bytes(int N, int P);
int bd_atss_assign(const float* points, int P, const int32_t* lvl_start_host, const int32_t* strides_host, int L, int topk,
                   float anchor_scale, const float* gt_boxes, const int32_t* num_gt, int N, int Gmax, int32_t* labels,
                   float* offsets, float* ctrness, float* stats, void* ws, size_t ws_bytes, bd_stream_t stream);

/* OTA.get_ground_truth with the top-k matcher (models/det/ota.py:76-181, layers/common/matcher.py:123-161; cfg MATCHING =
 * "topk", the reference default): per gt the dynamic number k = max(1, int(sum of the `candidate_k` largest IoUs with the predicted
 * boxes)) of lowest-cost points, cost = focal classification cost + reg_weight * -log(IoU) + 1e6 outside the gt / its
 * center_radius*stride centre box; a point taken by several gts goes to the gt of lowest cost.  logits bf16 [N*P][K], pred_ltrb
 * bf16 [N*P][4] (the head's decoded l,t,r,b), points fp32 [P][2].  Outputs: labels int32 [N][P] (class, 0 = background), targets
 * fp32 [N][P][4] (ltrb), gt_ious fp32 [N][P]; stats[0] = number of foreground points, stats[1] = 2 * stats[0] (the normaliser
 * of the 0.5-weighted IoU-branch loss).  Ties: lowest point / gt index.  ws: bd_ota_assign_workspace_bytes. */
size_t bd_ota_assign_workspace_bytes(int N, int P);
int bd_ota_assign(const float* points, int P, const int32_t* lvl_start_host, const int32_t* strides_host, int L, const void* logits,
                  int K, const void* pred_ltrb, const float* gt_boxes, const int32_t* num_gt, int N, int Gmax, float alpha,
                  float gamma, float reg_weight, float center_radius, int candidate_k, int32_t* labels, float* targets,
                  float* gt_ious, float* stats, void* ws, size_t ws_bytes, bd_stream_t stream);

/* The same assignment with the Sinkhorn matcher (cfg MATCHING = "sinkhorn"; layers/common/matcher.py:106-121,
 * layers/blocks/sinkhorn_distance.py:22-50): supplies mu_g = max(1, int(sum of the `topq` = 20 largest in-box IoUs)), background
 * supply P - sum mu, demands 1; `iters` = 50 log-domain Sinkhorn updates with regulariser eps = 0.1 on the (G+1) x P cost (last row:
 * the background focal cost); each point goes to the row of largest plan entry after every row is rescaled by its maximum (ties:
 * lowest row).  gt_ious are the IoUs masked by the in-box test (ota.py:155).  Gmax + 1 <= 512.  Outputs as bd_ota_assign. */
size_t bd_ota_sinkhorn_workspace_bytes(int N, int P, int Gmax);
int bd_ota_assign_sinkhorn(const float* points, int P, const int32_t* lvl_start_host, const int32_t* strides_host, int L,
                           const void* logits, int K, const void* pred_ltrb, const float* gt_boxes, const int32_t* num_gt, int N,
                           int Gmax, float alpha, float gamma, float reg_weight, float center_radius, int topq, float eps, int iters,
                           int32_t* labels, float* targets, float* gt_ious, float* stats, void* ws, size_t ws_bytes,
                           bd_stream_t stream);

/* FreeAnchor.get_losses after the network forward (models/det/free_anchor.py:38-142): positive bag loss over the `bucket`
 * anchors of largest IoU per gt (ties at the boundary: lowest anchor index) and negative loss over every (anchor, class) with the
 * box probabilities of the decoded predictions (two gts of one class claiming an anchor: the later gt's value), both with their
 * gradients.  logits bf16 [N*A][K]; offsets bf16 [N*A/anchors_per_pix][box_ld] (anchor a of a pixel at columns 4a..4a+3);
 * anchors fp32 [A][4]; gt [N][Gmax][5] (class 1-based), num_gt int32 [N].  loss_out[0] = alpha * pos / max(1, sum num_gt),
 * loss_out[1] = (1 - alpha) * neg / max(1, sum num_gt * bucket).  d_logits / d_offsets are overwritten (d_offsets zero outside
 * the bags).  Deterministic: fixed-order reductions, bag gradients applied gt after gt.  ws: bd_freeanchor_workspace_bytes. */
size_t bd_freeanchor_workspace_bytes(int N, int Gmax, int bucket, int A);
int bd_freeanchor_loss_fwd_bwd(const void* logits, const void* offsets, int box_ld, int anchors_per_pix, const float* anchors,
                               int A, int K, const float* gt_boxes, const int32_t* num_gt, int N, int Gmax, const float* mean4,
                               const float* std4, float iou_thresh, int bucket, float beta, float reg_weight, float alpha,
                               float gamma, float* loss_out, void* d_logits, void* d_offsets, void* ws, size_t ws_bytes,
                               bd_stream_t stream);

/* layers/common/post_processing.py:17-47 batched_nms (class-offset trick + greedy NMS, suppress iff IoU > thr).
 * boxes [n][4], scores [n], idxs [n] (may be NULL = plain NMS).  keep: int32[n] (descending score order),
 * num_keep: int32[1].  max_output <= 0 means unlimited.  ws from bd_nms_workspace_bytes(n). */
size_t bd_nms_workspace_bytes(int n);
int bd_batched_nms(const float* boxes, const float* scores, const int32_t* idxs, int n, float iou_thresh,
                   int max_output, int32_t* keep, int32_t* num_keep, void* ws, size_t ws_bytes,
                   bd_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Losses: fused forward value + gradient w.r.t. the prediction (bf16 in / bf16 grad out).
 * ------------------------------------------------------------------------------------------------------- */

/* sigmoid_focal_loss (layers/losses/sigmoid_focal_loss.py:9-36) as used at models/det/retinanet.py:148-156:
 * rows with label < 0 ignored, one-hot column = label-1 for label > 0; loss_sum[0] += sum(loss) / max(1,num_fg),
 * dlogits = dloss/dlogit * grad_scale / max(1,num_fg).  norm: device float/int scalar holding num_fg
 * (norm_is_float selects the type).  logits/dlogits bf16 [rows][K]. */
int bd_focal_loss_fwd_bwd(const void* logits, const int32_t* labels, int64_t rows, int K, float alpha,
                          float gamma, const void* norm, int norm_is_float, float grad_scale,
                          float* loss_sum, void* dlogits, bd_stream_t stream);
/* gamma == 2 takes its own instance (one sigmoid / log per logit, the positive class patched in); bd_focal_loss_fwd_bwd_general runs the
 * general-gamma kernel whatever gamma is (the instance is checked against it). */
int bd_focal_loss_fwd_bwd_general(const void* logits, const int32_t* labels, int64_t rows, int K, float alpha,
                                  float gamma, const void* norm, int norm_is_float, float grad_scale,
                                  float* loss_sum, void* dlogits, bd_stream_t stream);

/* smooth_l1_loss (layers/losses/smooth_l1_loss.py:7-34) over rows with label > 0 (retinanet.py:158-162).
 * Row r = pixel*A + a; pred/dpred are bf16 with `ld` channels per pixel (ld >= 4*A, ld % 4 == 0), element
 * (r, k) at pixel*ld + a*4 + k; pad channels of dpred are written as zero.  target fp32 [rows][4]. */
int bd_smooth_l1_fwd_bwd(const void* pred, const float* target, const int32_t* labels, int64_t pixels, int A,
                         int ld, float beta, const void* norm, int norm_is_float, float weight, float* loss_sum,
                         void* dpred, bd_stream_t stream);

/* iou_loss(box_mode="ltrb", loss_type="giou") * ctrness weight (layers/losses/iou_loss.py:9-105,
 * models/det/fcos.py:157-164); norm = max(1, sum_ctr). */
int bd_giou_ltrb_fwd_bwd(const void* pred, const float* target, const float* weight, const int32_t* labels,
                         int64_t rows, const float* norm, float loss_weight, float* loss_sum, void* dpred,
                         bd_stream_t stream);

/* binary_cross_entropy with logits over fg rows (layers/losses/cross_entropy.py:7-29, fcos.py:166-170).
 * Logit of row i at pred[i*ld + off] (the centre-ness channel of the fused bbox/ctrness conv output);
 * dpred is a dense bf16 [rows] vector. */
int bd_bce_logits_fwd_bwd(const void* pred, int ld, int off, const float* target, const int32_t* labels, int64_t rows,
                          const float* norm, float* loss_sum, void* dpred, bd_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * FCOS PointHead pieces (layers/head/point_head.py:47-58, 137-151).
 * ------------------------------------------------------------------------------------------------------- */

/* GroupNorm(32, C=256) (+ReLU) over every (image, pyramid level, group) of a pixel-major multi-level tensor:
 * level l of image n = pixel rows [n*pix_per_img + lvl_off[l], + lvl_cnt[l]).  stats: fp32 [N][L][32][2] = (mean, rstd),
 * kept for the backward.  Two-stage fixed-order reductions over 128-pixel slots (reproducible, independent of the chunking below).
 * ws: bd_groupnorm_workspace_bytes. */
size_t bd_groupnorm_workspace_bytes(int N, int L, int C, int64_t pix_per_img);
int bd_groupnorm_fwd(const void* y, const float* gamma, const float* beta, int N, int L, const int32_t* lvl_off_host,
                     const int32_t* lvl_cnt_host, int64_t pix_per_img, int C, float eps, int relu, float* stats,
                     void* z, void* ws, size_t ws_bytes, bd_stream_t stream);
/* The tower step conv -> GroupNorm(32) -> ReLU (point_head.py:47-58) with the statistics pass FUSED into the convolution (round 6):
 * bd_conv2d_fwd_gnstats = bd_conv2d_fwd of a 3x3 / stride 1 / pad 1 convolution into 256 channels (bias, no residual / ReLU) that also leaves,
 * per 4 x 16-pixel output patch and group of 8 channels, the (sum, sum of squares) of its fp32 results in `part` (bd_conv2d_fwd_gnstats_bytes;
 * the layout is the kernel's own and only bd_groupnorm_fwd_parts reads it); bd_groupnorm_fwd_parts = bd_groupnorm_fwd without its statistics
 * pass: it sums a level's patches in a fixed order into stats (mean, rstd) and applies.  The statistics come from the UNROUNDED results
 * (within 2e-3 of the separate pass, which reads the bf16 tensor); reproducible bit for bit.  d = the convolution's descriptor for both calls. */
size_t bd_conv2d_fwd_gnstats_bytes(const bd_conv_desc* d);
int bd_conv2d_fwd_gnstats(const bd_conv_desc* d, const void* x, const void* w_packed, const float* bias, void* y, float* part,
                          size_t part_bytes, bd_stream_t stream);
int bd_groupnorm_fwd_parts(const bd_conv_desc* d, const void* y, const float* part, const float* gamma, const float* beta, float eps,
                           int relu, float* stats, void* z, bd_stream_t stream);
/* dz = gradient w.r.t. the (ReLU'd) output z; relu != 0 gates it with z > 0, RECOMPUTED from y, stats, gamma and beta with the forward's
 * own arithmetic (round 5: z is no longer read -- two of the backward's seven tensor passes).  dy: gradient w.r.t. the conv output y;
 * dgamma/dbeta fp32 [C] (accumulate != 0 adds). */
int bd_groupnorm_bwd(const void* dz, const void* y, const float* gamma, const float* beta, const float* stats, int N, int L,
                     const int32_t* lvl_off_host, const int32_t* lvl_cnt_host, int64_t pix_per_img, int C, int relu,
                     void* dy, float* dgamma, float* dbeta, int accumulate, void* ws, size_t ws_bytes, bd_stream_t stream);

/* offsets = relu(bbox_pred * scale_l) * stride_l (point_head.py:143).  raw: bf16 [pixels][ld = 8] (channels 0-3 =
 * bbox_pred, 4 = ctrness logit, 5-7 = 0), scales: device fp32 [L]; out: bf16 [pixels][4]. */
int bd_fcos_offsets_fwd(const void* raw, int ld, const float* scales, int N, int L, const int32_t* lvl_off_host,
                        const int32_t* lvl_cnt_host, const int32_t* strides_host, int64_t pix_per_img, void* out,
                        bd_stream_t stream);
/* backward: d_raw[.][0..3] from d_off, d_raw[.][4] = d_ctr (dense bf16 [pixels]), d_raw[.][5..7] = 0; dscale fp32 [L]. */
size_t bd_fcos_offsets_workspace_bytes(void);
int bd_fcos_offsets_bwd(const void* raw, int ld, const float* scales, int N, int L, const int32_t* lvl_off_host,
                        const int32_t* lvl_cnt_host, const int32_t* strides_host, int64_t pix_per_img, const void* d_off,
                        const void* d_ctr, void* d_raw, float* dscale, void* ws, size_t ws_bytes, bd_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Faster R-CNN pieces (models/det/rpn.py, layers/head/rcnn.py, layers/common/roi_pool.py, sampling.py).
 * ------------------------------------------------------------------------------------------------------- */

/* RPN.get_ground_truth (rpn.py:215-227) before the random subsampling: same IoU + Matcher + BoxCoder.encode as
 * bd_retina_assign_encode, but class-agnostic labels in {-1, 0, 1}. */
int bd_rpn_assign_encode(const float* anchors, int A, const float* gt_boxes, const int32_t* num_gt, int N,
                         int Gmax, float thr_lo, float thr_hi, int allow_low_quality,
                         const float* mean4_host, const float* std4_host, int32_t* labels,
                         int32_t* match_idx, float* offsets, int32_t* num_fg, void* ws, size_t ws_bytes,
                         bd_stream_t stream);

/* sample_labels twice as at rpn.py:229-232 (layers/common/sampling.py:7-30), in place on labels [N][A]:
 * at most num_pos_max labels stay 1, then at most num_total - (#kept positives) stay 0; the rest become -1.
 * The reference draws megengine.random.uniform keys and drops the entries with the LARGEST keys (topk with negative
 * k); here the keys are inputs (fp32 in [0,1), [N][A] each) so that the selection is reproducible: the entries with
 * the smallest keys survive, ties broken by the lower index.  num_valid: int32[1] = number of labels >= 0 over the
 * batch (zeroed inside). */
int bd_sample_labels(int32_t* labels, const float* keys_pos, const float* keys_neg, int N, int A, int num_pos_max,
                     int num_total, int32_t* num_valid, bd_stream_t stream);

/* F.topk(descending=True) per (batch item, segment) (rpn.py:155; retinanet.py:188-192 at inference).
 * Item i of segment s of batch b is the element scores[b*batch_stride + (seg_start[s] + i / A)*ldc + coff + i % A]
 * (bf16 or fp32), i in [0, seg_rows[s]*A).  Optionally only items with score > min_score take part.
 * Output order: score descending, then item index ascending.  out_idx/out_score [B][nseg][k] (idx -1 past the count),
 * out_cnt [B][nseg].  k <= 2048. */
int bd_segment_topk(const void* scores, int is_bf16, int B, int64_t batch_stride, int A, int ldc, int coff, int nseg,
                    const int32_t* seg_start_host, const int32_t* seg_rows_host, int k, float min_score,
                    int use_min_score, int32_t* out_idx, float* out_score, int32_t* out_cnt, bd_stream_t stream);

/* batched_nms (post_processing.py:17-47) for B independent problems of capacity C (<= 16384) in three launches.
 * boxes [B][C][4], scores [B][C] (-inf marks an absent item), idxs [B][C] or NULL.  keep [B][keep_ld] (indices into
 * the problem's C items, descending score), num_keep [B]. */
size_t bd_nms_batched_workspace_bytes(int B, int C);
int bd_nms_batched(const float* boxes, const float* scores, const int32_t* idxs, int B, int C, float iou_thresh,
                   int max_output, int keep_ld, int32_t* keep, int32_t* num_keep, void* ws, size_t ws_bytes,
                   bd_stream_t stream);

/* RPN.find_top_rpn_proposals (rpn.py:134-186) for the whole batch: per image and level top pre_k scores ->
 * BoxCoder.decode -> clip to im_info[:, 0:2] -> drop empty boxes -> NMS across levels (level = NMS class) ->
 * first post_k.  raw: bf16 [N*pix_per_img][ldc], objectness logit of cell anchor a at channel cls_off + a, its
 * offsets at box_off + 4a.  anchors [pix_per_img*A][4].  rois [N][post_k][4] (zero past num_rois[n]). */
size_t bd_rpn_proposals_workspace_bytes(int N, int L, const int32_t* lvl_pixels_host, int A, int pre_k, int post_k);
int bd_rpn_proposals(const void* raw, int ldc, int A, int cls_off, int box_off, int N, int64_t pix_per_img, int L,
                     const int32_t* lvl_pix_off_host, const int32_t* lvl_pixels_host, const float* anchors,
                     const float* im_info, int info_ld, const float* mean4_host, const float* std4_host, int pre_k,
                     float nms_thresh, int post_k, float* rois, int32_t* num_rois, void* ws, size_t ws_bytes,
                     bd_stream_t stream);
/* bd_rpn_proposals runs its batched NMS level by level (N x L independent problems of <= pre_k boxes, then a merge into the joint order:
 * round 5); bd_rpn_proposals_joint runs it as one problem per image (rounds 1-4).  The proposals are the same bit for bit: boxes of
 * different levels never overlap after batched_nms's shift (post_processing.py:44-45), which both forms apply. */
int bd_rpn_proposals_joint(const void* raw, int ldc, int A, int cls_off, int box_off, int N, int64_t pix_per_img, int L,
                           const int32_t* lvl_pix_off_host, const int32_t* lvl_pixels_host, const float* anchors,
                           const float* im_info, int info_ld, const float* mean4_host, const float* std4_host, int pre_k,
                           float nms_thresh, int post_k, float* rois, int32_t* num_rois, void* ws, size_t ws_bytes,
                           bd_stream_t stream);

/* RCNN.get_ground_truth (rcnn.py:95-147) per image: candidates = proposals + gt boxes, IoU max/argmax over the gts,
 * fg (>= fg_thresh) / bg ([bg_lo, bg_hi)) masks, random subsampling with caller-supplied keys (see
 * bd_sample_labels; keys_fg/keys_bg [N][key_ld], key_ld >= post_k + Gmax), targets = BoxCoder.encode.
 * Outputs use num_samples fixed slots per image (kept candidates in index order, label -1 marks an empty slot):
 * out_rois [N][num_samples][4], out_labels [N][num_samples], out_targets [N][num_samples][4], out_count [N],
 * total_count int32[1] (zeroed inside). */
int bd_rcnn_sample_targets(const float* rois, const int32_t* num_rois, int post_k, const float* gt_boxes,
                           const int32_t* num_gt, int N, int Gmax, const float* keys_fg, const float* keys_bg,
                           int key_ld, int num_samples, int num_fg_max, float fg_thresh, float bg_thresh_hi,
                           float bg_thresh_lo, const float* mean4_host, const float* std4_host, float* out_rois,
                           int32_t* out_labels, float* out_targets, int32_t* out_count, int32_t* total_count,
                           bd_stream_t stream);

/* roi_pool(..., "roi_align") (roi_pool.py:35-78): level = clamp(floor(4 + log2(sqrt(area)/224))) (:12-25), then
 * RoIAlign (average, sample_points^2 samples per bin, aligned) on that level of a pixel-major bf16 pyramid.
 * RoI r belongs to image r / rois_per_img; labels (optional) < 0 marks an empty slot (output row zeroed).
 * out: bf16 [R][PH*PW][C] (bin-major, channel-minor).  bd_roi_align_bwd is the GENERAL backward (any pooled size, any number of RoIs
 * per image): it scatters into an fp32 gradient pyramid of the same pixel layout with atomic adds (the caller zeroes it; summation
 * order not fixed).  The 7 x 7 training default is bd_roi_align_bwd_bf16 below. */
int bd_roi_align_fwd(const void* feat, int64_t pix_per_img, int C, int L, const int32_t* lvl_pix_off_host,
                     const int32_t* lvl_h_host, const int32_t* lvl_w_host, const int32_t* strides_host,
                     const float* rois, const int32_t* labels, int R, int rois_per_img, int PH, int PW,
                     int sample_points, void* out, bd_stream_t stream);
int bd_roi_align_bwd(const void* gout, int64_t pix_per_img, int C, int L, const int32_t* lvl_pix_off_host,
                     const int32_t* lvl_h_host, const int32_t* lvl_w_host, const int32_t* strides_host,
                     const float* rois, const int32_t* labels, int R, int rois_per_img, int PH, int PW,
                     int sample_points, float* gfeat, bd_stream_t stream);

/* Deterministic backward of bd_roi_align_fwd (the training step's default since round 5; roi_pool.py:35-78 under autograd): the gradient
 * pyramid is cut into 8x8-pixel tiles, every tile gets the list of the RoIs whose samples touch it in slot order (count, scan, fill: no
 * atomics), and one wave per (tile, 128-channel slice) sums dF = A^T (g / S^2) B in registers -- fixed summation order, bitwise
 * reproducible.  gfeat: the bf16 gradient of ALL L_all pyramid levels of the pixel-major buffer.  accumulate == 0: every pixel is written
 * (zeros on levels >= L and where no sample lands); accumulate != 0: the sums are ADDED to what gfeat holds (one bf16 rounding of the
 * total), untouched pixels are left alone.  lvl_* arrays have L_all entries, strides the first L.  PH == PW == 7, C even,
 * rois_per_img <= 512. */
size_t bd_roi_align_bwd_bf16_workspace_bytes(int N, int L_all, const int32_t* lvl_h_host, const int32_t* lvl_w_host,
                                             int rois_per_img);
int bd_roi_align_bwd_bf16(const void* gout, int64_t pix_per_img, int C, int L, int L_all, const int32_t* lvl_pix_off_host,
                          const int32_t* lvl_h_host, const int32_t* lvl_w_host, const int32_t* strides_host,
                          const float* rois, const int32_t* labels, int N, int rois_per_img, int PH, int PW,
                          int sample_points, void* gfeat, int accumulate, void* ws, size_t ws_bytes, bd_stream_t stream);

/* FPNP6 (fpn_backbone.py:172-183): dst[n,y,x,:] = src[n,2y,2x,:]; backward adds gdst into gsrc at the even pixels. */
int bd_subsample2x_fwd(const void* src, int64_t src_pix_per_img, int64_t src_off, int Hs, int Ws, void* dst,
                       int64_t dst_pix_per_img, int64_t dst_off, int C, int N, bd_stream_t stream);
int bd_subsample2x_bwd_add(const void* gdst, int64_t dst_pix_per_img, int64_t dst_off, void* gsrc,
                           int64_t src_pix_per_img, int64_t src_off, int Hs, int Ws, int C, int N, bd_stream_t stream);
int bd_f32_to_bf16(const float* src, void* dst, int64_t n, bd_stream_t stream);
/* dst = bf16(float(dst) + src): the fp32 RoIAlign-backward pyramid joins the gradient already in dst -- the RPN head's dL/dP, whose
 * backward runs under the proposal chain (rpn.py:70-132 / rcnn.py:52-83 are independent up to that sum). n % 8 == 0. */
int bd_f32_to_bf16_add(const float* src, void* dst, int64_t n, bd_stream_t stream);

/* RPN losses (rpn.py:113-131): loss2[0] += mean BCE-with-logits over labels >= 0, loss2[1] += smooth-L1 sum over
 * labels > 0 / max(num_valid, 1); d_raw gets both gradients at the channels of the fused prediction row. */
int bd_rpn_loss_fwd_bwd(const void* raw, int ldc, int A, int cls_off, int box_off, const int32_t* labels,
                        const float* targets, int64_t rows, float beta, const int32_t* num_valid, float* loss2,
                        void* draw, bd_stream_t stream);
/* RCNN losses (rcnn.py:65-83): softmax cross entropy over K+1 classes + smooth-L1 on the gt-class deltas, both
 * divided by num_samples.  raw/draw bf16 [R][ld]: logits at [0, K], deltas at box_off + 4*(class-1). */
int bd_rcnn_loss_fwd_bwd(const void* raw, int ld, int K, int box_off, const int32_t* labels, const float* targets, int R,
                         float beta, const int32_t* num_samples, float* loss2, void* draw, bd_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Inference post-processing (retinanet.py:172-209, fcos.py:181-216, rcnn.py:84-93, post_processing.py:50-103):
 * bd_det_scores -> bd_segment_topk(min_score = TEST.CLS_THRESHOLD) -> bd_det_candidates -> bd_nms_batched ->
 * bd_det_finalize.  One image per call, like the reference.
 * ------------------------------------------------------------------------------------------------------- */

/* scores[r*K + k] = sigmoid(logits[r*K + k]), or sqrt(sigmoid(logit) * sigmoid(ctr[r*ctr_ld + ctr_off])) when ctr != NULL
 * (fcos.py:194).  logits/ctr bf16, scores fp32. */
int bd_det_scores(const void* logits, const void* ctr, int ctr_ld, int ctr_off, int64_t rows, int K, float* scores,
                  bd_stream_t stream);
/* RCNN test branch (rcnn.py:84-93): scores [R][K] = softmax(logits)[:, 1:] (-inf for empty RoI slots),
 * boxes [R][K][4] = BoxCoder.decode(roi, deltas of class k).  raw: bf16 [R][ld] as in bd_rcnn_loss_fwd_bwd. */
int bd_rcnn_predict(const void* raw, int ld, int K, int box_off, const float* rois, const int32_t* num_rois,
                    int rois_per_img, int R, const float* mean4_host, const float* std4_host, float* scores, float* boxes,
                    bd_stream_t stream);
/* Candidate list from the per-level top-k (topk_* as written by bd_segment_topk with B = 1, nseg = L): item index i of
 * level l -> label = i % K, row = lvl_row_off[l] + i / K.  mode 0: box = BoxCoder.decode(anchors[row], offsets of row)
 * with offsets bf16 at (row / A)*off_ld + (row % A)*4; mode 1: anchors are points [rows][2], box = PointCoder.decode;
 * mode 2: box = item_boxes[row*K + label].  Outputs [L*k] (score -inf past the per-level count). */
int bd_det_candidates(int mode, const int32_t* topk_idx, const float* topk_score, const int32_t* topk_cnt, int L, int k,
                      const int32_t* lvl_row_off_host, int K, const float* anchors, const void* offsets, int off_ld, int A,
                      const float* mean4_host, const float* std4_host, const float* item_boxes, float* boxes, float* scores,
                      int32_t* labels, bd_stream_t stream);
/* post_processing.py:93-101: out[j] = candidate keep[j] scaled by (im_info[2]/im_info[0], im_info[3]/im_info[1]) and
 * clipped to (im_info[2], im_info[3]); slots past num_keep[0]: zero box, label -1. */
int bd_det_finalize(const float* boxes, const float* scores, const int32_t* labels, const int32_t* keep,
                    const int32_t* num_keep, int max_out, const float* im_info, float* out_boxes, float* out_scores,
                    int32_t* out_labels, bd_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Optimizer (megengine.optimizer.SGD as configured at solver/default_solver.py:96-114).
 * g' = g*grad_scale + wd*w ; v = momentum*v + g' ; w -= lr*v   (all fp32, n elements)
 * ------------------------------------------------------------------------------------------------------- */
int bd_sgd_momentum_step(float* w, float* v, const float* g, int64_t n, float lr, float momentum, float wd,
                         float grad_scale, bd_stream_t stream);
/* Gradient clipping between the all-reduce and the optimizer step (engine/trainer.py:57-61: `clip_grad(model.parameters(), TYPE,
 * **ARGS)` -> Solver.grad_clip_fn; configs/extra_cfg.py:99-105: TYPE "value" ARGS lower/upper, TYPE "norm" ARGS max_norm/ord), over the
 * flat fp32 gradient arena in place.  pre_scale = the reduce-mode factor (1 / world for MEAN) that has not been applied yet: the
 * reference clips the averaged gradients, so g <- clip(pre_scale * g) and the SGD launch then runs with grad_scale 1.
 *   value: g = min(max(pre_scale * g, lower), upper)
 *   norm : nrm = ||pre_scale * g||_ord over ALL n elements (megengine.optimizer.clip_grad_norm: norm of the per-tensor norms = norm of
 *          the concatenation), g *= pre_scale * min(1, max_norm / (nrm + 1e-6)); ord = INFINITY -> max |g|.  nrm also goes to
 *          norm_out[0] (device, may be NULL).  Fixed-order double accumulation: bitwise reproducible.  No host synchronisation. */
int bd_clip_grad_value(float* g, int64_t n, float pre_scale, float lower, float upper, bd_stream_t stream);
size_t bd_clip_grad_norm_workspace_bytes(void);
int bd_clip_grad_norm(float* g, int64_t n, float pre_scale, float max_norm, float ord, float* norm_out, void* ws, size_t ws_bytes,
                      bd_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * fp8 (OCP e4m3) forward convolutions on the block-scaled MFMA (BASELINE config 5, "fp8 weights").  The reference's
 * mixed-precision hook is fp16 autocast + GradScaler (solver/default_solver.py:66-76, tools/det_train.py:77-78); there is
 * no fp8 counterpart, so the tolerance is stated against the fp32 oracle (tests/test_fp8_gpu.py).
 * ------------------------------------------------------------------------------------------------------- */
/* q[i] = e4m3(clamp(x[i] * scale, +-448)); x bf16, n % 16 == 0. */
int bd_quantize_fp8(const void* x_bf16, int64_t n, float scale, void* q, bd_stream_t stream);
/* wq [Cout][RS][Cin] = e4m3(w * row_scale / s_co), s_co = max |w_co * row_scale| / 448 (one scale per output channel);
 * wscale[co] = s_co / act_scale is the epilogue multiplier of bd_conv2d_fwd_fp8.  w fp32 [Cout][RS][Cin], row_scale = folded FrozenBN
 * scale or NULL. */
int bd_weight_pack_fp8(const float* w, const float* row_scale, int Cout, int RS, int Cin, float act_scale, void* wq, float* wscale,
                       bd_stream_t stream);
/* y (bf16) = epi(conv(xq, wq) * wscale[co] + bias [+ add]); xq = bd_quantize_fp8 of the NHWC input, same geometry rules as
 * bd_conv2d_fwd (any filter / stride 1-2 / multi-level); Cin % 16 == 0, Cout % 8 == 0; flags: BD_EPI_RELU, BD_EPI_ADD_BEFORE. */
int bd_conv2d_fwd_fp8(const bd_conv_desc* d, const void* xq, const void* wq, const float* wscale, const float* bias, const void* add,
                      void* y, int flags, bd_stream_t stream);
/* ... and also writes y8 (may be NULL) = e4m3(clamp(y * q_scale)), the input of a following fp8 convolution (saves its cast pass).
 * 3x3 / stride 1 launches with Cout > 128 take the fp8 instance of the staggered patch kernel (conv3x3_pp8.hip), the rest the
 * generic per-tap kernel (conv_fp8.hip); bd_conv_desc.route[3] = 1 forces the generic kernel (A/B). */
int bd_conv2d_fwd_fp8_ex(const bd_conv_desc* d, const void* xq, const void* wq, const float* wscale, const float* bias, const void* add,
                         void* y, void* y8, float q_scale, int flags, bd_stream_t stream);
/* fp8 DATA GRADIENT of a 3x3 / stride 1 / pad 1 convolution with Cin > 128 (the patch kernel's mirrored-tap mode): the gradient operand
 * g8 is e5m2 ("bf8": the format fp8 training uses for gradients) = bd_quantize_bf8(g, grad_scale) or the dx8 twin of the producing
 * launch; wq_t [Cin][RS][Cout] e4m3 with one scale per INPUT channel and wscale_t[ci] = s_ci / grad_scale from bd_weight_pack_fp8_t.
 * dx (bf16) = epi(conv_transpose(g8, wq_t) * wscale_t [+ add]); add / mask / flags as bd_conv2d_dgrad; dx8 (may be NULL) =
 * e5m2(clamp(dx * q_scale)). */
/* sr_seed != 0: stochastic rounding (a hash of (seed, element index) below the kept bits) instead of round-to-nearest -- the same rule
 * as bd_conv_desc.sr_seed for the dx8 twins of the convolution entry points. */
int bd_quantize_bf8(const void* x_bf16, int64_t n, float scale, void* q, uint32_t sr_seed, bd_stream_t stream);
/* out[0] = max(out[0], max |x|) over a bf16 tensor (n % 8 == 0; NaNs skipped; zero out[0] first): the statistic behind the delayed scaling of
 * the e5m2 gradients (the reference's loss-scale hook, solver/default_solver.py:66-76: GradScaler of the AMP path). */
int bd_absmax_bf16(const void* x_bf16, int64_t n, float* out, bd_stream_t stream);
int bd_weight_pack_fp8_t(const float* w, const float* row_scale, int Cout, int RS, int Cin, float grad_scale, void* wq_t, float* wscale_t,
                         bd_stream_t stream);
int bd_conv2d_dgrad_fp8(const bd_conv_desc* d, const void* g8, const void* wq_t, const float* wscale_t, const void* add, const void* mask,
                        void* dx, void* dx8, float q_scale, int flags, bd_stream_t stream);
/* Backward of a THIN 1x1 prediction layer in one pass over its input (Faster R-CNN's RPN objectness + box deltas: rpn.py:60-68 under
 * autograd; 3 + 12 output channels padded to 16 over every pixel of the pyramid).  x: bf16 [M][Cin] the layer's input, the output of the
 * ReLU in front of it (rpn.py:70-75) -- its non-zero pattern gates dx; g: bf16 [M][Cout] dL/d(prediction); w: the fp32 master weights
 * [Cout][Cin] (rounded to bf16 in the kernel, as bd_weight_pack does).
 *   dx[p][c] = (x[p][c] > 0) * sum_o g[p][o] w[o][c]      dw[o][c] = sum_p g[p][o] x[p][c]      dbias[o] = sum_p g[p][o] (may be NULL)
 * Rows >= cout_real of dw / dbias are written as zeros.  Supported: Cin == 256, Cout == 16.  Fixed summation order (per-workgroup
 * partial sums in ws, added in workgroup order): bitwise reproducible.  Replaces bd_conv2d_wgrad_bias + bd_conv2d_dgrad_ex(EPI_MASK)
 * for such a layer (1.05 -> 0.3 ms at C4's 1.43 M pixels). */
size_t bd_conv1x1_thin_bwd_workspace_bytes(void);
/* Forward of the same layer: y[p][o] = sum_c x[p][c] w[o][c] + bias[o] (bias may be NULL), bf16 [M][Cout]; Cin == 256, Cout == 16. */
int bd_conv1x1_thin_fwd(const void* x, const float* w, const float* bias, int64_t M, int Cin, int Cout, void* y, bd_stream_t stream);
int bd_conv1x1_thin_bwd(const void* x, const void* g, const float* w, int64_t M, int Cin, int Cout, void* dx, float* dw, float* dbias,
                        int cout_real, void* ws, size_t ws_bytes, bd_stream_t stream);

/* fp8 form of the dense 1x1 launches (the ResNet bottleneck's conv1 / conv3, models/cls/resnet.py:70-113; 1x1 / stride 1 / pad 0 over one
 * dense level, K % 128 == 0, produced channels % 32 == 0).  mode 0 = forward: xq = e4m3 twin of the NHWC input (x * act_scale), wq /
 * wscale from bd_weight_pack_fp8 (RS = 1); mode 1 = data gradient: xq = e5m2 twin of the output gradient (g * grad_scale), wq / wscale
 * from bd_weight_pack_fp8_t.  y (bf16) = epi(conv(xq, wq) * wscale[c] + bias [+ add]) with the side inputs / outputs of
 * bd_conv2d_fwd_ex / bd_conv2d_dgrad_ex: mask or maskbits (mode 1), ybits (mode 0), y8 = e4m3 (mode 0) / e5m2 (mode 1) of y * q_scale;
 * each may be NULL.  Two kernels, the same bits: conv1x1_fp8_kernel (conv1x1.hip) and, round 6, the one-byte form of conv1x1_ring_kernel
 * (conv1x1_ring.hip) for launches of up to four 128-channel K steps over many pixels or into >= 256 channels; bd_conv_desc.route[0] applies
 * as for the bf16 launches (3 = the first kernel only, 5 = the ring form wherever legal). */
int bd_conv1x1_fp8(const bd_conv_desc* d, int mode, const void* xq, const void* wq, const float* wscale, const float* bias,
                   const void* add, const void* mask, const uint32_t* maskbits, void* y, uint32_t* ybits, void* y8, float q_scale,
                   int flags, bd_stream_t stream);
/* fp8 WEIGHT GRADIENT of a 3x3 / stride 1 / pad 1 convolution (Cin % 16 == 0, Cout % 16 == 0): x8 = e4m3 twin of the activations
 * (x * act_scale), g8 = e5m2 twin of the output gradient (g * grad_scale), inv_scale = 1 / (act_scale * grad_scale);
 * dw fp32 [Cout][9][Cin] (= or += with accumulate) = row_scale[co] * sum_pix g * x, as bd_conv2d_wgrad (the weight-gradient leg of the
 * detectors' backward through M.Conv2d; the reference's precision hook is solver/default_solver.py:66-76).  The bias gradient is
 * not produced here: bd_colsum_bf16 on the bf16 gradient. */
size_t bd_conv2d_wgrad_fp8_workspace_bytes(const bd_conv_desc* d);
int bd_conv2d_wgrad_fp8(const bd_conv_desc* d, const void* x8, const void* g8, float inv_scale, const float* row_scale, float* dw,
                        int accumulate, void* ws, size_t ws_bytes, bd_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Operator surface of basedet.layers / basedet.structures as stand-alone fp32 entry points (callers written against the
 * reference's Python signatures; the training step uses the fused label-driven kernels above).  Every *_elem function
 * writes loss (value per element, may be NULL) and/or the gradient w.r.t. the prediction times gout (NULL = ones).
 * ------------------------------------------------------------------------------------------------------- */
/* sigmoid_focal_loss(logits, targets, alpha=-1, gamma=0) (layers/losses/sigmoid_focal_loss.py:9-36). */
int bd_sigmoid_focal_loss_elem(const float* logits, const float* targets, int64_t n, float alpha, float gamma, const float* gout,
                               float* loss, float* dlogits, bd_stream_t stream);
/* binary_cross_entropy(pred, label, with_logits=True) (layers/losses/cross_entropy.py:7-29). */
int bd_bce_elem(const float* pred, const float* label, int64_t n, int with_logits, const float* gout, float* loss, float* dpred,
                bd_stream_t stream);
/* smooth_l1_loss(pred, target, beta=1.0) (layers/losses/smooth_l1_loss.py:7-34); beta < 1e-5 = L1. */
int bd_smooth_l1_elem(const float* pred, const float* target, int64_t n, float beta, const float* gout, float* loss, float* dpred,
                      bd_stream_t stream);
/* iou_loss(pred, target, box_mode="ltrb", loss_type, eps) (layers/losses/iou_loss.py:9-56,59-105), row-wise on [n][4] distances.
 * loss_type: 0 "iou" (-log clip(iou, eps)), 1 "linear_iou", 2 "giou", 3 "square_iou".  ious gets iou (giou for type 2). */
int bd_iou_loss_ltrb(const float* pred, const float* target, int64_t n, int loss_type, float eps, const float* gout, float* loss,
                     float* ious, float* dpred, bd_stream_t stream);
/* the loss map of iou_loss applied to an already computed iou / giou array (its xyxy branch works on the PAIRWISE matrix of
 * Boxes.iou / Boxes.giou, iou_loss.py:83-91: bd_box_pairwise mode 0 / 3, then this). */
int bd_iou_to_loss(const float* ious, int64_t n, int loss_type, float eps, float* loss, bd_stream_t stream);
/* Matcher.__call__(matrix) (layers/common/matcher.py:31-51): matrix fp32 [G][A]; per column max / argmax (first maximum), label of
 * the band [thr[k-1], thr[k]) the maximum falls in (thresholds padded with -inf / +inf as the reference does), and with
 * allow_low_quality label 1 for every column that equals some row's maximum.  ws_rowmax: G floats. */
int bd_matcher_matrix(const float* matrix, int G, int64_t A, const float* thresholds_host, const int32_t* labels_host, int n_thresholds,
                      int allow_low_quality, int32_t* match_idx, int32_t* labels, float* ws_rowmax, bd_stream_t stream);
/* assign_rois (layers/common/roi_pool.py:12-25): level index (0-based from min_level) of rois [R][ld], ld 4 (xyxy) or 5 (index + xyxy). */
int bd_assign_roi_levels(const float* rois, int ld, int R, int min_level, int max_level, int32_t* levels, bd_stream_t stream);
/* roi_pool(pooler_type="roi_pool") = F.nn.roi_pooling(mode="max") (roi_pool.py:65): Caffe ROIPooling on one fp32 NCHW level.
 * rois5 [R][5] = (batch index, x1, y1, x2, y2); out fp32 [R][C][PH][PW]. */
int bd_roi_pool_max_fwd(const float* feat_nchw, int N, int C, int H, int W, const float* rois5, int R, float scale, int PH, int PW,
                        float* out, bd_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Data-parallel collectives (RCCL over xGMI; one communicator per process / GPU).  Replaces
 *   dist.bcast_list_(model.parameters() / buffers())      configs/detection_cfg.py:80-82      -> bd_comm_bcast
 *   dist.make_allreduce_cb(reduce_mode) on GradManager     solver/default_solver.py:58-63,121  -> bd_comm_allreduce_async + bd_comm_wait
 *   all_reduce_mean(num_fg), all_reduce_mean(sum_ctr)      models/det/fcos.py:143-144          -> bd_comm_allreduce (caller's stream)
 * RCCL is resolved with dlopen at the first bd_comm_* call (an already mapped copy, else $BD_RCCL_LIB, else
 * /opt/rocm/lib/librccl.so.1).  Rendezvous of the 128-byte id (rank 0 -> all) is the caller's business (a TCP store, MPI, a file).
 * ------------------------------------------------------------------------------------------------------- */
typedef struct bd_comm* bd_comm_t;
#define BD_COMM_ID_BYTES 128
#define BD_COMM_F32 0
#define BD_COMM_BF16 1
#define BD_COMM_I32 2
#define BD_COMM_F64 3
#define BD_COMM_SUM 0
#define BD_COMM_MAX 1
#define BD_COMM_AVG 2   /* sum / world (ncclAvg) */

/* rank 0: fill id128_host (HOST memory, BD_COMM_ID_BYTES) with a fresh ncclUniqueId. */
int bd_comm_unique_id(void* id128_host);
/* collective over all ranks: ncclCommInitRank on `device`; creates the communicator's own high-priority communication stream. */
int bd_comm_init(bd_comm_t* comm, const void* id128_host, int rank, int world, int device);
int bd_comm_rank(bd_comm_t comm);
int bd_comm_world(bd_comm_t comm);
bd_stream_t bd_comm_stream(bd_comm_t comm);
/* in-place broadcast of `count` elements from `root`, ordered on `stream`. */
int bd_comm_bcast(bd_comm_t comm, void* buf, size_t count, int dtype, int root, bd_stream_t stream);
/* in-place all-reduce ordered on the CALLER's stream (small forward-side exchanges). */
int bd_comm_allreduce(bd_comm_t comm, void* buf, size_t count, int dtype, int op, bd_stream_t stream);
/* in-place all-reduce on the communication stream, after everything enqueued so far on the `producers` streams
 * (n_producers <= 8; host array of hipStream_t).  Returns at once; the compute streams are not touched. */
int bd_comm_allreduce_async(bd_comm_t comm, void* buf, size_t count, int dtype, int op, const bd_stream_t* producers_host,
                            int n_producers);
/* `consumer` waits (stream-side) for every collective enqueued so far on the communication stream. */
/* The same with the bucket compressed to bf16 on the wire (SOLVER.ALLREDUCE_DTYPE = "bf16", opt-in; the reference's all-reduce callback
 * keeps the gradient dtype, solver/default_solver.py:58-63): on the communication stream, after the producers' events, buf (fp32) is rounded
 * into tmp_bf16 (count elements, caller-owned), all-reduced as ncclBfloat16, and widened back into buf.  Half the bytes per xGMI link;
 * every rank ends with the same values (the rounded sums), each within bf16 resolution (2^-9 relative) of the fp32 result. */
int bd_comm_allreduce_async_bf16(bd_comm_t comm, float* buf, void* tmp_bf16, size_t count, int op, const bd_stream_t* producers_host,
                                 int n_producers);
int bd_comm_wait(bd_comm_t comm, bd_stream_t consumer);
int bd_comm_destroy(bd_comm_t comm);

#ifdef __cplusplus
}
#endif
#endif /* BASEDET_HIP_H */
