/*
 * fcosdet.h — C-ABI of libfcosdet_hip.so, the MI355X (gfx950) FCOS / HISFCOS hot path.
 *
 * Every entry point replaces a stock ATen / cuDNN / torchvision call the reference makes on its
 * detection hot path (citations are file:line under the reference repo root).  Conventions:
 *   - plain C, no torch types: every pointer is a DEVICE pointer owned by the caller
 *     (tensor.data_ptr()), sizes are int32, the last argument is the hipStream_t to enqueue on
 *     (torch.cuda.current_stream().cuda_stream on ROCm);
 *   - the library never allocates or frees user-visible memory; scratch comes from a caller
 *     workspace whose size the matching fd_*_workspace_bytes() call returns;
 *   - every call only enqueues work (no hidden synchronisation) and returns FD_OK or a negative
 *     error code; the message is available through fd_last_error() (thread local);
 *   - activations are fp32 NHWC ("rows x channels"): row m of a level is pixel
 *     (n, h, w) = (m / (H*W), (m / W) % H, m % W); a tensor may be a channel slice of a wider
 *     buffer, described by (ptr, channel stride cs, channel offset co):  elem = ptr[m*cs + co + c].
 *   - a pyramid (the 5 FCOS levels sharing head weights) is ONE buffer, levels concatenated along
 *     the row axis and described by fd_segs; a plain feature map is a pyramid with one segment.
 */
#ifndef FCOSDET_H_
#define FCOSDET_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FD_OK 0
#define FD_E_INVAL (-1)       /* bad argument (shape, alignment, null pointer)        */
#define FD_E_UNSUPPORTED (-2) /* valid request this build has no kernel for            */
#define FD_E_LAUNCH (-3)      /* hipLaunch / hipGetLastError failure                   */

#define FD_MAX_SEG 8

/* activation ids for conv / depthwise / groupnorm epilogues */
#define FD_ACT_NONE 0
#define FD_ACT_RELU 1
#define FD_ACT_SILU 2
#define FD_ACT_EXP 3 /* exp(v * seg_param[level])  — ScaleExp, reference model/modules/modules.py:170-176 */
#define FD_ACT_SIGMOID 4

typedef void* fd_stream_t; /* hipStream_t */

/* Pyramid-level table.  Rows of level s are [m_start[s], m_start[s+1]); m_start[s+1]-m_start[s] = batch*H[s]*W[s]. */
typedef struct fd_segs {
    int32_t nseg;
    int32_t batch;
    int32_t H[FD_MAX_SEG];
    int32_t W[FD_MAX_SEG];
    int32_t m_start[FD_MAX_SEG + 1];
} fd_segs;

/* ------------------------------------------------------------------------------------------- */
/* library                                                                                       */
int32_t fd_version(void);
const char* fd_last_error(void);

/* ------------------------------------------------------------------------------------------- */
/* Convolution = implicit GEMM on v_mfma_f32_32x32x2_f32 (exact fp32), fused epilogue
 *     y = act( conv(x, w) * scale[c] + shift[c] + res )
 * replaces nn.Conv2d + (frozen) nn.BatchNorm2d + ReLU/SiLU (+ residual add) of
 *   torchvision resnet50 bottlenecks (reference model/backbone/resnet50.py:68-80),
 *   HalfInvertedStageFPN / HisBlock (model/od/HISFcos.py:77-179), HISFCOSHead (HISFcos.py:182-229),
 *   FeaturePyramidNetwork / HeadFCOS (model/od/Fcos.py:61-133).
 * w is packed [Cout][ceil(Cin/32)][KH][KW][32] (K contiguous, 32-channel chunk major, then filter tap; input
 * channels beyond Cin are zero).  Cin % 4 == 0 (a last partial chunk is read as zeros: EfficientNet widths such
 * as 24, 40, 48, 136, 144, 232, 816, 1392), x_cs % 4 == 0, x_co % 4 == 0.
 * in.nseg > 1 requires stride 1 and "same" padding (pad == dil*(K-1)/2).
 * mode FD_CONV_STEM: x is [N][H][W][4] (3 channels + zero pad), 7x7 stride 2 pad 3,
 *   w packed [Cout][7][8][4] (zeros at kw=7 and c=3).
 */
#define FD_CONV_GENERIC 0
#define FD_CONV_STEM 1

/* Arithmetic of the conv kernel.
 *  FD_PREC_F32   : v_mfma_f32_32x32x2_f32, bit-for-bit an fp32 fma chain (default, the parity baseline).
 *  FD_PREC_F16X3 : every fp32 operand is split x = hi + lo*2^-11 (hi, lo f16) and a product is three
 *                  v_mfma_f32_32x32x16_f16 (hi*hi + hi*lo + lo*hi, fp32 accumulation): error ~2^-22 per product,
 *                  operands must stay below f16's range (|x| < 65504).  Activations stay fp32 in HBM (split in the
 *                  loader); w is the pre-split packing [Cout][Cin/32][KH][KW][2][32] f16 (hi plane, lo plane). */
#define FD_PREC_F32 0
#define FD_PREC_F16X3 1
#define FD_PREC_F16 2   /* single-plane f16: operands rounded to f16 once (activations in the loader, weights at pack time: the hi plane of the
                           FD_PREC_F16X3 packing), ONE v_mfma_f32_32x32x16_f16 per product, fp32 accumulation and epilogue, fp32 output -- the
                           arithmetic of a convolution under torch.autocast(float16), which is how the reference trains (train.py:33,175-181).
                           Operands must stay inside f16's range.  Tiles: AUTO, 128x128(_SB), 128x64, 64x128, 64x64, 128x32, 128x96. */

/* block tiles (output pixels x output channels) of the conv kernel */
#define FD_TILE_AUTO 0
#define FD_TILE_128x128 1
#define FD_TILE_128x64 2
#define FD_TILE_64x128 3
#define FD_TILE_64x64 4
#define FD_TILE_128x32 5
#define FD_TILE_128x96 6
#define FD_TILE_128x128_SB 7 /* _SB: single LDS buffer (half the LDS, two barriers per K-tile) */
#define FD_TILE_128x64_SB 8
#define FD_TILE_64x128_SB 9
#define FD_TILE_256x128 10   /* 8 waves (512 threads), 4 x 2 wave tiles of 64 x 64 */
#define FD_TILE_256x128_SB 11
#define FD_TILE_128x96_SB 12
#define FD_TILE_128x128_PATCH 13 /* 3x3 stride-1 'same' convs only: the (128-row tile + halo) input patch is staged ONCE per
                                    32-channel chunk in LDS and the 9 taps are formed from it (9x less L2 -> LDS traffic for
                                    the activations); needs 128 + 2*dil*(W + 1) <= 320 rows, no split-K */
#define FD_TILE_WINOGRAD 14 /* 3x3 stride-1 'same' convs (dilation 1 or 2), fp32: Winograd F(2x2, 3x3) on the fp32 MFMA -- 16 multiplies
                               per (cin, cout) and 2x2 output tile instead of 36 (fd_conv_wino.hip).  `w` must be the
                               fd_wino_pack_weights_f32 packing; Cin % 8 == 0, Cout % 4 == 0, 16-byte addressable y / res / scale /
                               shift; same epilogue contract incl. ksplit (chunk loop split over workgroups, combined in slice
                               order); the result differs from the direct kernel's fma chain by fp32 rounding (~1e-5 relative),
                               deterministically */
#define FD_TILE_WAVE64 15   /* GEMM-addressed layers (1x1, stride 1, no padding; Cin % 32 == 0, fp32): wave-autonomous 64 x 64 tiles, one wave per
                               workgroup, no barrier in the K loop (fd_conv_wave.hip) -- needs fd_conv_params.w_frag (fd_pack_conv_weight_wave_f32) */
#define FD_TILE_WINOGRAD4 16 /* 3x3 stride-1 'same' convs (dilation 1 or 2), fp32: Winograd F(4x4, 3x3) on the fp32 MFMA -- 36 multiplies per 4x4 output tile and
                                channel pair (2.25 per output: 1.78x fewer than F(2x2), 4x fewer than direct); needs the fd_wino4_pack_weights_f32 packing in `w`;
                                ksplit as FD_TILE_WINOGRAD; no gate / gn_stats.  Rounding error ~2x F(2x2)'s, inside the 1e-4 parity bar */
#define FD_TILE_NARROW 17    /* 3x3 stride-1 pad-1 convs with Cout <= 8 (the centre-ness + box-distance predictor, HISFcos.py:207-209), fp32, Cin % 16 == 0, no
                                residual / gn_stats / split-K: exact fp32 FMA chains on the VECTOR unit, one thread per output pixel (fd_conv_narrow.hip).
                                `gate` + `gate_b` (+ gate_act): the preceding GroupNorm's affine + activation applied to the patch on its way to LDS, zero
                                padding AFTER it as in the reference (any number of levels; coefficient rows as documented at gate_b below).
                                `w` = [Cin / 16][3 r][4 quads][3 q][4 k][8 couts] fp32: channel 16 chunk + 4 quad + k, filter tap (r, q); zero filters past Cout */
#define FD_TILE_F16K64 18    /* FD_PREC_F16 only (the AMP training step, train.py:175-181): f16 operands on K-tiles of 64 channels -- twice the matrix work per LDS tile of the
                                single-plane f16 instantiations of the other tiles, and activation maps stored as f16 (io_f16) go to LDS unconverted, 16 bytes per lane.  `w` = the
                                fd_pack_conv_weight_f32 mode | 16 packing; Cin % 64 == 0, x_cs / x_co multiples of 8, 4-channel aligned output / residual views; ReLU / SiLU / none;
                                any stride / dilation / pyramid / scatter; no split-K, gate, gn_stats, x2.  The library picks the block tile. */
#define FD_TILE_COUNT 17

typedef struct fd_conv_params {
    const float* x;
    const float* w;
    const float* scale; /* [Cout] or NULL (=1) */
    const float* shift; /* [Cout] or NULL (=0) */
    const float* res;   /* optional residual in output geometry, or NULL */
    float* y;
    int32_t x_cs, x_co;
    int32_t res_cs, res_co;
    int32_t y_cs, y_co;
    int32_t Cin, Cout, KH, KW, stride, pad, dil;
    int32_t act;    /* FD_ACT_* applied to output channels >= act_c0; channels below get identity */
    int32_t act_c0;
    int32_t mode;   /* FD_CONV_GENERIC | FD_CONV_STEM */
    int32_t tile;   /* 0 = built-in heuristic; FD_TILE_* forces a block tile (plan-time autotuning) */
    int32_t tag;    /* 1 = launch under a separate kernel symbol (<..., TAG=1>) so a profiler can isolate this layer */
    int32_t ksplit; /* <= 1: off.  > 1: split the K loop over `ksplit` workgroups per tile (finer work units for maps
                       with fewer tiles than CU slots); partial sums go to `workspace`, a second launch combines them
                       in slice order (deterministic) and applies the epilogue */
    void* workspace;           /* split-K scratch, fd_conv_workspace_bytes(out_rows, Cout, ksplit) bytes, 16-B aligned */
    int64_t workspace_bytes;
    int32_t precision; /* FD_PREC_F32 (exact fp32 MFMA) | FD_PREC_F16X3 (opt-in split-f16 products, see below) */
    int32_t res_mode;  /* what `res` does: 0 = added (residual connection); 1 = ReLU mask: y = res > 0 ? v : 0 -- the data
                          gradient of a layer whose input came out of a ReLU is masked in the epilogue that produces it
                          (train step: removes the separate threshold pass); 2 = `res` is a HALF-resolution map [N][H / 2][W / 2] added
                          AFTER the activation, y[n, i, j] = act(conv * scale + shift) + res[n, i / 2, j / 2]: an FPN lateral with the x2 nearest-
                          neighbour upsampling of the coarser level and the add folded into its epilogue (HISFcos.py:155-165, Fcos.py:77-91:
                          no upsample-add pass over the finer map).  fp32 1x1 stride-1 unpadded single-level convs with even H, W, Cin % 32 == 0,
                          16-byte views, no split-K / scatter / gate / gn_stats / x2; tiles AUTO, 64x64, 128x64_SB, 64x128_SB */
    float seg_param[FD_MAX_SEG]; /* per-level scalar for FD_ACT_EXP */
    fd_segs in;     /* input geometry */
    /* Data gradient of a STRIDED conv as parity classes (train.py:175-181; the 3x3 s2 / 1x1 s2 layers of the ResNet trunk):
     * dX pixels with (h % s, w % s) = (a, b) only see the filter taps r = (a + pad) % s + s*t, so each class is a stride-1
     * conv over dY with those taps (KH x KW may differ) whose outputs interleave into dX.  Single-level input only.
     *   out_H, out_W > 0: explicit output size (instead of the size derived from pad / K): taps past the input read zero;
     *   sc_H, sc_W > 0:   output pixel (n, i, j) is written to pixel (sc_sy*i + sc_oy, sc_sx*j + sc_ox) of an
     *                     [N][sc_H][sc_W] map at y (and `res` is read there); y_cs / y_co / res_* describe that map. */
    int32_t out_H, out_W;
    int32_t sc_sy, sc_sx, sc_oy, sc_ox, sc_H, sc_W;
    /* Per-(image, input channel) gate applied to the INPUT in the loader: y = act(conv(x * gate[n][c], w) * scale + shift + res).
     * The squeeze-excitation gate of an MBConv block (efficientnet_pytorch MBConvBlock: x * sigmoid(se_expand(...)), then _project_conv)
     * folded into the project conv: the scaling pass over the expanded map (a read and a write of it) disappears.  1x1 stride-1
     * unpadded convs on a single level, FD_PREC_F32, no split-K; gate is [batch][gate_cs] floats, gate_cs >= Cin, 16-byte aligned rows.
     * The library picks the block tile (tile / tag are ignored).  NULL = off. */
    const float* gate;
    int32_t gate_cs;
    int32_t reserved0;
    /* FD_TILE_WAVE64 only: the SAME weights as `w` in MFMA fragment order (fd_pack_conv_weight_wave_f32); NULL: that tile is unavailable.
     * A caller that holds both packings can switch tiles per launch (plan-time autotuning). */
    const float* w_frag;
    /* GroupNorm fused around the conv (HISFCOSHead, HISFcos.py:216-225: pw1 -> GN -> ReLU -> dw1 -> GN -> SiLU -> pw2; towers -> GN -> ReLU):
     *   gn_stats != NULL: the epilogue also writes, per output row m and channel group g of `gn_groups` groups over Cout, the fp32 pair
     *     (sum, sum of squares) of the group's STORED values to gn_stats[m][g] (float2; rows x gn_groups x 8 bytes) in a fixed summation order;
     *     fd_groupnorm_from_rowstats turns them into the per-(level, image, group) statistics -- the statistics pass over the map disappears.
     *     Needs Cout % 32 == 0, (Cout / gn_groups) in {4, 8, 16, 32}, 16-byte output views, no split-K, no output scatter.
     *   gate_b != NULL (with `gate`): the loader applies x' = act(x * gate[img][c] + gate_b[img][c]), gate_act in {NONE, RELU, SILU}, img =
     *     level * batch + image -- the GroupNorm affine + activation of the PRECEDING layer applied on the way to LDS (coefficients from
     *     fd_groupnorm_from_rowstats), so the normalise pass over the map disappears too.  `gate` alone keeps meaning x * gate (MBConv SE).
     *     Consumers: the 1x1 GEMM layers (above) and FD_TILE_NARROW (3x3, padded: the padding stays zero). */
    float* gn_stats;
    int32_t gn_groups;
    int32_t gate_act;
    const float* gate_b;
    /* A SECOND input whose channels continue the reduction (K-concatenation; 1x1 unpadded fp32 layers on a single level, no split-K / gate):
     *     y = act((x . W[:, :Cin]^T  +  x2' . W[:, Cin:]^T) * scale + shift (+ res)),   x2'[n, i, j] = x2[n, x2_stride * i, x2_stride * j]
     * with `w` the [Cout][Cin + x2_Cin] bank in the plain packing.  A ResNet bottleneck that changes resolution / width ends in
     * relu(bn3(conv3(o2)) + bn_d(downsample(x))) (torchvision Bottleneck.forward behind model/backbone/resnet50.py:68-80): with the BatchNorm scales
     * folded into the two filter banks this is ONE GEMM over K = planes + inplanes, and the 4 * planes wide identity map (420 MB per 16 images in
     * layer1) is neither written nor read back.  x2 is an [batch][x2_H][x2_W] NHWC map; Cin % 32 == 0, x2_Cin % 32 == 0.  NULL = off. */
    const float* x2;
    int32_t x2_cs, x2_co, x2_Cin, x2_stride, x2_H, x2_W;
    /* FD_TILE_WINOGRAD4 only: launch the workgroups [wg_first, wg_first + wg_count) of the layer's grid (fd_conv_workgroups(p) in all; wg_first % 8 == 0;
     * no split-K) instead of all of them -- the layer as several launches whose union is the layer.  One workgroup of that kernel owns a CU, so a grid that
     * is no multiple of the CU count ends in a round with most of the chip idle (the HISFCOS head tower at 16 x 640 x 640: 2 152 workgroups = 8.4 rounds
     * on 256 CUs); launched as whole rounds + a tail, the caller can schedule other work beside the tail (pipeline.TwoLanePipeline).  wg_count = 0: all. */
    int32_t wg_first, wg_count;
    /* FD_TILE_WINOGRAD4 only: the PERSISTENT stream-K form of the launch.  sk_wgs > 0 (a multiple of 8, at most the CU count: 256 on MI355X): that many workgroups, each
     * walks a contiguous, equally long range of the layer's (tile block, 8-channel chunk) units; an item cut by a range boundary is finished by the workgroup that holds its
     * first chunks, which adds the other parts' partial outputs (workspace slots, signalled through flags) in a fixed order -- deterministic, and bit-identical to the plain
     * launch for every item that is not cut.  The layer then takes total-work / CUs instead of whole rounds of workgroups (HISFCOS cls_logits, HISFcos.py:207: 538 items on
     * 256 CUs = 2.1 rounds that ran as 3), and a CU's items follow each other without a dispatch in between.  `workspace` = fd_conv_sk_workspace_bytes(sk_wgs) bytes,
     * 256-byte aligned, its first 8192 bytes (slot flags + the XCDs' queue heads: a fixed header, so launches with different sk_wgs may share a workspace that is large
     * enough for the largest) ZERO before the first launch (every launch leaves them zero); one workspace per concurrently running launch.  No split-K,
     * no wg_first / wg_count.  0: off. */
    int32_t sk_wgs;
    /* FD_PREC_F16 only: AMP activations stored as f16 in HBM (train.py:33,175-181 trains under torch.autocast: its convolutions read and write fp16 tensors).
     * Bit 0: `x` holds _Float16 elements, bit 1: `y`, bit 2: `res` (added or used as ReLU mask).  Channel strides / offsets stay in ELEMENTS; the loader fetches a
     * lane's four channels as 8 bytes and stores them to LDS unconverted, the epilogue rounds the fp32 accumulator result once (round to nearest even).  Plain
     * implicit-GEMM tiles (incl. split-K and the output scatter of strided data gradients); no gate / gn_stats / x2 / Winograd / narrow / wave / patch tiles.  0: fp32 maps. */
    int32_t io_f16;
} fd_conv_params;
/* Workspace bytes of the persistent stream-K form (flags + one 128 KB slot per workgroup); -1: sk_wgs not a multiple of 8 in 8 .. 1024. */
int64_t fd_conv_sk_workspace_bytes(int32_t sk_wgs);

int32_t fd_conv2d_nhwc_f32(const fd_conv_params* p, fd_stream_t stream);
/* Workgroups (grid.x) of the launch `p` describes -- FD_TILE_WINOGRAD4 only (FD_E_UNSUPPORTED otherwise): the range wg_first / wg_count index. */
int32_t fd_conv_workgroups(const fd_conv_params* p);
/* ... and how many workgroups of the slice `p` names (wg_count = 0: of the whole grid) own a tile -- the grid is padded to 8 XCD shares of equal size, the
 * padding workgroups exit at once: the work a slice does is live(slice) / live(all) of the layer's. */
int32_t fd_conv_workgroups_live(const fd_conv_params* p);
/* Two GEMM-addressed (1x1, stride 1, unpadded) layers back to back in ONE launch -- a ResNet bottleneck's conv3 + BN + residual + ReLU and the next
 * block's conv1 + BN + ReLU (torchvision Bottleneck.forward behind model/backbone/resnet50.py:68-80):
 *     y = act1(x . W1^T * scale1 + shift1 + res)   [rows][N1]   written to HBM (it is the next block's residual)
 *     z = act2(y . W2^T * scale2 + shift2)         [rows][N2]   computed from the rows of y while they are still on chip: y is not read back
 * w1_frag / w2_frag: the [N1][K1] / [N2][N1] filter banks in MFMA fragment order (fd_pack_conv_weight_wave_f32).  K1 % 32 == 0, N1 % 64 == 0,
 * N2 in {64, 128}; activations NONE / RELU / SILU; all views 16-byte addressable.  y and z are bit-identical to two fd_conv2d_nhwc_f32 launches. */
typedef struct fd_b2b_params {
    const float* x; const float* w1_frag; const float* scale1; const float* shift1; const float* res; float* y;
    const float* w2_frag; const float* scale2; const float* shift2; float* z;
    int32_t x_cs, x_co, res_cs, res_co, y_cs, y_co, z_cs, z_co;
    int32_t K1, N1, N2, act1, act2, reserved0;
    int64_t rows;
} fd_b2b_params;
int32_t fd_conv1x1_b2b_f32(const fd_b2b_params* p, fd_stream_t stream);
/* Accumulators per output pixel the FD_TILE_NARROW kernel instantiation of a Cout-channel layer carries (4, 5 or 8); -1: Cout not in 1..8. */
int32_t fd_conv_narrow_nco(int32_t Cout);
int64_t fd_conv_workspace_bytes(int64_t out_rows, int32_t Cout, int32_t ksplit);

/* Weights for FD_TILE_WINOGRAD4: OIHW fp32 [Cout][Cin][3][3] -> U = G g G^T (6x6 per filter, computed in double, rounded once) packed
 * [ceil(Cout/32)][Cin/8][36 frequencies][32 cout][8 cin] (fd_wino4_weight_bytes(Cout, Cin) bytes, zero rows past Cout). */
int64_t fd_wino4_weight_bytes(int32_t N, int32_t K);
/* mode 0: forward weights (N = Cout, K = Cin).  mode 1: data-gradient weights (N = Cin, K = Cout: taps flipped, channel roles swapped,
 * times scale[Cout] when given) -- as fd_wino_pack_weights_f32. */
int32_t fd_wino4_pack_weights_f32(const float* w, const float* scale, float* out, int32_t Cout, int32_t Cin, int32_t mode, fd_stream_t stream);

/* Weights for FD_TILE_WAVE64: [Cout][Cin] fp32 (an OIHW 1x1 filter bank, Cin % 32 == 0) -> MFMA fragment order
 * [ceil(Cout/64)][Cin/32][2 sub-tiles][4 k-steps][64 lanes][4 floats] (fd_conv_weight_wave_bytes bytes, zero rows past Cout): lane
 * (l & 31, l >> 5) of sub-tile j, k-step s, K-tile kt holds w[64 nt + 32 j + (l & 31)][32 kt + 8 s + 4 (l >> 5) + 0..3]. */
int64_t fd_conv_weight_wave_bytes(int32_t Cout, int32_t Cin);
int32_t fd_pack_conv_weight_wave_f32(const float* w, float* out, int32_t Cout, int32_t Cin, fd_stream_t stream);

/* Weights for FD_TILE_WINOGRAD: OIHW fp32 [Cout][Cin][3][3] -> U = G g G^T (computed in double, rounded once) packed
 * [ceil(Cout/32)][Cin/8][16 frequencies][32 cout][8 cin] (fd_wino_weight_bytes(Cout, Cin) bytes, zero rows past Cout).
 * mode 0: forward weights.  mode 1: weights of the stride-1 data-gradient conv (output channels = Cin, reduction over Cout,
 * Cout % 8 == 0): g'[ci][co][r][q] = g[co][ci][2-r][2-q] * (scale ? scale[co] : 1); buffer of fd_wino_weight_bytes(Cin, Cout). */
int64_t fd_wino_weight_bytes(int32_t Cout, int32_t Cin);
int32_t fd_wino_pack_weights_f32(const float* w, const float* scale, float* out, int32_t Cout, int32_t Cin, int32_t mode,
                                 fd_stream_t stream);

/* Convolution backward for the train step (reference train.py:175-181: scaler.scale(loss).backward() runs torch's
 * convolution_backward through cuDNN / MIOpen).
 *  - weight gradient: fd_conv2d_bwd_weight_f32, a pixel-reduction GEMM on the fp32 MFMA:
 *        dw[co][r][q][ci] = sum_m dy[m][co] * x[pix(m, r, q)][ci]          (dw is [Cout][KH][KW][Cin], OHWI)
 *    any stride / padding / dilation, pyramids included; Cin % 4 == 0, Cout % 4 == 0.
 *  - data gradient of a stride-1 layer: fd_conv2d_nhwc_f32 itself on dy with the weights flipped and transposed
 *    (w'[ci][co][r][q] = w[co][ci][KH-1-r][KW-1-q], pad' = dil*(K-1) - pad); needs Cout % 32 == 0.
 */
typedef struct fd_conv_wgrad_params {
    const float* x;  /* forward input rows  */
    const float* dy; /* gradient w.r.t. the forward output rows */
    float* dw;       /* [Cout][KH][KW][Cin] */
    int32_t x_cs, x_co, dy_cs, dy_co;
    int32_t Cin, Cout, KH, KW, stride, pad, dil;
    void* workspace; /* fd_conv_wgrad_workspace_bytes(out_rows, Cin, Cout, KH, KW) bytes, 16-B aligned */
    int64_t workspace_bytes;
    int32_t nsplit;  /* 0 = library chooses; else the pixel-range split count (workspace >= (nsplit + 8) * |dw| * 4 bytes:
                        ranges never cross a pyramid level, so every level adds at most one) */
    int32_t layout;  /* dw layout: 0 = [Cout][KH][KW][Cin] (OHWI), 1 = [Cout][Cin][KH][KW] (OIHW, torch's parameter layout) */
    const float* scale; /* optional [Cout]: dw[co] *= scale[co] (a frozen BatchNorm folded into the forward epilogue) */
    fd_segs in;      /* forward INPUT geometry */
    int32_t precision; /* FD_PREC_F32 (exact, default) | FD_PREC_F16: x and dy rounded to f16 on their way to LDS, v_mfma_f32_32x32x16_f16, fp32
                          accumulation -- the weight gradient of a convolution under torch.autocast(float16) (train.py:175-181) */
    int32_t io_f16;    /* FD_PREC_F16 only: bit 0: `x` holds _Float16 elements, bit 1: `dy` (AMP activations / gradients stored as f16: fetched as 8 bytes
                          per lane and staged without conversion; strides / offsets stay in elements).  0: fp32 maps */
} fd_conv_wgrad_params;

int64_t fd_conv_wgrad_workspace_bytes(int64_t out_rows, int32_t Cin, int32_t Cout, int32_t KH, int32_t KW);
int32_t fd_conv2d_bwd_weight_f32(const fd_conv_wgrad_params* p, fd_stream_t stream);
/* OIHW fp32 weights -> the [N][K/32][KH][KW][32] layout fd_conv2d_nhwc_f32 reads, in one pass (the per-step weight
 * preparation of the train step; plans pack once at build time).  mode 0: forward weights (N = Cout, K = Cin, Cin % 32
 * == 0).  mode 1: weights of the stride-1 data-gradient conv (N = Cin, K = Cout, Cout % 32 == 0):
 * w'[ci][co][r][q] = w[co][ci][KH-1-r][KW-1-q] * (scale ? scale[co] : 1).
 * mode | 4: the same weights in the FD_PREC_F16X3 / FD_PREC_F16 operand format [N][K/32][KH][KW][2][32] f16 (hi = f16(w) round-to-nearest,
 * lo = f16((w - hi) * 2^11)); the output buffer has the same byte size as the fp32 packing.
 * mode | 16 (not with | 4): FD_TILE_F16K64's operand -- f16 [N][K/64][KH][KW][64], K % 64 == 0; HALF the byte size of the fp32 packing. */
int32_t fd_pack_conv_weight_f32(const float* w, const float* scale, float* out, int32_t Cout, int32_t Cin, int32_t KH,
                                int32_t KW, int32_t mode, fd_stream_t stream);

/* The same for many weight tensors in ONE launch (a train step re-packs every layer after the optimizer update).
 * `jobs_dev` is a DEVICE array of n_jobs descriptors (pointers are device pointers; the caller checks the divisibility
 * rules of fd_pack_conv_weight_f32); max_elems = the largest Cout*Cin*KH*KW among them (sizes the grid). */
typedef struct fd_pack_job {
    const float* w;      /* [Cout][Cin][KH][KW] */
    const float* scale;  /* [Cout] or NULL (mode 1 only) */
    float* out;
    int32_t Cout, Cin, KH, KW, mode, reserved; /* mode: 0 / 1 (| 4 or | 16) as fd_pack_conv_weight_f32; 2 / 3 = the Winograd packing of fd_wino_pack_weights_f32 (mode 0 / 1), 8 / 9 = that of fd_wino4_pack_weights_f32; 3x3 only */
} fd_pack_job;
int32_t fd_pack_conv_weights_batch_f32(const fd_pack_job* jobs_dev, int32_t n_jobs, int64_t max_elems, fd_stream_t stream);

/* The ResNet stem as its own kernel: y = act(conv7x7_s2_p3(x) * scale + shift), 3 -> 64 channels, on the [N][H][W][4] image layout
 * (torchvision resnet50 conv1 + bn1 + relu behind the reference's model/backbone/resnet50.py:68-80).  The workgroup's input patch and
 * the filter bank are staged once in LDS, K = 7 x 22 (5 % padding; FD_CONV_STEM of fd_conv2d_nhwc_f32 pads 147 to 224).
 * w packed [7 filter rows][22][64]: k = 3 * (filter column) + (input channel), k = 21 zero.  Output [N][H/2][W/2] rows, 64 channels. */
int32_t fd_stem7x7_nhwc4(const float* x4, const float* w, const float* scale, const float* shift, float* y, int32_t y_cs, int32_t y_co,
                         int32_t N, int32_t H, int32_t W, int32_t act, fd_stream_t stream);

/* The stem WITH the max-pool that follows it (torchvision resnet50: conv1 + bn1 + relu + maxpool(3, stride 2, padding 1)) in one launch:
 * y_pooled [N][Hp][Wp] rows, 64 channels, Hp = (H/2 - 1) / 2 + 1.  The 64-channel stride-2 map is never written.  Windows that straddle two workgroups
 * are combined by integer atomicMax on the bits of the (non-negative, post-ReLU) values, starting from zeros that a first small launch of the same call
 * writes to exactly those pixels (two launches on `stream`; nothing outside y_pooled's 64-channel view is touched).  Bitwise reproducible. */
int32_t fd_stem7x7_pool_nhwc4(const float* x4, const float* w, const float* scale, const float* shift, float* y_pooled, int32_t y_cs, int32_t y_co,
                              int32_t N, int32_t H, int32_t W, fd_stream_t stream);

/* Both stem kernels reading the reference's own input tensor -- fp32 [N][3][H][W] (NCHW, dataset/voc.py:141-173), 4-byte aligned -- in their patch
 * loaders: no fd_nchw3_to_nhwc4 pass.  pool != 0: conv1 + bn1 + relu + maxpool as fd_stem7x7_pool_nhwc4 (y = the pooled map, act ignored); pool == 0:
 * as fd_stem7x7_nhwc4.  Same arithmetic and summation order: bit-identical to the nhwc4 entry points on the converted input. */
int32_t fd_stem7x7_nchw3(const float* x, const float* w, const float* scale, const float* shift, float* y, int32_t y_cs, int32_t y_co,
                         int32_t N, int32_t H, int32_t W, int32_t act, int32_t pool, fd_stream_t stream);

/* [N][3][H][W] fp32 (NCHW, the reference's input layout, dataset/voc.py:141-173) -> [N][H][W][4] (c=3 zero) */
int32_t fd_nchw3_to_nhwc4(const float* x, float* y, int32_t N, int32_t H, int32_t W, fd_stream_t stream);
/* Input pipeline tail on the device: uint8 [N][H][W][3] images, already resized and zero padded on
 * the host (cv2.resize is third-party arithmetic and stays there, dataset/voc.py:110-139) -> normalised fp32
 * [N][H][W][4] = ((u8/255) - mean) / std per channel (transforms.ToTensor + Normalize, voc.py:57-58,104,155),
 * channel 3 = 0: the stem conv's input, no NCHW detour.  mean3 / std3 are HOST pointers to 3 floats. */
int32_t fd_preprocess_u8_nhwc4(const uint8_t* x, float* y, int32_t N, int32_t H, int32_t W, const float* mean3,
                               const float* std3, fd_stream_t stream);
/* Output pipeline tail (Test_coco.py:147-151): boxes /= scale, then xyxy -> xywh (COCO), in place on [n][4]. */
int32_t fd_boxes_rescale_xywh(float* boxes, int64_t n_boxes, float scale, fd_stream_t stream);
/* NHWC rows -> NCHW copy (only for callers that insist on contiguous NCHW) */
int32_t fd_nhwc_to_nchw(const float* x, int32_t x_cs, int32_t x_co, float* y, int32_t N, int32_t HW,
                        int32_t C, fd_stream_t stream);

/* ------------------------------------------------------------------------------------------- */
/* Bandwidth-bound layer ops                                                                     */

/* nn.MaxPool2d(k, s, pad) on NHWC; k=3,s=2,pad=1 is the ResNet stem pool, k=2,s=2,pad=0 the FPN
 * down_sample{1..6} (HISFcos.py:131-136); optional fused "+ add" of a tensor in output geometry
 * (torch.add(p5_2, p4), HISFcos.py:168-177). */
int32_t fd_maxpool_nhwc(const float* x, int32_t x_cs, int32_t x_co, float* y, int32_t y_cs, int32_t y_co,
                        const float* add, int32_t add_cs, int32_t add_co, int32_t N, int32_t H, int32_t W,
                        int32_t C, int32_t k, int32_t s, int32_t pad, fd_stream_t stream);

/* y = nearest_upsample_x2(x) + lat      (nn.Upsample(scale_factor=2) + torch.add, HISFcos.py:155-165) */
int32_t fd_upsample2x_add_nhwc(const float* x, int32_t x_cs, int32_t x_co, const float* lat, int32_t lat_cs,
                               int32_t lat_co, float* y, int32_t y_cs, int32_t y_co, int32_t N, int32_t H,
                               int32_t W, int32_t C, fd_stream_t stream);

/* Depthwise 3x3, stride 1, pad 1 (DepthWiseConv2d, modules.py:40-49): y = act(dw(x)*scale + shift).
 * w packed [9][C]. */
int32_t fd_dwconv3x3_nhwc(const float* x, int32_t x_cs, int32_t x_co, const float* w, const float* scale,
                          const float* shift, float* y, int32_t y_cs, int32_t y_co, int32_t C, int32_t act,
                          const fd_segs* segs, fd_stream_t stream);

/* Weight gradient of the depthwise 3x3 conv above (stride 1, pad 1): dw[t][c] = sum_m x[pix(m, t)][c] * dy[m][c],
 * t = 3*r + q, same [9][C] layout as `w` (layout 0) or torch's [C][1][3][3] (layout 1); optional scale[C] multiplies
 * the result per channel (a frozen BatchNorm folded into the forward).  Replaces the depthwise half of torch's convolution_backward in the
 * reference's train step (train.py:175-181, modules.py:40-49).  The data gradient is fd_dwconv3x3_nhwc itself with the
 * taps reversed (w'[t] = w[8-t]).  Deterministic: fixed row partition, fixed-order fp64 final sum.
 * workspace: fd_dwconv3x3_wgrad_workspace_bytes(segs, C).  C/4 must divide 256 or be a multiple of 256. */
int64_t fd_dwconv3x3_wgrad_workspace_bytes(const fd_segs* segs, int32_t C);
int32_t fd_dwconv3x3_bwd_weight_nhwc(const float* x, int32_t x_cs, int32_t x_co, const float* dy, int32_t dy_cs,
                                     int32_t dy_co, float* dw, int32_t C, const float* scale, int32_t layout,
                                     const fd_segs* segs, void* workspace, fd_stream_t stream);

/* Dilated depthwise k x k convolution, stride 1, 'same' zero padding (pad = dil * (k - 1) / 2), y = act(dw(x) * scale + shift), over any
 * pyramid: MNBlock.DilatedDepthWiseConv + BN of the reference's model/modules/modules.py:195-216 (MNFCOS: model/od/MNFcos.py:222-297;
 * as shipped MNBlock pads with `dilation`, which keeps the size only for k = 3 -- its residual add raises for k = 5 / 7 -- the
 * 'same' padding is the repaired behaviour).  k in {3, 5, 7}; w packed [K*K][C] (tap major); C % 4 == 0. */
int32_t fd_dwconv_dilated_nhwc(const float* x, int32_t x_cs, int32_t x_co, const float* w, const float* scale, const float* shift,
                               float* y, int32_t y_cs, int32_t y_co, int32_t C, int32_t K, int32_t dil, int32_t act,
                               const fd_segs* segs, fd_stream_t stream);

/* Depthwise k x k convolution with stride and asymmetric zero padding, y = act(dw(x)*scale + shift): the depthwise
 * stage of an EfficientNet MBConv block (efficientnet_pytorch 0.7.1 MBConvBlock._depthwise_conv + _bn1 + swish, wrapped by
 * the reference's model/backbone/efficientnetv1.py:11-26; Conv2dStaticSamePadding pads (pad//2, pad - pad//2), i.e. more at
 * the bottom / right).  k in {3, 5, 7}, stride in {1, 2}; the caller states the top / left padding and the output size,
 * every tap outside the H x W input reads as zero.  w packed [K*K][C] (tap major).  Single level (no pyramid). */
int32_t fd_dwconv2d_nhwc(const float* x, int32_t x_cs, int32_t x_co, const float* w, const float* scale,
                         const float* shift, float* y, int32_t y_cs, int32_t y_co, int32_t N, int32_t H, int32_t W,
                         int32_t C, int32_t K, int32_t stride, int32_t pad_top, int32_t pad_left, int32_t Ho, int32_t Wo,
                         int32_t act, fd_stream_t stream);
/* MBConv expand -> depthwise in ONE kernel (efficientnet_pytorch 0.7.1 MBConvBlock.forward behind model/backbone/efficientnetv1.py:11-26):
 *     y = swish(bn1(dwconv_kxk(swish(bn0(conv1x1(x, W_expand))))))   k in {3, 5}, stride in {1, 2}, static "SAME" padding (pad_top / pad_left; zeros on every side)
 * without the expanded map (six times the block input) ever reaching HBM: a workgroup stages the input patch of one TO x TO output tile in LDS, computes the expand
 * GEMM of the patch on the fp32 MFMA 32 expanded channels at a time and runs the depthwise conv from LDS.  It also writes the squeeze-excitation pooling's partial
 * sums -- pool[n][tile][c], fd_mbconv_pool_bytes() bytes -- so fd_se_gate_from_pool needs no pass over y.  Cin % 8 == 0 in 8 .. 48, mid % 4 == 0.
 * w_expand_frag: [ceil(mid / 32)][Cin / 8][2][32][4] floats, element (cb, g, h, l, jj) = W_expand[32 cb + l][h * (Cin / 2) + 4 g + jj] (rows >= mid zero); w_dw: [K * K][mid].
 * Results equal the separate launches up to the summation order of the expand GEMM. */
int64_t fd_mbconv_pool_bytes(int32_t N, int32_t Ho, int32_t Wo, int32_t mid, int32_t K, int32_t stride);
int32_t fd_mbconv_expand_dw_nhwc(const float* x, int32_t x_cs, int32_t x_co, const float* w_expand_frag, const float* scale0, const float* shift0,
                                 const float* w_dw, const float* scale1, const float* shift1, float* y, int32_t y_cs, int32_t y_co, float* pool,
                                 int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t mid, int32_t K, int32_t stride, int32_t pad_top, int32_t pad_left,
                                 int32_t Ho, int32_t Wo, fd_stream_t stream);
/* The SE gates (fd_se_scale_nhwc with y = NULL) from those per-tile partial sums: T = tiles per image; workspace = fd_se_workspace_bytes(N, HW, C). */
int32_t fd_se_gate_from_pool(const float* pool, int32_t T, const float* w1, const float* b1, const float* w2, const float* b2, int32_t N, int32_t HW,
                             int32_t C, int32_t Cr, void* workspace, fd_stream_t stream);

/* 3-channel stem convolution (EfficientNet._conv_stem 3x3 stride 2 + _bn0 + swish) on the [N][H][W][4] image layout
 * fd_nchw3_to_nhwc4 / fd_preprocess_u8_nhwc4 / fd_collate_u8_nhwc4 produce: y = act(conv(x)*scale + shift).
 * w packed [K*K][4][Cout] (tap, input channel, output channel; input channel 3 is ignored).  K = 3, stride in {1, 2},
 * Cout % 4 == 0; padding as for fd_dwconv2d_nhwc. */
int32_t fd_stem_conv_nhwc4(const float* x4, const float* w, const float* scale, const float* shift, float* y,
                           int32_t y_cs, int32_t y_co, int32_t N, int32_t H, int32_t W, int32_t Cout, int32_t K,
                           int32_t stride, int32_t pad_top, int32_t pad_left, int32_t Ho, int32_t Wo, int32_t act,
                           fd_stream_t stream);

/* Mixed-aspect batch assembly on the device (dataset/voc.py:128-132 pad-to-32, :141-156 collate_fn):
 * N resized uint8 images of different sizes -> ONE normalised fp32 [N][H][W][4] batch (the stem's input layout).
 * images_dev: DEVICE array of N device pointers to [h_n][w_n][3] uint8; hw_dev: DEVICE int32 [N][2] = (h_n, w_n);
 * H, W >= every (h_n, w_n): the batch canvas (the host picks max over n of h_n + 32 - h_n % 32, as the reference does).
 * Pixels outside an image are uint8 zeros BEFORE normalisation, i.e. (0 - mean) / std, exactly what the reference's
 * zero-pad-then-Normalize produces.  cv2.resize stays on the host.  mean3 / std3 are HOST pointers to 3 floats. */
int32_t fd_collate_u8_nhwc4(const uint8_t* const* images_dev, const int32_t* hw_dev, float* y, int32_t N, int32_t H,
                            int32_t W, const float* mean3, const float* std3, fd_stream_t stream);

/* GroupNorm(G, C) + activation (nn.GroupNorm in HISFCOSHead / HeadFCOS, HISFcos.py:190-204, Fcos.py:102-109).
 * Two launches: partial moments (fp64 accumulation, fixed order) then normalise+affine+act.
 * workspace: fd_groupnorm_workspace_bytes(segs, G). x and y may alias. */
int64_t fd_groupnorm_workspace_bytes(const fd_segs* segs, int32_t G);
int32_t fd_groupnorm_act_nhwc(const float* x, int32_t x_cs, int32_t x_co, const float* gamma, const float* beta,
                              float* y, int32_t y_cs, int32_t y_co, int32_t C, int32_t G, float eps, int32_t act,
                              const fd_segs* segs, void* workspace, fd_stream_t stream);

/* Backward of the call above for the train step (reference train.py:175-181 differentiates nn.GroupNorm + ReLU / SiLU,
 * HISFcos.py:190-222): given dy (gradient w.r.t. the activated output) writes dx, dgamma[C], dbeta[C].
 * `fwd_workspace` is the forward call's workspace, unmodified (its partial moments give mean / rstd);
 * `workspace`: fd_groupnorm_bwd_workspace_bytes(segs, C).  act in {NONE, RELU, SILU}.  dx may alias dy.
 * fp64 partial sums in fixed order: deterministic. */
int64_t fd_groupnorm_bwd_workspace_bytes(const fd_segs* segs, int32_t C);
int32_t fd_groupnorm_act_bwd_nhwc(const float* x, int32_t x_cs, int32_t x_co, const float* dy, int32_t dy_cs,
                                  int32_t dy_co, const float* gamma, const float* beta, float* dx, int32_t dx_cs,
                                  int32_t dx_co, float* dgamma, float* dbeta, int32_t C, int32_t G, float eps,
                                  int32_t act, const fd_segs* segs, const void* fwd_workspace, void* workspace,
                                  fd_stream_t stream);

/* Squeeze-excitation (SEBlock, modules.py:107-121; also the SE stage of an EfficientNet MBConv block, whose middle
 * activation is the same swish): y = x * sigmoid(W2 silu(W1 mean_hw(x) + b1) + b2).
 * w1 [Cr][C], w2 [C][Cr]; C % 4 == 0, C <= 4096, Cr <= 1024.  workspace: fd_se_workspace_bytes(N, HW, C).  x and y may alias. */
int64_t fd_se_workspace_bytes(int32_t N, int32_t HW, int32_t C);
/* (y == NULL: compute the gates only -- they are left as [N][C] floats at byte offset fd_se_workspace_bytes(N, HW, C) - N * C * 4 of the
 *  workspace -- for a consumer that applies them itself: fd_conv_params.gate) */
int32_t fd_se_scale_nhwc(const float* x, int32_t x_cs, int32_t x_co, const float* w1, const float* b1,
                         const float* w2, const float* b2, float* y, int32_t y_cs, int32_t y_co, int32_t N,
                         int32_t HW, int32_t C, int32_t Cr, void* workspace, fd_stream_t stream);

/* Backward of the call above for the train step (train.py:175-181 differentiates HisBlock.conv1_2): dx and the gradients of
 * both 1x1 convs (dw1 [Cr][C], db1 [Cr], dw2 [C][Cr], db2 [C]).  `fwd_workspace` = the forward call's workspace, untouched
 * since (pooled sums, gates); `workspace`: fd_se_bwd_workspace_bytes(N, HW, C).  Deterministic (fixed-order sums). */
int64_t fd_se_bwd_workspace_bytes(int32_t N, int32_t HW, int32_t C);
int32_t fd_se_scale_bwd_nhwc(const float* x, int32_t x_cs, int32_t x_co, const float* dy, int32_t dy_cs, int32_t dy_co,
                             const float* w1, const float* b1, const float* w2, const float* b2, float* dx, int32_t dx_cs,
                             int32_t dx_co, float* dw1, float* db1, float* dw2, float* db2, int32_t N, int32_t HW, int32_t C,
                             int32_t Cr, const void* fwd_workspace, void* workspace, fd_stream_t stream);

/* Elementwise activation y = act(x) on a channel view and its backward dx = dy * act'(x) (x = the saved INPUT):
 * nn.SiLU / nn.ReLU after a BatchNorm (HISFcos.py:100,112), ScaleExp (modules.py:170-176: FD_ACT_EXP, param = scale). */
int32_t fd_act_nhwc(const float* x, int32_t x_cs, int32_t x_co, float* y, int32_t y_cs, int32_t y_co, int64_t rows, int32_t C,
                    int32_t act, float param, fd_stream_t stream);
int32_t fd_act_bwd_nhwc(const float* x, int32_t x_cs, int32_t x_co, const float* dy, int32_t dy_cs, int32_t dy_co, float* dx,
                        int32_t dx_cs, int32_t dx_co, int64_t rows, int32_t C, int32_t act, float param, fd_stream_t stream);
/* The same pass over f16 maps (x, dy, dx hold _Float16 elements; 8-byte aligned views): AMP activations / gradients stored as f16 (train.py:175-181);
 * the derivative is evaluated in fp32 and the product rounded once. */
int32_t fd_act_bwd_nhwc_h(const void* x, int32_t x_cs, int32_t x_co, const void* dy, int32_t dy_cs, int32_t dy_co, void* dx,
                          int32_t dx_cs, int32_t dx_co, int64_t rows, int32_t C, int32_t act, float param, fd_stream_t stream);

/* Backward of fd_maxpool_nhwc (x = the forward input): dx[p] = sum of dy over the windows whose first maximum is p (the
 * argmax torch's max_pool2d keeps).  Gather form, deterministic.  The fused "+ add" passes dy through unchanged.
 * Backward of fd_upsample2x_add_nhwc w.r.t. its low-resolution input: dx = sum of the 2x2 block of dy (dy: [N][2H][2W]). */
int32_t fd_maxpool_bwd_nhwc(const float* x, int32_t x_cs, int32_t x_co, const float* dy, int32_t dy_cs, int32_t dy_co, float* dx,
                            int32_t dx_cs, int32_t dx_co, int32_t N, int32_t H, int32_t W, int32_t C, int32_t k, int32_t s,
                            int32_t pad, fd_stream_t stream);
int32_t fd_upsample2x_bwd_nhwc(const float* dy, int32_t dy_cs, int32_t dy_co, float* dx, int32_t dx_cs, int32_t dx_co, int32_t N,
                               int32_t H, int32_t W, int32_t C, fd_stream_t stream);

/* GroupNorm fused into its neighbours (HISFCOSHead, HISFcos.py:216-225).  A producer (fd_conv_params.gn_stats, fd_dwconv3x3_gn_nhwc) leaves
 * rowstats[m][g] = (sum, sum of squares) of group g's channels of row m (fp32 float2).  fd_groupnorm_from_rowstats reduces them per
 * (level, image, group) in fp64 in a fixed order into `workspace` (fd_groupnorm_workspace_bytes(segs, G): the layout fd_groupnorm_act_nhwc
 * leaves, so fd_groupnorm_apply_nhwc and fd_groupnorm_act_bwd_nhwc work on it) and, with coef != NULL, writes the affine a consumer applies
 * in its loader: coef[(level * batch + image)][0][c] = rstd * gamma[c], [..][1][c] = beta[c] - mean * rstd * gamma[c]  ([imgs][2][C] floats).
 * 2 * G must divide 256.  fd_groupnorm_apply_nhwc: y = act(x * a + b) from statistics already in `workspace` (one pass over the map).
 * fd_dwconv3x3_gn_nhwc: depthwise 3x3 (stride 1, pad 1, no bias) that reads its input as in_act(x * a + b) (in_coef as above, NULL: plain;
 * the zero padding is applied AFTER the normalisation, as in the reference) and writes gn_stats of its output (NULL: none). */
int32_t fd_groupnorm_from_rowstats(const float* rowstats, int32_t C, int32_t G, float eps, const float* gamma, const float* beta,
                                   const fd_segs* segs, void* workspace, float* coef, fd_stream_t stream);
int32_t fd_groupnorm_apply_nhwc(const float* x, int32_t x_cs, int32_t x_co, const float* gamma, const float* beta, float* y,
                                int32_t y_cs, int32_t y_co, int32_t C, int32_t G, float eps, int32_t act, const fd_segs* segs,
                                const void* workspace, fd_stream_t stream);
/* The statistics steps of fd_groupnorm_act_nhwc alone (`workspace` as there) + optionally the affine `coef` [imgs][2][C] of
 * fd_groupnorm_from_rowstats: for a map whose consumers normalise it themselves (fd_conv_params.gate_b, fd_coef_apply_nhwc). */
int32_t fd_groupnorm_stats_nhwc(const float* x, int32_t x_cs, int32_t x_co, int32_t C, int32_t G, float eps, const float* gamma, const float* beta,
                                const fd_segs* segs, void* workspace, float* coef, fd_stream_t stream);
/* y = act(x * coef_a[img][c] + coef_b[img][c]) over a C-channel view, img = level * batch + image, coefficient rows coef_cs floats apart
 * (views into fd_groupnorm_from_rowstats' coef: a CHANNEL SLICE of a map whose statistics were reduced together with other channels -- the
 * class half of the head tower, whose box half is normalised inside the FD_TILE_NARROW predictor's loader).  C / 4 must divide 256. */
int32_t fd_coef_apply_nhwc(const float* x, int32_t x_cs, int32_t x_co, const float* coef_a, const float* coef_b, int32_t coef_cs, float* y,
                           int32_t y_cs, int32_t y_co, int32_t C, int32_t act, const fd_segs* segs, fd_stream_t stream);
int32_t fd_dwconv3x3_gn_nhwc(const float* x, int32_t x_cs, int32_t x_co, const float* w, const float* in_coef, int32_t in_act,
                             float* y, int32_t y_cs, int32_t y_co, int32_t C, float* gn_stats, int32_t gn_groups,
                             const fd_segs* segs, fd_stream_t stream);

/* nn.BatchNorm2d in TRAINING mode (the FPN BatchNorms under the reference's model.train(), train.py:151) runs on the
 * GroupNorm entry points: batch statistics per channel = GroupNorm statistics of ONE image of (batch*H) x W pixels with
 * G = C groups (fd_segs{nseg 1, batch 1, H = batch*H, W}); forward / backward are fd_groupnorm_act_nhwc /
 * fd_groupnorm_act_bwd_nhwc.  This call then folds the batch statistics the forward left in its workspace into the running
 * statistics (momentum update with the unbiased variance, as nn.BatchNorm2d). rows = batch*H*W. */
int32_t fd_batchnorm_update_running(const void* gn_workspace, int64_t rows, int32_t C, float momentum, float eps,
                                    float* running_mean, float* running_var, fd_stream_t stream);
/* The same with the row count in DEVICE memory (one double): nn.SyncBatchNorm's global count is the result of an all-reduce and differs
 * per step when the ranks' batches are padded to different H x W (dataset/voc.py:141-171); reading it here costs no host synchronisation. */
int32_t fd_batchnorm_update_running_dev(const void* gn_workspace, const double* count_dev, int32_t C, float momentum, float eps,
                                        float* running_mean, float* running_var, fd_stream_t stream);

/* nn.SyncBatchNorm in TRAINING mode (train.py:101-103: SyncBatchNorm.convert_sync_batchnorm after the DDP wrap; SURVEY 2.1 collective C3):
 * batch statistics over ALL ranks' rows.  Forward and backward are each cut in two around ONE all-reduce that the CALLER issues
 * (torch.distributed / RCCL; the library owns no communicator):
 *   forward   phase 1: sums[2c], sums[2c+1] = this rank's sum x, sum x^2 of channel c (fp64, fixed order)      -> all-reduce(sums, rows)
 *             phase 2: y = act((x - mean) * rstd * gamma + beta) with mean / rstd from the global sums and total_rows; (mean, rstd) stay in
 *                      `workspace` (fd_groupnorm_workspace_bytes of {1 image of rows x 1, G = C}) for the backward and for
 *                      fd_batchnorm_update_running(workspace, total_rows, ...)
 *   backward  phase 1: sums[c], sums[C+c] = this rank's sum dz, sum dz * xhat (dz = dy * act'); dgamma / dbeta from THESE local sums
 *                      (DDP averages them with the other gradients, as torch's SyncBatchNorm does)               -> all-reduce(sums)
 *             phase 2: dx = rstd * (gamma * dz - mean_global(gamma dz) - xhat * mean_global(gamma dz xhat))
 *   `workspace` of the backward: fd_groupnorm_bwd_workspace_bytes of the same segment table.  rows = this rank's batch*H*W.
 *   total_rows (phase 2): the GLOBAL row count as a host value, or <= 0: it is read on the device from sums[2C] -- the caller lets its own
 *   row count ride behind the 2C sums of the forward all-reduce ([2C + 1] doubles) and hands the backward a [2C + 1] buffer whose last word
 *   it copies from there (device to device): ranks whose batches are padded to different H x W (dataset/voc.py:141-171) then need no host
 *   round trip, and no rank assumes rows * world_size. */
int32_t fd_batchnorm_sync_fwd_nhwc(const float* x, int32_t x_cs, int32_t x_co, const float* gamma, const float* beta, float* y,
                                   int32_t y_cs, int32_t y_co, int64_t rows, int32_t C, float eps, int32_t act, int32_t phase,
                                   double* sums, double total_rows, void* workspace, fd_stream_t stream);
int32_t fd_batchnorm_sync_bwd_nhwc(const float* x, int32_t x_cs, int32_t x_co, const float* dy, int32_t dy_cs, int32_t dy_co,
                                   const float* gamma, const float* beta, float* dx, int32_t dx_cs, int32_t dx_co, float* dgamma,
                                   float* dbeta, int64_t rows, int32_t C, float eps, int32_t act, int32_t phase, double* sums,
                                   double total_rows, const void* fwd_workspace, void* workspace, fd_stream_t stream);

/* ------------------------------------------------------------------------------------------- */
/* Detection post-processing (reference model/modules/head.py:8-102,152-162, utill/utills.py:58-73)   */

/* Per location: score = sqrt(max_c sigmoid(cls) * sigmoid(cnt)), class = argmax_c + 1 (first max),
 * box = (cx - l, cy - t, cx + r, cy + b), (cx, cy) = (x*s + s/2, y*s + s/2).
 * cls/cnt/reg are pyramid buffers (rows x C / 1 / 4 with channel strides).  Outputs are per image,
 * levels concatenated in segment order, row-major inside a level: scores [N][L], classes [N][L]
 * (int32), boxes [N][L][4], L = sum H*W. */
int32_t fd_fcos_decode(const float* cls, int32_t cls_cs, int32_t cls_co, const float* cnt, int32_t cnt_cs,
                       int32_t cnt_co, const float* reg, int32_t reg_cs, int32_t reg_co, int32_t num_classes,
                       const fd_segs* segs, const int32_t* strides, float* scores, int32_t* classes,
                       float* boxes, fd_stream_t stream);

/* torch.topk(score, K, dim=1, largest=True, sorted=True) + gathers (head.py:69-80).  Ties: lower
 * location index first.  Outputs [N][K]; top_idx (int32 location index) may be NULL.  K <= 1024 runs in LDS / registers with no
 * workspace (fd_topk_workspace_bytes = 0); any larger K <= L (FCOSHead accepts any max_detection_box, head.py:41-50) sorts its candidate
 * list in `workspace` (fd_topk_workspace_bytes bytes, 8-byte aligned). */
int64_t fd_topk_workspace_bytes(int32_t N, int32_t L, int32_t K);
int32_t fd_fcos_topk(const float* scores, const int32_t* classes, const float* boxes, int32_t N, int32_t L,
                     int32_t K, float* top_scores, int64_t* top_classes, float* top_boxes, int32_t* top_idx,
                     void* workspace, fd_stream_t stream);

/* score >= score_thr mask + torchvision.ops.batched_nms (coordinate-offset trick) + gathers
 * (FCOSHead.post_process, head.py:84-102).  Input rows must be score-descending (fd_fcos_topk
 * output).  Greedy rule: suppress j when (double)iou(i,j) > iou_thr, iou on class-offset boxes
 * without the "+1" pixel convention.  Outputs padded to [N][K] (rows >= counts[n] zero-filled),
 * keep_idx[n][r] = index into the K input rows.  K <= 1024: two launches, the suppression bitmask is built by
 * K/64 workgroups per image into `workspace` (fd_nms_workspace_bytes), one workgroup per image then scans it in LDS.  K > 1024
 * (up to 262144): the same rule with the class-offset boxes and bitmask rows of ceil(K/64) words in the workspace (three launches).
 * Outputs must not alias inputs. */
int64_t fd_nms_workspace_bytes(int32_t N, int32_t K);
int32_t fd_batched_nms(const float* scores, const int64_t* classes, const float* boxes, int32_t N, int32_t K,
                       float score_thr, double iou_thr, float* out_scores, int64_t* out_classes,
                       float* out_boxes, int32_t* keep_idx, int32_t* counts, void* workspace, fd_stream_t stream);

/* Class-agnostic greedy NMS with the "+1" pixel convention, keep while ovr <= (float)thr
 * (DataEncoder._box_nms, utill/utills.py:221-255; mode 0 = 'union', 1 = 'min').
 * One problem per call-row: boxes [N][K][4], scores [N][K] (any order; sorted internally,
 * ties by lower index), n_valid [N] or NULL (=K).  keep_idx [N][K] (original indices, score
 * descending), counts [N].  K <= 1024. */
int32_t fd_box_nms_plus1(const float* boxes, const float* scores, const int32_t* n_valid, int32_t N, int32_t K,
                         float thr, int32_t mode, int32_t* keep_idx, int32_t* counts, fd_stream_t stream);

/* Pairwise IoU [Na][Nb]; plus_one=1: DataEncoder._box_iou (utills.py:201-218); 0: test.py:23-53 */
int32_t fd_pairwise_iou(const float* a, const float* b, int32_t Na, int32_t Nb, int32_t plus_one, float* out,
                        fd_stream_t stream);

/* ClipBoxes (head.py:152-162): clamp(min 0), x <= W-1, y <= H-1, in place on [n_boxes][4] */
int32_t fd_clip_boxes(float* boxes, int64_t n_boxes, int32_t img_h, int32_t img_w, fd_stream_t stream);

/* Detection records for the ONE collective of image-sharded multi-GPU inference: the padded
 * outputs of fd_batched_nms of B images as one fp32 message records[B][K+1][6]:
 *   records[b][0] = (counts[b], 0, 0, 0, 0, 0);  records[b][1+r] = (x1, y1, x2, y2, score, class), r < K.
 * Counts and class ids are < 2^24, exact in fp32.  fd_unpack_detections is the inverse (on the gathered [W*B] images). */
int32_t fd_pack_detections(const float* scores, const int64_t* classes, const float* boxes, const int32_t* counts,
                           int32_t B, int32_t K, float* records, fd_stream_t stream);
int32_t fd_unpack_detections(const float* records, int32_t B, int32_t K, float* scores, int64_t* classes, float* boxes,
                             int32_t* counts, fd_stream_t stream);

/* ------------------------------------------------------------------------------------------- */
/* LTRB IoU / GIoU regression loss (reference model/loss.py:116-177), fused masked forward + backward.
 * pred/target [B][L][4], mask [B][L] (uint8, positives).  mode 0 = 'iou', 1 = 'giou'.
 * loss_per_image [B] = sum over positives (NOT yet divided by num_pos), num_pos [B] int32.
 * backward: grad_pred [B][L][4] = d(sum_b loss_b * gscale[b]) / d pred  (gscale = upstream/num_pos). */
int32_t fd_ltrb_iou_loss_fwd(const float* pred, const float* target, const uint8_t* mask, int32_t B, int32_t L,
                             int32_t mode, float* loss_per_image, int32_t* num_pos, fd_stream_t stream);
int32_t fd_ltrb_iou_loss_bwd(const float* pred, const float* target, const uint8_t* mask, const float* gscale,
                             int32_t B, int32_t L, int32_t mode, float* grad_pred, fd_stream_t stream);

/* Focal classification loss from logits (model/loss.py:6-28,180-193), alpha 0.25 / gamma 2 in the reference.
 * logits [B][L][C], labels [B][L] int64 (1..C, 0 = background).  loss_per_image [B] = sum over [L][C]
 * (not yet divided by num_pos).  backward: grad [B][L][C] = dloss_b/dlogits * gscale[b].  Only gamma == 2 is built.
 * workspace: fd_focal_workspace_bytes(B). */
int64_t fd_focal_workspace_bytes(int32_t B);
int32_t fd_focal_loss_fwd(const float* logits, const int64_t* labels, int32_t B, int32_t L, int32_t C, float alpha,
                          float gamma, float* loss_per_image, void* workspace, fd_stream_t stream);
int32_t fd_focal_loss_bwd(const float* logits, const int64_t* labels, const float* gscale, int32_t B, int32_t L, int32_t C,
                          float alpha, float gamma, float* grad_logits, fd_stream_t stream);

/* Centerness loss: binary_cross_entropy_with_logits(reduction='sum') over positives (model/loss.py:31-57).
 * x / target / mask [B][L]; loss_per_image [B] (sum), num_pos [B]; backward grad [B][L] = (sigmoid(x)-t)*gscale[b]. */
int32_t fd_bce_logits_loss_fwd(const float* x, const float* target, const uint8_t* mask, int32_t B, int32_t L,
                               float* loss_per_image, int32_t* num_pos, fd_stream_t stream);
int32_t fd_bce_logits_loss_bwd(const float* x, const float* target, const uint8_t* mask, const float* gscale, int32_t B,
                               int32_t L, float* grad, fd_stream_t stream);

/* FCOS target assignment (FCOSGenTargets, model/modules/head.py:211-316): gt_boxes [B][M][4] xyxy (pad -1),
 * labels [B][M] int64 (pad -1); per level: stride, (range_lo, range_hi]; centre-sampling radius = stride*radius_ratio
 * (1.5 in the reference).  Outputs per image, levels concatenated: cls_target [B][L] int64 (0 = negative),
 * cnt_target [B][L] (-1 = negative), reg_target [B][L][4] LTRB (-1 = negative).  segs->batch = B. */
int32_t fd_fcos_gen_targets(const float* gt_boxes, const int64_t* labels, int32_t M, const fd_segs* segs,
                            const int32_t* strides, const int32_t* range_lo, const int32_t* range_hi, float radius_ratio,
                            int64_t* cls_target, float* cnt_target, float* reg_target, fd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FCOSDET_H_ */
