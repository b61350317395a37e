// fd_mbconv.hip — the layers an EfficientNet (MBConv) trunk adds to the FCOS stack on NHWC fp32 rows (gfx950):
//   * depthwise k x k conv, k in {3, 5}, stride in {1, 2}, asymmetric (TensorFlow "SAME", static) zero padding,
//     fused BatchNorm(eval) scale / shift + swish;
//   * the 3-channel stem conv (k x k, stride s, asymmetric padding) on the [N][H][W][4] image layout;
//   * batch assembly of mixed-aspect images: pad-to-32 + pad-to-batch-max + normalise (dataset/voc.py:128-132,141-156).
// Restates efficientnet_pytorch 0.7.1 (third-party, pinned by the reference's README.md:16; wrapped by the reference's
// model/backbone/efficientnetv1.py:11-26).  HBM-bound: 16 bytes per lane, rows x channel-quads flattened.
#include "fd_common.h"

#define FD_GRID_CAP 16384

static inline unsigned grid_for(long work, int block) {
    long g = (work + block - 1) / block;
    if (g > FD_GRID_CAP) g = FD_GRID_CAP;
    if (g < 1) g = 1;
    return (unsigned)g;
}

static inline bool view_ok(const void* p, int cs, int co, int C) {
    return p && C % 4 == 0 && cs % 4 == 0 && co % 4 == 0 && cs >= co + C && ((uintptr_t)p & 15) == 0;
}

// ------------------------------------------------------------------------------ depthwise k x k, stride s
// One thread = S adjacent output pixels of one output row x one channel quad: the K x ((S-1)*STRIDE + K) input window is
// loaded once per row of taps (a pixel-per-thread kernel is bound by vector-load issue, see fd_layers.hip), the K*K
// weights once per S outputs.  Taps outside the image contribute nothing (zero padding on every side, so the caller only
// states the top / left padding and the output size).
template <int K, int STRIDE, int S>
__global__ __launch_bounds__(256) void dwconv_kxk_kernel(const float* __restrict__ x, int x_cs, int x_co,
                                                          const float* __restrict__ wt, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, float* __restrict__ y, int y_cs,
                                                          int y_co, int C, int act, int H, int W, int Ho, int Wo, int pad_t,
                                                          int pad_l, long total) {
    constexpr int WIN = (S - 1) * STRIDE + K;
    const int C4 = C >> 2;
    const int spr = (Wo + S - 1) / S;          // strips per output row
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int q;
        long t = fd_div(i, C4, q);
        int sx; t = fd_div(t, spr, sx);
        int ho;
        const long n = fd_div(t, Ho, ho);
        const int wo0 = sx * S;
        const int hi0 = ho * STRIDE - pad_t, wi0 = wo0 * STRIDE - pad_l;
        const float* xb = x + (n * H * (long)W) * x_cs + x_co + 4 * q;
        float4 acc[S];
#pragma unroll
        for (int j = 0; j < S; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < K; ++r) {
            const int hi = hi0 + r;
            if ((unsigned)hi >= (unsigned)H) continue;
            float4 u[WIN];
#pragma unroll
            for (int c = 0; c < WIN; ++c) {
                const int wi = wi0 + c;
                u[c] = (unsigned)wi < (unsigned)W ? *reinterpret_cast<const float4*>(xb + ((long)hi * W + wi) * x_cs)
                                                  : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int c = 0; c < K; ++c) {
                const float4 k = *reinterpret_cast<const float4*>(wt + (r * K + c) * C + 4 * q);
#pragma unroll
                for (int j = 0; j < S; ++j) {
                    if ((unsigned)(wi0 + j * STRIDE + c) < (unsigned)W) {   // a padding tap adds nothing (not even +0*w)
                        const float4 v = u[j * STRIDE + c];
                        acc[j].x = fmaf(v.x, k.x, acc[j].x); acc[j].y = fmaf(v.y, k.y, acc[j].y);
                        acc[j].z = fmaf(v.z, k.z, acc[j].z); acc[j].w = fmaf(v.w, k.w, acc[j].w);
                    }
                }
            }
        }
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sf = make_float4(0.f, 0.f, 0.f, 0.f);
        if (scale) sc = *reinterpret_cast<const float4*>(scale + 4 * q);
        if (shift) sf = *reinterpret_cast<const float4*>(shift + 4 * q);
        float* yb = y + ((n * Ho + ho) * (long)Wo) * y_cs + y_co + 4 * q;
#pragma unroll
        for (int j = 0; j < S; ++j) {
            if (wo0 + j >= Wo) break;
            float4 o;
            o = fd_act4(make_float4(acc[j].x * sc.x + sf.x, acc[j].y * sc.y + sf.y, acc[j].z * sc.z + sf.z, acc[j].w * sc.w + sf.w), act, 0.f);
            *reinterpret_cast<float4*>(yb + (long)(wo0 + j) * y_cs) = o;
        }
    }
}

extern "C" int32_t fd_dwconv2d_nhwc(const float* x, int32_t x_cs, int32_t x_co, const float* w, const float* scale,
                                    const float* shift, float* y, int32_t y_cs, int32_t y_co, int32_t N, int32_t H, int32_t W,
                                    int32_t C, int32_t K, int32_t stride, int32_t pad_top, int32_t pad_left, int32_t Ho,
                                    int32_t Wo, int32_t act, fd_stream_t stream) {
    FD_REQUIRE(view_ok(x, x_cs, x_co, C) && view_ok(y, y_cs, y_co, C) && w && ((uintptr_t)w & 15) == 0, FD_E_INVAL,
               "fd_dwconv2d: channel views must be 4-aligned (C=%d)", C);
    FD_REQUIRE(N >= 1 && H >= 1 && W >= 1 && Ho >= 1 && Wo >= 1 && pad_top >= 0 && pad_left >= 0, FD_E_INVAL,
               "fd_dwconv2d: bad geometry");
    FD_REQUIRE(pad_top < K && pad_left < K && (long)(Ho - 1) * stride - pad_top < H && (long)(Wo - 1) * stride - pad_left < W,
               FD_E_INVAL, "fd_dwconv2d: output %dx%d reaches past the %dx%d input (k=%d s=%d pad %d/%d)", Ho, Wo, H, W, K, stride,
               pad_top, pad_left);
    FD_REQUIRE((long)N * H * W * x_cs < (1L << 31) * 4 && (long)N * Ho * Wo * y_cs < (1L << 31) * 4, FD_E_UNSUPPORTED,
               "fd_dwconv2d: tensor too large");
    constexpr int S = 4;
    const long total = (long)N * Ho * ((Wo + S - 1) / S) * (C / 4);
    hipStream_t st = (hipStream_t)stream;
#define FD_DW_LAUNCH(KK, SS)                                                                                                      \
    hipLaunchKernelGGL((dwconv_kxk_kernel<KK, SS, S>), dim3(grid_for(total, 256)), dim3(256), 0, st, x, x_cs, x_co, w, scale, shift, \
                       y, y_cs, y_co, C, act, H, W, Ho, Wo, pad_top, pad_left, total)
    if (K == 3 && stride == 1) FD_DW_LAUNCH(3, 1);
    else if (K == 3 && stride == 2) FD_DW_LAUNCH(3, 2);
    else if (K == 5 && stride == 1) FD_DW_LAUNCH(5, 1);
    else if (K == 5 && stride == 2) FD_DW_LAUNCH(5, 2);
    else if (K == 7 && stride == 1) FD_DW_LAUNCH(7, 1);
    else if (K == 7 && stride == 2) FD_DW_LAUNCH(7, 2);
    else { fd_set_error("fd_dwconv2d: kernel %d stride %d has no kernel (k in {3,5,7}, stride in {1,2})", K, stride); return FD_E_UNSUPPORTED; }
#undef FD_DW_LAUNCH
    FD_CHECK_LAUNCH("fd_dwconv2d_nhwc");
    return FD_OK;
}

// ------------------------------------------------------------------------------ 3-channel stem conv
// x is [N][H][W][4] (3 channels + zero), w packed [K*K][4][Cout] (tap, input channel, output channel; channel 3 zero).
// One thread = S adjacent output pixels x 4 output channels; 4*K*K weight quads are read once per S pixels.
template <int K, int STRIDE, int S>
__global__ __launch_bounds__(256) void stem_conv_kernel(const float4* __restrict__ x, const float* __restrict__ wt,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         float* __restrict__ y, int y_cs, int y_co, int Cout, int act, int H,
                                                         int W, int Ho, int Wo, int pad_t, int pad_l, long total) {
    constexpr int WIN = (S - 1) * STRIDE + K;
    const int C4 = Cout >> 2;
    const int spr = (Wo + S - 1) / S;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int q;
        long t = fd_div(i, C4, q);
        int sx; t = fd_div(t, spr, sx);
        int ho;
        const long n = fd_div(t, Ho, ho);
        const int wo0 = sx * S;
        const int hi0 = ho * STRIDE - pad_t, wi0 = wo0 * STRIDE - pad_l;
        const float4* xb = x + n * H * (long)W;
        float4 acc[S];
#pragma unroll
        for (int j = 0; j < S; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < K; ++r) {
            const int hi = hi0 + r;
            if ((unsigned)hi >= (unsigned)H) continue;
            float4 u[WIN];
#pragma unroll
            for (int c = 0; c < WIN; ++c) {
                const int wi = wi0 + c;
                u[c] = (unsigned)wi < (unsigned)W ? xb[(long)hi * W + wi] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int c = 0; c < K; ++c) {
                const float* wp = wt + (long)((r * K + c) * 4) * Cout + 4 * q;
                const float4 k0 = *reinterpret_cast<const float4*>(wp);
                const float4 k1 = *reinterpret_cast<const float4*>(wp + Cout);
                const float4 k2 = *reinterpret_cast<const float4*>(wp + 2 * Cout);
#pragma unroll
                for (int j = 0; j < S; ++j) {
                    if ((unsigned)(wi0 + j * STRIDE + c) < (unsigned)W) {
                        const float4 v = u[j * STRIDE + c];
                        acc[j].x = fmaf(v.x, k0.x, acc[j].x); acc[j].y = fmaf(v.x, k0.y, acc[j].y);
                        acc[j].z = fmaf(v.x, k0.z, acc[j].z); acc[j].w = fmaf(v.x, k0.w, acc[j].w);
                        acc[j].x = fmaf(v.y, k1.x, acc[j].x); acc[j].y = fmaf(v.y, k1.y, acc[j].y);
                        acc[j].z = fmaf(v.y, k1.z, acc[j].z); acc[j].w = fmaf(v.y, k1.w, acc[j].w);
                        acc[j].x = fmaf(v.z, k2.x, acc[j].x); acc[j].y = fmaf(v.z, k2.y, acc[j].y);
                        acc[j].z = fmaf(v.z, k2.z, acc[j].z); acc[j].w = fmaf(v.z, k2.w, acc[j].w);
                    }
                }
            }
        }
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sf = make_float4(0.f, 0.f, 0.f, 0.f);
        if (scale) sc = *reinterpret_cast<const float4*>(scale + 4 * q);
        if (shift) sf = *reinterpret_cast<const float4*>(shift + 4 * q);
        float* yb = y + ((n * Ho + ho) * (long)Wo) * y_cs + y_co + 4 * q;
#pragma unroll
        for (int j = 0; j < S; ++j) {
            if (wo0 + j >= Wo) break;
            float4 o;
            o = fd_act4(make_float4(acc[j].x * sc.x + sf.x, acc[j].y * sc.y + sf.y, acc[j].z * sc.z + sf.z, acc[j].w * sc.w + sf.w), act, 0.f);
            *reinterpret_cast<float4*>(yb + (long)(wo0 + j) * y_cs) = o;
        }
    }
}

extern "C" int32_t fd_stem_conv_nhwc4(const float* x4, const float* w, const float* scale, const float* shift, float* y,
                                      int32_t y_cs, int32_t y_co, int32_t N, int32_t H, int32_t W, int32_t Cout, int32_t K,
                                      int32_t stride, int32_t pad_top, int32_t pad_left, int32_t Ho, int32_t Wo, int32_t act,
                                      fd_stream_t stream) {
    FD_REQUIRE(x4 && w && ((uintptr_t)x4 & 15) == 0 && ((uintptr_t)w & 15) == 0 && view_ok(y, y_cs, y_co, Cout), FD_E_INVAL,
               "fd_stem_conv: bad pointer / output channel view (Cout=%d must be a multiple of 4)", Cout);
    FD_REQUIRE(N >= 1 && H >= 1 && W >= 1 && Ho >= 1 && Wo >= 1 && pad_top >= 0 && pad_left >= 0 && pad_top < K && pad_left < K &&
                   (long)(Ho - 1) * stride - pad_top < H && (long)(Wo - 1) * stride - pad_left < W,
               FD_E_INVAL, "fd_stem_conv: bad geometry");
    constexpr int S = 4;
    const long total = (long)N * Ho * ((Wo + S - 1) / S) * (Cout / 4);
    hipStream_t st = (hipStream_t)stream;
    if (K == 3 && stride == 2)
        hipLaunchKernelGGL((stem_conv_kernel<3, 2, S>), dim3(grid_for(total, 256)), dim3(256), 0, st, (const float4*)x4, w, scale, shift,
                           y, y_cs, y_co, Cout, act, H, W, Ho, Wo, pad_top, pad_left, total);
    else if (K == 3 && stride == 1)
        hipLaunchKernelGGL((stem_conv_kernel<3, 1, S>), dim3(grid_for(total, 256)), dim3(256), 0, st, (const float4*)x4, w, scale, shift,
                           y, y_cs, y_co, Cout, act, H, W, Ho, Wo, pad_top, pad_left, total);
    else { fd_set_error("fd_stem_conv: kernel %d stride %d has no kernel (3x3, stride 1 or 2)", K, stride); return FD_E_UNSUPPORTED; }
    FD_CHECK_LAUNCH("fd_stem_conv_nhwc4");
    return FD_OK;
}

// ------------------------------------------------------------------------------ mixed-aspect batch assembly
// dataset/voc.py:128-132,141-156: every resized image is zero-padded (uint8 zeros) to the next multiple of 32, then in
// collate_fn to the batch's largest (H, W) with 0. BEFORE Normalize -- so every padding pixel ends up as (0 - mean) / std, not 0.
// One launch per batch: `src` is a device array of N pointers to the RESIZED uint8 [h_n][w_n][3] images (cv2.resize is
// third-party arithmetic and stays on the host), hw = device int32 [N][2]; the output is the stem's [N][H][W][4] layout:
// ((u8 / 255) - mean) / std with u8 = 0 outside the image, channel 3 = 0.  Same fp32 op order as ToTensor + Normalize.
__global__ __launch_bounds__(256) void collate_u8_kernel(const unsigned char* const* __restrict__ src, const int* __restrict__ hw,
                                                          float4* __restrict__ y, int H, int W, float m0, float m1, float m2,
                                                          float s0, float s1, float s2, long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int wq;
        const long t = fd_div(i, W, wq);
        int hq;
        const int n = (int)fd_div(t, H, hq);
        const int h = hw[2 * n], w = hw[2 * n + 1];
        float r = 0.f, g = 0.f, b = 0.f;
        if (hq < h && wq < w) {
            const unsigned char* p = src[n] + ((long)hq * w + wq) * 3;
            r = (float)p[0]; g = (float)p[1]; b = (float)p[2];
        }
        y[i] = make_float4((r / 255.0f - m0) / s0, (g / 255.0f - m1) / s1, (b / 255.0f - m2) / s2, 0.f);
    }
}

extern "C" int32_t fd_collate_u8_nhwc4(const uint8_t* const* images_dev, const int32_t* hw_dev, float* y, int32_t N, int32_t H,
                                       int32_t W, const float* mean3, const float* std3, fd_stream_t stream) {
    FD_REQUIRE(images_dev && hw_dev && y && mean3 && std3 && N >= 1 && H >= 1 && W >= 1, FD_E_INVAL, "fd_collate_u8: bad argument");
    FD_REQUIRE(((uintptr_t)y & 15) == 0, FD_E_INVAL, "fd_collate_u8: y not 16-byte aligned");
    FD_REQUIRE(std3[0] != 0.f && std3[1] != 0.f && std3[2] != 0.f, FD_E_INVAL, "fd_collate_u8: zero std");
    const long total = (long)N * H * W;
    hipLaunchKernelGGL(collate_u8_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, images_dev, hw_dev, (float4*)y,
                       H, W, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], total);
    FD_CHECK_LAUNCH("fd_collate_u8_nhwc4");
    return FD_OK;
}

// ------------------------------------------------------------------------------ MBConv: expand 1x1 -> depthwise k x k in ONE kernel
// efficientnet_pytorch 0.7.1 MBConvBlock.forward (behind model/backbone/efficientnetv1.py:11-26): x -> _expand_conv (1x1, Cin -> 6 Cin) -> _bn0 -> swish ->
// _depthwise_conv (k in {3, 5}, stride 1 / 2, static "SAME" padding) -> _bn1 -> swish -> squeeze-excitation pooling.  As separate launches the expanded map -- six times
// the block's input, 2.6 GB for B3's first stage-2 block at 16 x 832 x 1344 -- is written by the expand conv and read back by the depthwise conv, and the SE pooling
// reads the depthwise output once more.  Here one workgroup (8 waves, one per CU) owns a TO x TO tile of the DEPTHWISE OUTPUT of one image:
//   * the (TO - 1) * stride + k square input patch (at most 16 x 16 = 256 pixels, Cin <= 48 channels) is staged in LDS once (row stride 4 * odd floats: the A-operand
//     reads are 16 bytes per lane and conflict-free);
//   * the expanded channels go through in blocks of 32 as a TWO-STAGE PIPELINE over the waves, one barrier per block:
//       waves 0-3 (matrix pipe): the expand GEMM of block cb -- 64 patch pixels per wave x 32 channels, K = Cin on v_mfma_f32_32x32x2_f32, B = the block's weights staged in
//         LDS in fragment order (16 bytes per lane; the next block's weights are fetched under the MFMAs) -- then _bn0 + swish, the pixels OUTSIDE the image set to zero (the
//         depthwise conv pads the EXPANDED map with zeros, not with swish(bn0(0))), written to the expanded tile Es[cb & 1];
//       waves 4-7 (vector unit / LDS): the depthwise conv of block cb - 1 from Es[(cb - 1) & 1]: a thread owns a channel quad (its k x k taps live in registers, fetched one
//         block ahead) and strips of adjacent outputs (stride 1: two per strip, sharing their window reads), taps in dwconv_kxk_kernel's order, then _bn1 + swish, 16-byte
//         stores, and the SE pooling: per-thread sums -> fixed xor-shuffle tree per wave -> four wave partials added in order -> pool[image][tile][channel]
//         (fd_se_gate_from_pool adds the tiles in index order: deterministic).
// The expanded map never reaches HBM; the expand GEMM is recomputed on the halo (1.15 - 1.8 x).  Blocks with Cin > 48 (LDS: patch + two expanded tiles = 147 KB at Cin = 48) stay on the separate launches.
// (First version, same results: every wave did GEMM then depthwise, three barriers per block, taps and weights re-read from LDS per output, the pooling summed by eight
//  threads over 64 dependent LDS reads: 43 us per tile where the separate launches need the equivalent of 34 -- profiles/r05_mbconv_fused_v1_layer_times.tsv.)
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct MbFusedArgs {
    const float* x; int x_cs, x_co;
    const float* we;                       // [ceil(mid / 32)][Cin / 8][2][32][4]: element (cb, g, h, l, jj) = W_expand[32 cb + l][h * (Cin / 2) + 4 g + jj] (rows >= mid zero)
    const float* sc0; const float* sf0;    // _bn0 folded, [mid]
    const float* wd;                       // depthwise weights [K * K][mid]
    const float* sc1; const float* sf1;    // _bn1 folded, [mid]
    float* y; int y_cs, y_co;              // depthwise output rows [N][Ho][Wo] x mid
    float* pool;                           // [N][tiles_y * tiles_x][mid] partial sums of y over the tile's pixels
    int N, H, W, Ho, Wo, Cin, mid, pad_t, pad_l, tiles_x, tiles_y;
};

__host__ __device__ static inline int mb_xs(int Cin) { return Cin + (((Cin >> 2) & 1) ? 0 : 4); }      // patch row stride in floats: 4 * odd

template <int K, int STRIDE>
__global__ __launch_bounds__(512, 1) void mbconv_expand_dw_kernel(MbFusedArgs a) {
    constexpr int TO = (STRIDE == 1) ? (K == 3 ? 14 : 12) : (K == 3 ? 7 : 6);      // output tile side
    constexpr int PS = (TO - 1) * STRIDE + K;                                        // patch side: 16 or 15
    constexpr int ES = 36;                                                           // floats per pixel row of the expanded tile
    constexpr int SO = (STRIDE == 1) ? 2 : 1;                                        // adjacent outputs per strip
    constexpr int WIN = (SO - 1) * STRIDE + K;                                       // window columns per filter row
    constexpr int NSX = TO / SO, NSTRIP = TO * NSX;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int XS = mb_xs(a.Cin);
    float* Xs = reinterpret_cast<float*>(smem);                                      // [256][XS]
    float* Es = Xs + 256 * XS;                                                       // [2][256][ES]
    float* Ws = Es + 2 * 256 * ES;                                                   // [2][Cin * 32]
    float* Rw = Ws + 2 * a.Cin * 32;                                                 // [2][4 waves][8 quads] float4
    float* Wd = Rw + 2 * 4 * 8 * 4;                                                  // [2][K * K][32]: depthwise taps of a block, staged one block ahead
    int* Vm = reinterpret_cast<int*>(Wd + 2 * K * K * 32);                           // [256] 1 = patch pixel inside the image
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;

    int b = blockIdx.x;
    const int tx = b % a.tiles_x; b /= a.tiles_x;
    const int ty = b % a.tiles_y;
    const int n = b / a.tiles_y;
    const int oy0 = ty * TO, ox0 = tx * TO;
    const int iy0 = oy0 * STRIDE - a.pad_t, ix0 = ox0 * STRIDE - a.pad_l;
    const int ncb = (a.mid + 31) >> 5;
    const int wblk4 = a.Cin * 8;                                 // float4s per weight block

    // ---- prologue: the input patch (256 pixel slots, PS * PS used; zero outside the image) and weight block 0 ----
    {
        const int C4 = a.Cin >> 2;
        for (int i = tid; i < 256 * C4; i += 512) {
            const int p = i / C4, q = i - p * C4;
            const int py = p / PS, px = p - py * PS;
            const int iy = iy0 + py, ix = ix0 + px;
            const bool ok = p < PS * PS && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            *reinterpret_cast<float4*>(Xs + p * XS + 4 * q) =
                ok ? *reinterpret_cast<const float4*>(a.x + ((size_t)(n * a.H + iy) * a.W + ix) * a.x_cs + a.x_co + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (q == 0) Vm[p] = ok ? 1 : 0;
        }
        for (int i = tid; i < wblk4; i += 512) reinterpret_cast<float4*>(Ws)[i] = reinterpret_cast<const float4*>(a.we)[i];
    }
    __syncthreads();

    const bool producer = wave < 4;                              // (wave-uniform)
    // producer: patch row blocks 2 wave, 2 wave + 1
    const int nkh = a.Cin >> 1, ng = nkh >> 2;                   // K steps per lane half, groups of four
    // consumer: channel quad q, strip slot
    const int ctid = tid - 256, q = ctid & 7, slot = ctid >> 3;
    float4 kw[K * K];                                            // this block's depthwise taps of quad q (from LDS, where they were staged one block ahead)
    float4 psum = make_float4(0.f, 0.f, 0.f, 0.f);
    // taps of block cbn -> Wd[cbn & 1]: K * K * 8 float4s, one per consumer thread (K = 5: 200 of the 256)
    auto fetch_taps = [&](int cbn) -> float4 {
        const int t = ctid >> 3, c = cbn * 32 + 4 * (ctid & 7);
        return (ctid < K * K * 8 && c < a.mid) ? *reinterpret_cast<const float4*>(a.wd + (size_t)t * a.mid + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    if (!producer && ctid < K * K * 8) reinterpret_cast<float4*>(Wd)[ctid] = fetch_taps(0);       // (visible behind the first loop barrier: block 0's taps are read in iteration 1)

    for (int cb = 0; cb <= ncb; ++cb) {
        if (producer) {
            if (cb < ncb) {
                const int buf = cb & 1;
                // the next block's weights: global -> registers now, -> LDS behind the MFMAs (wblk4 <= 512: at most one float4 per thread ... of the 256 producer threads: two)
                float4 wn0 = make_float4(0.f, 0.f, 0.f, 0.f), wn1 = wn0;
                const bool more = cb + 1 < ncb;
                if (more) {
                    const float4* src = reinterpret_cast<const float4*>(a.we) + (size_t)(cb + 1) * wblk4;
                    if (tid < wblk4) wn0 = src[tid];
                    if (tid + 256 < wblk4) wn1 = src[tid + 256];
                }
                f32x16 acc0, acc1;
#pragma unroll
                for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
                const float* A0 = Xs + (64 * wave + l31) * XS + lh * nkh;
                const float* A1 = A0 + 32 * XS;
                const float* Bw = Ws + buf * a.Cin * 32 + (lh * 32 + l31) * 4;
                for (int g = 0; g < ng; ++g) {
                    const float4 a0 = *reinterpret_cast<const float4*>(A0 + 4 * g), a1 = *reinterpret_cast<const float4*>(A1 + 4 * g);
                    const float4 bw = *reinterpret_cast<const float4*>(Bw + g * 256);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, bw.x, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, bw.x, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, bw.y, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, bw.y, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, bw.z, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, bw.z, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, bw.w, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, bw.w, acc1, 0, 0, 0);
                }
                const int c = cb * 32 + l31;
                const bool c_ok = c < a.mid;
                const float s0 = c_ok ? a.sc0[c] : 0.f, t0 = c_ok ? a.sf0[c] : 0.f;
                float* Eb = Es + buf * 256 * ES;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int p = 64 * wave + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    const float v0 = fd_act(fmaf(acc0[e], s0, t0), FD_ACT_SILU, 0.f), v1 = fd_act(fmaf(acc1[e], s0, t0), FD_ACT_SILU, 0.f);
                    Eb[p * ES + l31] = (Vm[p] && c_ok) ? v0 : 0.f;                   // the depthwise conv sees zeros outside the image
                    Eb[(p + 32) * ES + l31] = (Vm[p + 32] && c_ok) ? v1 : 0.f;
                }
                if (more) {
                    float4* dst = reinterpret_cast<float4*>(Ws + (buf ^ 1) * a.Cin * 32);
                    if (tid < wblk4) dst[tid] = wn0;
                    if (tid + 256 < wblk4) dst[tid + 256] = wn1;
                }
            }
        } else {
            // ---- the pooling partials of block cb - 2 (written into Rw[cb & 1] one iteration ago): four wave sums in wave order ----
            if (cb >= 2 && ctid < 8) {
                const int c = (cb - 2) * 32 + 4 * ctid;
                if (c < a.mid) {
                    const float4* r = reinterpret_cast<const float4*>(Rw) + (cb & 1) * 32 + ctid;
                    float4 s = r[0];
                    const float4 s1 = r[8], s2 = r[16], s3 = r[24];
                    s.x = ((s.x + s1.x) + s2.x) + s3.x; s.y = ((s.y + s1.y) + s2.y) + s3.y; s.z = ((s.z + s1.z) + s2.z) + s3.z; s.w = ((s.w + s1.w) + s2.w) + s3.w;
                    *reinterpret_cast<float4*>(a.pool + ((size_t)n * a.tiles_y * a.tiles_x + ty * a.tiles_x + tx) * a.mid + c) = s;
                }
            }
            if (cb >= 1) {
                const int c0 = (cb - 1) * 32;
                const int cq = c0 + 4 * q;
                const float4 tn = (cb < ncb) ? fetch_taps(cb) : make_float4(0.f, 0.f, 0.f, 0.f);      // the next block's taps travel under this block's work
#pragma unroll
                for (int t = 0; t < K * K; ++t) kw[t] = reinterpret_cast<const float4*>(Wd + ((cb - 1) & 1) * K * K * 32)[t * 8 + q];
                float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), t1 = s1;
                if (cq < a.mid) { s1 = *reinterpret_cast<const float4*>(a.sc1 + cq); t1 = *reinterpret_cast<const float4*>(a.sf1 + cq); }
                const float* Eb = Es + ((cb - 1) & 1) * 256 * ES + 4 * q;
                psum = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int st = slot; st < NSTRIP; st += 32) {
                    const int oy = st / NSX, sx = st - oy * NSX;
                    const int ox = sx * SO;
                    const float* eb = Eb + ((oy * STRIDE) * PS + ox * STRIDE) * ES;
                    float4 o[SO];
#pragma unroll
                    for (int j = 0; j < SO; ++j) o[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int r = 0; r < K; ++r) {
                        float4 u[WIN];
#pragma unroll
                        for (int c = 0; c < WIN; ++c) u[c] = *reinterpret_cast<const float4*>(eb + (r * PS + c) * ES);
#pragma unroll
                        for (int c = 0; c < K; ++c) {
                            const float4 k = kw[r * K + c];
#pragma unroll
                            for (int j = 0; j < SO; ++j) {
                                const float4 v = u[j * STRIDE + c];
                                o[j].x = fmaf(v.x, k.x, o[j].x); o[j].y = fmaf(v.y, k.y, o[j].y); o[j].z = fmaf(v.z, k.z, o[j].z); o[j].w = fmaf(v.w, k.w, o[j].w);
                            }
                        }
                    }
                    const int gy = oy0 + oy;
#pragma unroll
                    for (int j = 0; j < SO; ++j) {
                        const int gx = ox0 + ox + j;
                        if (gy < a.Ho && gx < a.Wo && cq < a.mid) {
                            float4 v;
                            v.x = fd_act(fmaf(o[j].x, s1.x, t1.x), FD_ACT_SILU, 0.f); v.y = fd_act(fmaf(o[j].y, s1.y, t1.y), FD_ACT_SILU, 0.f);
                            v.z = fd_act(fmaf(o[j].z, s1.z, t1.z), FD_ACT_SILU, 0.f); v.w = fd_act(fmaf(o[j].w, s1.w, t1.w), FD_ACT_SILU, 0.f);
                            *reinterpret_cast<float4*>(a.y + ((size_t)(n * a.Ho + gy) * a.Wo + gx) * a.y_cs + a.y_co + cq) = v;
                            psum.x += v.x; psum.y += v.y; psum.z += v.z; psum.w += v.w;
                        }
                    }
                }
                // lanes q, q + 8, .., q + 56 of a wave hold channel quad q: a fixed xor tree (8, 16, 32), then one slot per (wave, quad)
#pragma unroll
                for (int off = 8; off < 64; off <<= 1) {
                    psum.x += __shfl_xor(psum.x, off); psum.y += __shfl_xor(psum.y, off); psum.z += __shfl_xor(psum.z, off); psum.w += __shfl_xor(psum.w, off);
                }
                if (lane < 8) reinterpret_cast<float4*>(Rw)[((cb - 1) & 1) * 32 + (wave - 4) * 8 + lane] = psum;
                if (cb < ncb && ctid < K * K * 8) reinterpret_cast<float4*>(Wd + (cb & 1) * K * K * 32)[ctid] = tn;
            }
        }
        __syncthreads();
    }
    // the last block's pooling partials (written in the final iteration, behind the loop's last barrier)
    if (ctid >= 0 && ctid < 8) {
        const int c = (ncb - 1) * 32 + 4 * ctid;
        if (c < a.mid) {
            const float4* r = reinterpret_cast<const float4*>(Rw) + ((ncb - 1) & 1) * 32 + ctid;
            float4 s = r[0];
            const float4 s1 = r[8], s2 = r[16], s3 = r[24];
            s.x = ((s.x + s1.x) + s2.x) + s3.x; s.y = ((s.y + s1.y) + s2.y) + s3.y; s.z = ((s.z + s1.z) + s2.z) + s3.z; s.w = ((s.w + s1.w) + s2.w) + s3.w;
            *reinterpret_cast<float4*>(a.pool + ((size_t)n * a.tiles_y * a.tiles_x + ty * a.tiles_x + tx) * a.mid + c) = s;
        }
    }
}

static int mb_tile_side(int K, int stride) { return stride == 1 ? (K == 3 ? 14 : 12) : (K == 3 ? 7 : 6); }

extern "C" int64_t fd_mbconv_pool_bytes(int32_t N, int32_t Ho, int32_t Wo, int32_t mid, int32_t K, int32_t stride) {
    if (N < 1 || Ho < 1 || Wo < 1 || mid < 4 || (K != 3 && K != 5) || (stride != 1 && stride != 2)) return -1;
    const int to = mb_tile_side(K, stride);
    return (int64_t)N * ((Ho + to - 1) / to) * ((Wo + to - 1) / to) * mid * 4;
}

extern "C" int32_t fd_mbconv_expand_dw_nhwc(const float* x, int32_t x_cs, int32_t x_co, const float* w_expand_frag, const float* scale0, const float* shift0,
                                            const float* w_dw, const float* scale1, const float* shift1, float* y, int32_t y_cs, int32_t y_co, float* pool,
                                            int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t mid, int32_t K, int32_t stride, int32_t pad_top, int32_t pad_left,
                                            int32_t Ho, int32_t Wo, fd_stream_t stream) {
    FD_REQUIRE(view_ok(x, x_cs, x_co, Cin) && view_ok(y, y_cs, y_co, mid) && w_expand_frag && scale0 && shift0 && w_dw && scale1 && shift1 && pool &&
                   ((((uintptr_t)w_expand_frag | (uintptr_t)scale1 | (uintptr_t)shift1 | (uintptr_t)pool) & 15) == 0), FD_E_INVAL,
               "fd_mbconv_expand_dw: bad pointer / channel view (Cin=%d mid=%d)", Cin, mid);
    FD_REQUIRE((K == 3 || K == 5) && (stride == 1 || stride == 2) && Cin >= 8 && Cin <= 48 && Cin % 8 == 0 && mid % 4 == 0, FD_E_UNSUPPORTED,
               "fd_mbconv_expand_dw: k in {3, 5}, stride in {1, 2}, Cin %% 8 == 0 in 8 .. 48, mid %% 4 == 0 (got k=%d s=%d Cin=%d mid=%d)", K, stride, Cin, mid);
    FD_REQUIRE(N >= 1 && H >= 1 && W >= 1 && Ho >= 1 && Wo >= 1 && pad_top >= 0 && pad_left >= 0 && pad_top < K && pad_left < K &&
                   (long)(Ho - 1) * stride - pad_top < H && (long)(Wo - 1) * stride - pad_left < W, FD_E_INVAL, "fd_mbconv_expand_dw: bad geometry");
    FD_REQUIRE((long)N * H * W * x_cs < (1L << 31) && (long)N * Ho * Wo * y_cs < (1L << 31), FD_E_UNSUPPORTED, "fd_mbconv_expand_dw: tensor too large");
    MbFusedArgs a;
    a.x = x; a.x_cs = x_cs; a.x_co = x_co; a.we = w_expand_frag; a.sc0 = scale0; a.sf0 = shift0; a.wd = w_dw; a.sc1 = scale1; a.sf1 = shift1;
    a.y = y; a.y_cs = y_cs; a.y_co = y_co; a.pool = pool;
    a.N = N; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.Cin = Cin; a.mid = mid; a.pad_t = pad_top; a.pad_l = pad_left;
    const int to = mb_tile_side(K, stride);
    a.tiles_y = (Ho + to - 1) / to; a.tiles_x = (Wo + to - 1) / to;
    const long blocks = (long)N * a.tiles_y * a.tiles_x;
    FD_REQUIRE(blocks < (1L << 31), FD_E_UNSUPPORTED, "fd_mbconv_expand_dw: too many tiles");
    const int lds = (256 * mb_xs(Cin) + 2 * 256 * 36 + 2 * Cin * 32 + 2 * 4 * 8 * 4 + 2 * K * K * 32 + 256) * 4;       // patch + two expanded tiles + two weight blocks + pooling slots + taps + validity: <= 147 KB
    hipStream_t st = (hipStream_t)stream;
#define FD_MB_LAUNCH(KK, SS)                                                                                                   \
    do {                                                                                                                      \
        static std::atomic<unsigned> m{0};                                                                                    \
        fd_set_max_lds_once(m, reinterpret_cast<const void*>(mbconv_expand_dw_kernel<KK, SS>), 160 * 1024);                   \
        hipLaunchKernelGGL((mbconv_expand_dw_kernel<KK, SS>), dim3((unsigned)blocks), dim3(512), lds, st, a);                  \
    } while (0)
    if (K == 3 && stride == 1) FD_MB_LAUNCH(3, 1);
    else if (K == 3) FD_MB_LAUNCH(3, 2);
    else if (stride == 1) FD_MB_LAUNCH(5, 1);
    else FD_MB_LAUNCH(5, 2);
#undef FD_MB_LAUNCH
    FD_CHECK_LAUNCH("fd_mbconv_expand_dw_nhwc");
    return FD_OK;
}

