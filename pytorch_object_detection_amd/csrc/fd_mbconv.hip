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
            o.x = fd_act(acc[j].x * sc.x + sf.x, act, 0.f); o.y = fd_act(acc[j].y * sc.y + sf.y, act, 0.f);
            o.z = fd_act(acc[j].z * sc.z + sf.z, act, 0.f); o.w = fd_act(acc[j].w * sc.w + sf.w, act, 0.f);
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
            o.x = fd_act(acc[j].x * sc.x + sf.x, act, 0.f); o.y = fd_act(acc[j].y * sc.y + sf.y, act, 0.f);
            o.z = fd_act(acc[j].z * sc.z + sf.z, act, 0.f); o.w = fd_act(acc[j].w * sc.w + sf.w, act, 0.f);
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
