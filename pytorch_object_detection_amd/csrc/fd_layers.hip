// fd_layers.hip — HBM-bound layer ops of the FCOS / HISFCOS stack on NHWC fp32 rows (gfx950).
// All kernels move 16 bytes per lane (float4 over channels), rows x channel-quads flattened so that a wave
// reads/writes contiguous 1 KiB spans.  See include/fcosdet.h for the reference call sites each replaces.
#include "fd_common.h"

#define FD_GRID_CAP 16384

static inline unsigned grid_for(long work, int block) {
    long g = (work + block - 1) / block;
    if (g > FD_GRID_CAP) g = FD_GRID_CAP;
    if (g < 1) g = 1;
    return (unsigned)g;
}

// ------------------------------------------------------------------------------ layout converters
__global__ __launch_bounds__(256) void nchw3_to_nhwc4_kernel(const float* __restrict__ x, float4* __restrict__ y,
                                                              int HW, long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int pi;
        const long n = fd_div(i, HW, pi);
        const long p = pi;
        const float* b = x + n * 3 * HW + p;
        y[i] = make_float4(b[0], b[HW], b[2 * (long)HW], 0.f);
    }
}

extern "C" int32_t fd_nchw3_to_nhwc4(const float* x, float* y, int32_t N, int32_t H, int32_t W, fd_stream_t stream) {
    FD_REQUIRE(x && y && N >= 1 && H >= 1 && W >= 1, FD_E_INVAL, "fd_nchw3_to_nhwc4: bad argument");
    FD_REQUIRE(((uintptr_t)y & 15) == 0, FD_E_INVAL, "fd_nchw3_to_nhwc4: y not 16-byte aligned");
    const long total = (long)N * H * W;
    hipLaunchKernelGGL(nchw3_to_nhwc4_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x,
                       (float4*)y, H * W, total);
    FD_CHECK_LAUNCH("fd_nchw3_to_nhwc4");
    return FD_OK;
}

// uint8 HWC image (already resized + zero padded, dataset/voc.py:110-139) -> normalised fp32 [N][H][W][4]:
// ToTensor (u8 / 255) then Normalize ((v - mean) / std), voc.py:57-58,104,155; channel 3 = 0.  Same fp32 op order.
__global__ __launch_bounds__(256) void preprocess_u8_kernel(const unsigned char* __restrict__ x, float4* __restrict__ y,
                                                             float m0, float m1, float m2, float s0, float s1, float s2,
                                                             long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const unsigned char* p = x + 3 * i;
        y[i] = make_float4(((float)p[0] / 255.0f - m0) / s0, ((float)p[1] / 255.0f - m1) / s1,
                           ((float)p[2] / 255.0f - m2) / s2, 0.f);
    }
}

extern "C" int32_t fd_preprocess_u8_nhwc4(const uint8_t* x, float* y, int32_t N, int32_t H, int32_t W, const float* mean3,
                                          const float* std3, fd_stream_t stream) {
    FD_REQUIRE(x && y && mean3 && std3 && N >= 1 && H >= 1 && W >= 1, FD_E_INVAL, "fd_preprocess_u8_nhwc4: bad argument");
    FD_REQUIRE(((uintptr_t)y & 15) == 0, FD_E_INVAL, "fd_preprocess_u8_nhwc4: y not 16-byte aligned");
    FD_REQUIRE(std3[0] != 0.f && std3[1] != 0.f && std3[2] != 0.f, FD_E_INVAL, "fd_preprocess_u8_nhwc4: zero std");
    const long total = (long)N * H * W;
    hipLaunchKernelGGL(preprocess_u8_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, (float4*)y,
                       mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], total);
    FD_CHECK_LAUNCH("fd_preprocess_u8_nhwc4");
    return FD_OK;
}

// detections back to source-image scale, xyxy -> xywh (Test_coco.py:147-151): boxes /= scale; w = x2 - x1; h = y2 - y1
__global__ __launch_bounds__(256) void boxes_to_xywh_kernel(float4* boxes, long n, float scale) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float4 b = boxes[i];
    b.x = b.x / scale; b.y = b.y / scale; b.z = b.z / scale; b.w = b.w / scale;
    b.z = b.z - b.x; b.w = b.w - b.y;
    boxes[i] = b;
}

extern "C" int32_t fd_boxes_rescale_xywh(float* boxes, int64_t n_boxes, float scale, fd_stream_t stream) {
    FD_REQUIRE(boxes && ((uintptr_t)boxes & 15) == 0 && scale != 0.f, FD_E_INVAL, "fd_boxes_rescale_xywh: bad argument");
    if (n_boxes <= 0) return FD_OK;
    hipLaunchKernelGGL(boxes_to_xywh_kernel, dim3((unsigned)((n_boxes + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (float4*)boxes, (long)n_boxes, scale);
    FD_CHECK_LAUNCH("fd_boxes_rescale_xywh");
    return FD_OK;
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ x, int x_cs, int x_co,
                                                            float* __restrict__ y, int HW, int C) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int p = p0 + r, c = c0 + tx;
        tile[r][tx] = (p < HW && c < C) ? x[((long)n * HW + p) * x_cs + x_co + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, p = p0 + tx;
        if (c < C && p < HW) y[((long)n * C + c) * HW + p] = tile[tx][r];
    }
}

extern "C" int32_t fd_nhwc_to_nchw(const float* x, int32_t x_cs, int32_t x_co, float* y, int32_t N, int32_t HW,
                                   int32_t C, fd_stream_t stream) {
    FD_REQUIRE(x && y && N >= 1 && HW >= 1 && C >= 1 && N <= 65535, FD_E_INVAL, "fd_nhwc_to_nchw: bad argument");
    dim3 grid((HW + 31) / 32, (C + 31) / 32, N);
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, x_cs, x_co, y, HW, C);
    FD_CHECK_LAUNCH("fd_nhwc_to_nchw");
    return FD_OK;
}

// ------------------------------------------------------------------------------ elementwise activation (+ backward)
// y = act(x) on a channel view; act in {RELU, SILU, EXP (exp(x * param), ScaleExp of modules.py:170-176), SIGMOID}.
// Backward: dx = dy * act'(x) from the saved INPUT (SiLU needs the pre-activation; ReLU / EXP / SIGMOID could use the
// output, the input form keeps one code path).
__device__ __forceinline__ float fd_act_deriv(float x, int act, float p) {
    switch (act) {
        case FD_ACT_RELU: return x > 0.f ? 1.f : 0.f;
        case FD_ACT_SILU: { const float sg = fd_sigmoid(x); return sg * (1.f + x * (1.f - sg)); }
        case FD_ACT_EXP: return p * expf(x * p);
        case FD_ACT_SIGMOID: { const float sg = fd_sigmoid(x); return sg * (1.f - sg); }
        default: return 1.f;
    }
}

__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, int x_cs, int x_co, float* __restrict__ y, int y_cs,
                                                       int y_co, int C4, int act, float prm, long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int q;
        const long m = fd_div(i, C4, q);
        const float4 v = *reinterpret_cast<const float4*>(x + m * x_cs + x_co + 4 * q);
        *reinterpret_cast<float4*>(y + m * y_cs + y_co + 4 * q) =
            make_float4(fd_act(v.x, act, prm), fd_act(v.y, act, prm), fd_act(v.z, act, prm), fd_act(v.w, act, prm));
    }
}

__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ x, int x_cs, int x_co, const float* __restrict__ dy,
                                                       int dy_cs, int dy_co, float* __restrict__ dx, int dx_cs, int dx_co, int C4,
                                                       int act, float prm, long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int q;
        const long m = fd_div(i, C4, q);
        const float4 v = *reinterpret_cast<const float4*>(x + m * x_cs + x_co + 4 * q);
        const float4 g = *reinterpret_cast<const float4*>(dy + m * dy_cs + dy_co + 4 * q);
        *reinterpret_cast<float4*>(dx + m * dx_cs + dx_co + 4 * q) =
            make_float4(g.x * fd_act_deriv(v.x, act, prm), g.y * fd_act_deriv(v.y, act, prm), g.z * fd_act_deriv(v.z, act, prm),
                        g.w * fd_act_deriv(v.w, act, prm));
    }
}

static inline bool view_ok(const void* p, int cs, int co, int C);

// the same pass over f16 maps (AMP activations / gradients stored as f16): 8-byte accesses, the derivative evaluated in fp32, one rounding of the product
__global__ __launch_bounds__(256) void act_bwd_h_kernel(const _Float16* __restrict__ x, int x_cs, int x_co, const _Float16* __restrict__ dy,
                                                         int dy_cs, int dy_co, _Float16* __restrict__ dx, int dx_cs, int dx_co, int C4,
                                                         int act, float prm, long total) {
    typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int q;
        const long m = fd_div(i, C4, q);
        const h4_t v = *reinterpret_cast<const h4_t*>(x + m * x_cs + x_co + 4 * q);
        const h4_t g = *reinterpret_cast<const h4_t*>(dy + m * dy_cs + dy_co + 4 * q);
        h4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (_Float16)((float)g[e] * fd_act_deriv((float)v[e], act, prm));
        *reinterpret_cast<h4_t*>(dx + m * dx_cs + dx_co + 4 * q) = o;
    }
}

extern "C" int32_t fd_act_bwd_nhwc_h(const void* x, int32_t x_cs, int32_t x_co, const void* dy, int32_t dy_cs, int32_t dy_co, void* dx,
                                     int32_t dx_cs, int32_t dx_co, int64_t rows, int32_t C, int32_t act, float param, fd_stream_t stream) {
    auto ok = [](const void* p, int cs, int co, int C_) { return p && C_ >= 4 && C_ % 4 == 0 && cs % 4 == 0 && co % 4 == 0 && cs >= co + C_ && ((uintptr_t)p & 7) == 0; };
    FD_REQUIRE(ok(x, x_cs, x_co, C) && ok(dy, dy_cs, dy_co, C) && ok(dx, dx_cs, dx_co, C) && rows >= 1, FD_E_INVAL, "fd_act_bwd_h: bad pointer / channel view (C=%d)", C);
    FD_REQUIRE(act >= FD_ACT_NONE && act <= FD_ACT_SIGMOID, FD_E_INVAL, "fd_act_bwd_h: unknown activation %d", act);
    const long total = (long)rows * (C / 4);
    hipLaunchKernelGGL(act_bwd_h_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)x, x_cs, x_co, (const _Float16*)dy, dy_cs, dy_co,
                       (_Float16*)dx, dx_cs, dx_co, C / 4, act, param, total);
    FD_CHECK_LAUNCH("fd_act_bwd_nhwc_h");
    return FD_OK;
}

extern "C" int32_t fd_act_nhwc(const float* x, int32_t x_cs, int32_t x_co, float* y, int32_t y_cs, int32_t y_co, int64_t rows,
                               int32_t C, int32_t act, float param, fd_stream_t stream) {
    FD_REQUIRE(view_ok(x, x_cs, x_co, C) && view_ok(y, y_cs, y_co, C) && rows >= 1, FD_E_INVAL, "fd_act: bad pointer / channel view (C=%d)", C);
    FD_REQUIRE(act >= FD_ACT_NONE && act <= FD_ACT_SIGMOID, FD_E_INVAL, "fd_act: unknown activation %d", act);
    const long total = (long)rows * (C / 4);
    hipLaunchKernelGGL(act_fwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, x_cs, x_co, y, y_cs, y_co, C / 4,
                       act, param, total);
    FD_CHECK_LAUNCH("fd_act_nhwc");
    return FD_OK;
}

extern "C" int32_t fd_act_bwd_nhwc(const float* x, int32_t x_cs, int32_t x_co, const float* dy, int32_t dy_cs, int32_t dy_co, float* dx,
                                   int32_t dx_cs, int32_t dx_co, int64_t rows, int32_t C, int32_t act, float param, fd_stream_t stream) {
    FD_REQUIRE(view_ok(x, x_cs, x_co, C) && view_ok(dy, dy_cs, dy_co, C) && view_ok(dx, dx_cs, dx_co, C) && rows >= 1, FD_E_INVAL,
               "fd_act_bwd: bad pointer / channel view (C=%d)", C);
    FD_REQUIRE(act >= FD_ACT_NONE && act <= FD_ACT_SIGMOID, FD_E_INVAL, "fd_act_bwd: unknown activation %d", act);
    const long total = (long)rows * (C / 4);
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, x_cs, x_co, dy, dy_cs, dy_co, dx,
                       dx_cs, dx_co, C / 4, act, param, total);
    FD_CHECK_LAUNCH("fd_act_bwd_nhwc");
    return FD_OK;
}

// ------------------------------------------------------------------------------ max-pool (+ add)
__global__ __launch_bounds__(256) void maxpool_kernel(const float* __restrict__ x, int x_cs, int x_co,
                                                       float* __restrict__ y, int y_cs, int y_co,
                                                       const float* __restrict__ add, int add_cs, int add_co, int H,
                                                       int W, int Ho, int Wo, int C4, int k, int s, int pad, long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int q;
        const long m = fd_div(i, C4, q);
        int wo;
        const long t = fd_div(m, Wo, wo);
        int ho;
        const long n = fd_div(t, Ho, ho);
        float4 v = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        for (int r = 0; r < k; ++r) {
            const int hi = ho * s - pad + r;
            if ((unsigned)hi >= (unsigned)H) continue;
            for (int c = 0; c < k; ++c) {
                const int wi = wo * s - pad + c;
                if ((unsigned)wi >= (unsigned)W) continue;
                const float4 u = *reinterpret_cast<const float4*>(x + ((n * H + hi) * W + wi) * x_cs + x_co + 4 * q);
                v.x = fmaxf(v.x, u.x); v.y = fmaxf(v.y, u.y); v.z = fmaxf(v.z, u.z); v.w = fmaxf(v.w, u.w);
            }
        }
        if (add) {
            const float4 u = *reinterpret_cast<const float4*>(add + m * add_cs + add_co + 4 * q);
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        *reinterpret_cast<float4*>(y + m * y_cs + y_co + 4 * q) = v;
    }
}

static inline bool view_ok(const void* p, int cs, int co, int C) {
    return p && C % 4 == 0 && cs % 4 == 0 && co % 4 == 0 && cs >= co + C && ((uintptr_t)p & 15) == 0;
}

extern "C" int32_t fd_maxpool_nhwc(const float* x, int32_t x_cs, int32_t x_co, float* y, int32_t y_cs, int32_t y_co,
                                   const float* add, int32_t add_cs, int32_t add_co, int32_t N, int32_t H, int32_t W,
                                   int32_t C, int32_t k, int32_t s, int32_t pad, fd_stream_t stream) {
    FD_REQUIRE(view_ok(x, x_cs, x_co, C) && view_ok(y, y_cs, y_co, C), FD_E_INVAL,
               "fd_maxpool: channel views must be 4-aligned (C=%d)", C);
    FD_REQUIRE(!add || view_ok(add, add_cs, add_co, C), FD_E_INVAL, "fd_maxpool: bad add view");
    FD_REQUIRE(N >= 1 && k >= 1 && s >= 1 && pad >= 0 && 2 * pad <= k, FD_E_INVAL, "fd_maxpool: bad geometry");
    const int Ho = (H + 2 * pad - k) / s + 1, Wo = (W + 2 * pad - k) / s + 1;
    FD_REQUIRE(Ho >= 1 && Wo >= 1, FD_E_INVAL, "fd_maxpool: empty output");
    const long total = (long)N * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(maxpool_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, x_cs, x_co, y,
                       y_cs, y_co, add, add_cs, add_co, H, W, Ho, Wo, C / 4, k, s, pad, total);
    FD_CHECK_LAUNCH("fd_maxpool_nhwc");
    return FD_OK;
}

// ------------------------------------------------------------------------------ nearest x2 upsample + add
__global__ __launch_bounds__(256) void upsample2x_add_kernel(const float* __restrict__ x, int x_cs, int x_co,
                                                              const float* __restrict__ lat, int lat_cs, int lat_co,
                                                              float* __restrict__ y, int y_cs, int y_co, int H, int W,
                                                              int C4, long total) {
    const int Ho = 2 * H, Wo = 2 * W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int q;
        const long m = fd_div(i, C4, q);
        int wo;
        const long t = fd_div(m, Wo, wo);
        int ho;
        const long n = fd_div(t, Ho, ho);
        const float4 u = *reinterpret_cast<const float4*>(x + ((n * H + (ho >> 1)) * W + (wo >> 1)) * x_cs + x_co + 4 * q);
        const float4 l = *reinterpret_cast<const float4*>(lat + m * lat_cs + lat_co + 4 * q);
        *reinterpret_cast<float4*>(y + m * y_cs + y_co + 4 * q) = make_float4(u.x + l.x, u.y + l.y, u.z + l.z, u.w + l.w);
    }
}

extern "C" int32_t fd_upsample2x_add_nhwc(const float* x, int32_t x_cs, int32_t x_co, const float* lat, int32_t lat_cs,
                                          int32_t lat_co, float* y, int32_t y_cs, int32_t y_co, int32_t N, int32_t H,
                                          int32_t W, int32_t C, fd_stream_t stream) {
    FD_REQUIRE(view_ok(x, x_cs, x_co, C) && view_ok(lat, lat_cs, lat_co, C) && view_ok(y, y_cs, y_co, C), FD_E_INVAL,
               "fd_upsample2x_add: channel views must be 4-aligned (C=%d)", C);
    FD_REQUIRE(N >= 1 && H >= 1 && W >= 1, FD_E_INVAL, "fd_upsample2x_add: bad geometry");
    const long total = (long)N * 4 * H * W * (C / 4);
    hipLaunchKernelGGL(upsample2x_add_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, x_cs,
                       x_co, lat, lat_cs, lat_co, y, y_cs, y_co, H, W, C / 4, total);
    FD_CHECK_LAUNCH("fd_upsample2x_add_nhwc");
    return FD_OK;
}

// ---- backward of max-pool: dx[pixel] = sum of dy over the windows whose FIRST maximum (row-major scan, strict >: the
// element torch's max_pool2d keeps as argmax) is that pixel.  Gather form, no atomics: deterministic.  The fused "+ add"
// of the forward passes its gradient through unchanged (dy itself).
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ x, int x_cs, int x_co, const float* __restrict__ dy,
                                                           int dy_cs, int dy_co, float* __restrict__ dx, int dx_cs, int dx_co, int H,
                                                           int W, int Ho, int Wo, int C4, int k, int s, int pad, long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int q;
        const long m = fd_div(i, C4, q);
        int wi;
        const long t = fd_div(m, W, wi);
        int hi;
        const long n = fd_div(t, H, hi);
        const float* xb = x + (n * H * (long)W) * x_cs + x_co + 4 * q;
        const float4 me = *reinterpret_cast<const float4*>(xb + ((long)hi * W + wi) * x_cs);
        float g[4] = {0.f, 0.f, 0.f, 0.f};
        const int ho_lo = max(0, (hi + pad - k + s) / s), ho_hi = min(Ho - 1, (hi + pad) / s);
        const int wo_lo = max(0, (wi + pad - k + s) / s), wo_hi = min(Wo - 1, (wi + pad) / s);
        for (int ho = ho_lo; ho <= ho_hi; ++ho)
            for (int wo = wo_lo; wo <= wo_hi; ++wo) {
                // is (hi, wi) the first maximum of window (ho, wo)?  per channel
                bool first[4] = {true, true, true, true};
                for (int r = 0; r < k; ++r) {
                    const int h2 = ho * s - pad + r;
                    if ((unsigned)h2 >= (unsigned)H) continue;
                    for (int c = 0; c < k; ++c) {
                        const int w2 = wo * s - pad + c;
                        if ((unsigned)w2 >= (unsigned)W || (h2 == hi && w2 == wi)) continue;
                        const float4 u = *reinterpret_cast<const float4*>(xb + ((long)h2 * W + w2) * x_cs);
                        const bool before = h2 < hi || (h2 == hi && w2 < wi);   // scanned earlier: wins ties
                        first[0] = first[0] && (before ? me.x > u.x : me.x >= u.x);
                        first[1] = first[1] && (before ? me.y > u.y : me.y >= u.y);
                        first[2] = first[2] && (before ? me.z > u.z : me.z >= u.z);
                        first[3] = first[3] && (before ? me.w > u.w : me.w >= u.w);
                    }
                }
                const float4 d = *reinterpret_cast<const float4*>(dy + (((n * Ho + ho) * (long)Wo) + wo) * dy_cs + dy_co + 4 * q);
                if (first[0]) g[0] += d.x;
                if (first[1]) g[1] += d.y;
                if (first[2]) g[2] += d.z;
                if (first[3]) g[3] += d.w;
            }
        *reinterpret_cast<float4*>(dx + m * dx_cs + dx_co + 4 * q) = make_float4(g[0], g[1], g[2], g[3]);
    }
}

static inline bool view_ok(const void* p, int cs, int co, int C);

extern "C" int32_t fd_maxpool_bwd_nhwc(const float* x, int32_t x_cs, int32_t x_co, const float* dy, int32_t dy_cs, int32_t dy_co,
                                       float* dx, int32_t dx_cs, int32_t dx_co, int32_t N, int32_t H, int32_t W, int32_t C, int32_t k,
                                       int32_t s, int32_t pad, fd_stream_t stream) {
    FD_REQUIRE(view_ok(x, x_cs, x_co, C) && view_ok(dy, dy_cs, dy_co, C) && view_ok(dx, dx_cs, dx_co, C), FD_E_INVAL,
               "fd_maxpool_bwd: channel views must be 4-aligned (C=%d)", C);
    FD_REQUIRE(N >= 1 && k >= 1 && s >= 1 && pad >= 0 && 2 * pad <= k, FD_E_INVAL, "fd_maxpool_bwd: bad geometry");
    const int Ho = (H + 2 * pad - k) / s + 1, Wo = (W + 2 * pad - k) / s + 1;
    FD_REQUIRE(Ho >= 1 && Wo >= 1, FD_E_INVAL, "fd_maxpool_bwd: empty output");
    const long total = (long)N * H * W * (C / 4);
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, x_cs, x_co, dy, dy_cs, dy_co,
                       dx, dx_cs, dx_co, H, W, Ho, Wo, C / 4, k, s, pad, total);
    FD_CHECK_LAUNCH("fd_maxpool_bwd_nhwc");
    return FD_OK;
}

// ---- backward of nearest x2 upsample: dx[h][w] = dy[2h][2w] + dy[2h][2w+1] + dy[2h+1][2w] + dy[2h+1][2w+1] (fixed order)
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const float* __restrict__ dy, int dy_cs, int dy_co, float* __restrict__ dx,
                                                              int dx_cs, int dx_co, int H, int W, int C4, long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int q;
        const long m = fd_div(i, C4, q);
        int w;
        const long t = fd_div(m, W, w);
        int h;
        const long n = fd_div(t, H, h);
        const float* b = dy + ((n * 2 * H + 2 * h) * (long)(2 * W) + 2 * w) * dy_cs + dy_co + 4 * q;
        const float4 a0 = *reinterpret_cast<const float4*>(b), a1 = *reinterpret_cast<const float4*>(b + dy_cs);
        const float4 a2 = *reinterpret_cast<const float4*>(b + (long)2 * W * dy_cs), a3 = *reinterpret_cast<const float4*>(b + (long)(2 * W + 1) * dy_cs);
        *reinterpret_cast<float4*>(dx + m * dx_cs + dx_co + 4 * q) =
            make_float4(((a0.x + a1.x) + a2.x) + a3.x, ((a0.y + a1.y) + a2.y) + a3.y, ((a0.z + a1.z) + a2.z) + a3.z, ((a0.w + a1.w) + a2.w) + a3.w);
    }
}

extern "C" int32_t fd_upsample2x_bwd_nhwc(const float* dy, int32_t dy_cs, int32_t dy_co, float* dx, int32_t dx_cs, int32_t dx_co,
                                          int32_t N, int32_t H, int32_t W, int32_t C, fd_stream_t stream) {
    FD_REQUIRE(view_ok(dy, dy_cs, dy_co, C) && view_ok(dx, dx_cs, dx_co, C) && N >= 1 && H >= 1 && W >= 1, FD_E_INVAL,
               "fd_upsample2x_bwd: bad pointer / channel view (C=%d)", C);
    const long total = (long)N * H * W * (C / 4);
    hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, dy, dy_cs, dy_co, dx, dx_cs,
                       dx_co, H, W, C / 4, total);
    FD_CHECK_LAUNCH("fd_upsample2x_bwd_nhwc");
    return FD_OK;
}

// ------------------------------------------------------------------------------ depthwise 3x3 (+scale/shift/act)
struct SegTab { fd_segs s; };

__device__ __forceinline__ void seg_decode(const fd_segs& sg, long m, int& H, int& W, long& img_row0, int& h, int& w) {
    int s = 0;
#pragma unroll
    for (int t = 1; t < FD_MAX_SEG; ++t)
        if (t < sg.nseg && m >= sg.m_start[t]) s = t;
    H = sg.H[s]; W = sg.W[s];
    const long local = m - sg.m_start[s];
    const int hw = H * W;
    int rem;
    const long n = fd_div(local, hw, rem);
    h = rem / W; w = rem - h * W;
    img_row0 = sg.m_start[s] + n * hw;
}

// Strip-mined: one thread produces S adjacent output pixels of a row for one channel quad, so the 3 x (S+2) input window
// is loaded once (4.5 input loads per output at S = 4, 3.75 at S = 8, instead of 9) and the 9 weights once per S outputs:
// a one-pixel-per-thread kernel is bound by the vector loads the CU's address unit can issue (2.1 TB/s), not by HBM.
struct StripTab {
    fd_segs s;
    long strip_start[FD_MAX_SEG + 1];   // first strip of each level
    int spr[FD_MAX_SEG];                // strips per image row = ceil(W / S)
};

__device__ __forceinline__ float dw_quad_xor1(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)); }
__device__ __forceinline__ float dw_quad_xor2(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)); }

// FUSE: GroupNorm around the conv (HISFCOSHead: pw1 -> GN1 -> ReLU -> dw1 -> GN2 -> SiLU -> pw2).  in_coef [levels * batch][2][C] = the
// preceding GroupNorm's per-(level, image, channel) affine (a, b): every in-image input pixel is read as in_act(x * a + b) -- the zero
// padding stays zero, the reference pads the NORMALISED map -- and gn_stats[m][g] receives the (sum, sum of squares) of the gn_cg
// channels of group g of output pixel m (fp32 float2, fixed order: the cg / 4 adjacent threads of a pixel are combined by xor-shuffles).
template <int S, bool FUSE = false>
__global__ __launch_bounds__(256) void dwconv3x3_strip_kernel(const float* __restrict__ x, int x_cs, int x_co,
                                                               const float* __restrict__ wt, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, float* __restrict__ y, int y_cs,
                                                               int y_co, int C, int act, StripTab tab, long total,
                                                               const float* __restrict__ in_coef = nullptr, int in_act = 0,
                                                               float* __restrict__ gn_stats = nullptr, int gn_G = 1) {
    const int C4 = C >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int q;
        const long t = fd_div(i, C4, q);
        int s = 0;
#pragma unroll
        for (int k = 1; k < FD_MAX_SEG; ++k)
            if (k < tab.s.nseg && t >= tab.strip_start[k]) s = k;
        const int H = tab.s.H[s], W = tab.s.W[s], spr = tab.spr[s];
        const long local = t - tab.strip_start[s];
        const int per_img = H * spr;
        int rem;
        const long n = fd_div(local, per_img, rem);
        const int h = rem / spr, w0 = (rem - h * spr) * S;
        const long r0 = (long)tab.s.m_start[s] + n * H * W;
        float4 ca = make_float4(1.f, 1.f, 1.f, 1.f), cb = make_float4(0.f, 0.f, 0.f, 0.f);
        const float act_lo = (FUSE && in_act == FD_ACT_RELU) ? 0.f : -INFINITY;
        if (FUSE && in_coef) {
            const float* cp = in_coef + ((long)(s * tab.s.batch + (int)n) * 2) * C + 4 * q;
            ca = *reinterpret_cast<const float4*>(cp);
            cb = *reinterpret_cast<const float4*>(cp + C);
        }
        float4 acc[S];
#pragma unroll
        for (int j = 0; j < S; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int hi = h + r - 1;
            if ((unsigned)hi >= (unsigned)H) continue;
            float4 u[S + 2];
#pragma unroll
            for (int c = 0; c < S + 2; ++c) {
                const int wi = w0 + c - 1;
                u[c] = (unsigned)wi < (unsigned)W ? *reinterpret_cast<const float4*>(x + (r0 + (long)hi * W + wi) * x_cs + x_co + 4 * q)
                                                  : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (FUSE && in_coef) {
                // all loads of the row are in flight before the first of them is touched; the affine + ReLU is branch-free (a lower bound of
                // 0 or -inf), the zero padding is re-imposed by a select
#pragma unroll
                for (int c = 0; c < S + 2; ++c) {
                    float4 t = make_float4(fmaf(u[c].x, ca.x, cb.x), fmaf(u[c].y, ca.y, cb.y), fmaf(u[c].z, ca.z, cb.z), fmaf(u[c].w, ca.w, cb.w));
                    t.x = fmaxf(t.x, act_lo); t.y = fmaxf(t.y, act_lo); t.z = fmaxf(t.z, act_lo); t.w = fmaxf(t.w, act_lo);
                    if (in_act == FD_ACT_SILU) {        // (uniform)
                        t.x *= __builtin_amdgcn_rcpf(1.0f + __expf(-t.x)); t.y *= __builtin_amdgcn_rcpf(1.0f + __expf(-t.y));
                        t.z *= __builtin_amdgcn_rcpf(1.0f + __expf(-t.z)); t.w *= __builtin_amdgcn_rcpf(1.0f + __expf(-t.w));
                    }
                    const bool inb = (unsigned)(w0 + c - 1) < (unsigned)W;
                    u[c] = inb ? t : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float4 k = *reinterpret_cast<const float4*>(wt + (r * 3 + c) * C + 4 * q);
#pragma unroll
                for (int j = 0; j < S; ++j) {
                    // a tap outside the image contributes nothing: same value as the per-pixel kernel, which skips it
                    if ((unsigned)(w0 + j + c - 1) < (unsigned)W) {
                        acc[j].x = fmaf(u[j + c].x, k.x, acc[j].x); acc[j].y = fmaf(u[j + c].y, k.y, acc[j].y);
                        acc[j].z = fmaf(u[j + c].z, k.z, acc[j].z); acc[j].w = fmaf(u[j + c].w, k.w, acc[j].w);
                    }
                }
            }
        }
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sf = make_float4(0.f, 0.f, 0.f, 0.f);
        if (scale) sc = *reinterpret_cast<const float4*>(scale + 4 * q);
        if (shift) sf = *reinterpret_cast<const float4*>(shift + 4 * q);
#pragma unroll
        for (int j = 0; j < S; ++j) {
            if (w0 + j >= W) break;
            float4 o;
            o = fd_act4(make_float4(acc[j].x * sc.x + sf.x, acc[j].y * sc.y + sf.y, acc[j].z * sc.z + sf.z, acc[j].w * sc.w + sf.w), act, 0.f);
            const long mo = r0 + (long)h * W + w0 + j;
            *reinterpret_cast<float4*>(y + mo * y_cs + y_co + 4 * q) = o;
            if (FUSE && gn_stats) {
                // the cg / 4 threads that hold one group's channels of this pixel are adjacent lanes (q is the fastest index, C4 a multiple
                // of cg / 4 and of ... see the host checks): all of them are here together (same strip, same j)
                float s1 = (o.x + o.y) + (o.z + o.w), s2 = (o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w);
                const int cgq = (C / gn_G) >> 2;
                if (cgq > 1) { s1 += dw_quad_xor1(s1); s2 += dw_quad_xor1(s2); }
                if (cgq > 2) { s1 += dw_quad_xor2(s1); s2 += dw_quad_xor2(s2); }
                if (cgq > 4) { s1 += __shfl_xor(s1, 4); s2 += __shfl_xor(s2, 4); }
                if ((q & (cgq - 1)) == 0) reinterpret_cast<float2*>(gn_stats)[mo * gn_G + (4 * q) / (C / gn_G)] = make_float2(s1, s2);
            }
        }
    }
}

extern "C" int32_t fd_dwconv3x3_nhwc(const float* x, int32_t x_cs, int32_t x_co, const float* w, const float* scale,
                                     const float* shift, float* y, int32_t y_cs, int32_t y_co, int32_t C, int32_t act,
                                     const fd_segs* segs, fd_stream_t stream) {
    FD_REQUIRE(fd_segs_ok(segs), FD_E_INVAL, "fd_dwconv3x3: bad segment table");
    FD_REQUIRE(view_ok(x, x_cs, x_co, C) && view_ok(y, y_cs, y_co, C) && w && ((uintptr_t)w & 15) == 0, FD_E_INVAL,
               "fd_dwconv3x3: channel views must be 4-aligned (C=%d)", C);
    StripTab tab; tab.s = *segs;
    // 8-pixel strips once the launch is large enough to fill the machine with them (measured: head pyramid 512 ch
    // 185 -> 164 us; the 128-channel HisBlock maps are faster with 4)
    const int S = (long)segs->m_start[segs->nseg] * (C / 4) >= (16L << 20) ? 8 : 4;
    long strips = 0;
    for (int s = 0; s < FD_MAX_SEG; ++s) {
        tab.strip_start[s] = strips;
        tab.spr[s] = s < segs->nseg ? (segs->W[s] + S - 1) / S : 1;
        if (s < segs->nseg) strips += (long)segs->batch * segs->H[s] * tab.spr[s];
    }
    tab.strip_start[FD_MAX_SEG] = strips;
    const long total = strips * (C / 4);
    if (S == 8)
        hipLaunchKernelGGL(dwconv3x3_strip_kernel<8>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, x_cs, x_co,
                           w, scale, shift, y, y_cs, y_co, C, act, tab, total);
    else
        hipLaunchKernelGGL(dwconv3x3_strip_kernel<4>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, x_cs, x_co,
                           w, scale, shift, y, y_cs, y_co, C, act, tab, total);
    FD_CHECK_LAUNCH("fd_dwconv3x3_nhwc");
    return FD_OK;
}

extern "C" int32_t fd_dwconv3x3_gn_nhwc(const float* x, int32_t x_cs, int32_t x_co, const float* w, const float* in_coef, int32_t in_act,
                                        float* y, int32_t y_cs, int32_t y_co, int32_t C, float* gn_stats, int32_t gn_groups,
                                        const fd_segs* segs, fd_stream_t stream) {
    FD_REQUIRE(fd_segs_ok(segs), FD_E_INVAL, "fd_dwconv3x3_gn: bad segment table");
    FD_REQUIRE(view_ok(x, x_cs, x_co, C) && view_ok(y, y_cs, y_co, C) && w && ((uintptr_t)w & 15) == 0, FD_E_INVAL,
               "fd_dwconv3x3_gn: channel views must be 4-aligned (C=%d)", C);
    FD_REQUIRE(!in_coef || (((uintptr_t)in_coef & 15) == 0 && (in_act == FD_ACT_NONE || in_act == FD_ACT_RELU || in_act == FD_ACT_SILU)), FD_E_INVAL,
               "fd_dwconv3x3_gn: in_coef must be 16-byte aligned, in_act in {NONE, RELU, SILU}");
    if (gn_stats) {
        const int cg = gn_groups >= 1 && C % gn_groups == 0 ? C / gn_groups : 0;
        // a group's cg / 4 threads must sit in one wave, aligned: C / 4 a multiple of 64 (or a divisor of 64) keeps a pixel's quads wave-aligned
        FD_REQUIRE(cg >= 4 && cg % 4 == 0 && 32 % cg == 0 && (((C / 4) % 64 == 0) || (64 % (C / 4) == 0)) && ((uintptr_t)gn_stats & 7) == 0,
                   FD_E_UNSUPPORTED, "fd_dwconv3x3_gn: gn_stats needs 4 | C / groups | 32 and C / 4 a multiple or divisor of 64 (C=%d groups=%d)", C, gn_groups);
    }
    StripTab tab; tab.s = *segs;
    const int S = (long)segs->m_start[segs->nseg] * (C / 4) >= (16L << 20) ? 8 : 4;
    long strips = 0;
    for (int s = 0; s < FD_MAX_SEG; ++s) {
        tab.strip_start[s] = strips;
        tab.spr[s] = s < segs->nseg ? (segs->W[s] + S - 1) / S : 1;
        if (s < segs->nseg) strips += (long)segs->batch * segs->H[s] * tab.spr[s];
    }
    tab.strip_start[FD_MAX_SEG] = strips;
    const long total = strips * (C / 4);
    // whole waves only: the shuffles of the statistics need every lane of a group active (the grid-stride loop keeps wave-aligned i)
    if (S == 8)
        hipLaunchKernelGGL((dwconv3x3_strip_kernel<8, true>), dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, x_cs, x_co,
                           w, (const float*)nullptr, (const float*)nullptr, y, y_cs, y_co, C, FD_ACT_NONE, tab, total, in_coef, in_act, gn_stats, gn_groups);
    else
        hipLaunchKernelGGL((dwconv3x3_strip_kernel<4, true>), dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, x_cs, x_co,
                           w, (const float*)nullptr, (const float*)nullptr, y, y_cs, y_co, C, FD_ACT_NONE, tab, total, in_coef, in_act, gn_stats, gn_groups);
    FD_CHECK_LAUNCH("fd_dwconv3x3_gn_nhwc");
    return FD_OK;
}

// ---- dilated depthwise k x k, stride 1, 'same' padding (pad = dil * (k - 1) / 2), any pyramid: the DilatedDepthWiseConv of the
// reference's MNBlock (model/modules/modules.py:195-216, used by MNFCOS's light-weight FPN / head, model/od/MNFcos.py:222-297) with its
// BatchNorm folded in.  One thread = one pixel x 4 channels (MNFCOS maps are 128-256 channels on <= 80 x 80: a launch-latency-sized op).
__global__ __launch_bounds__(256) void dwconv_dilated_kernel(const float* __restrict__ x, int x_cs, int x_co, const float* __restrict__ wt,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              float* __restrict__ y, int y_cs, int y_co, int C, int K, int dil, int act,
                                                              SegTab tab, long total) {
    const int C4 = C >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int q;
        const long m = fd_div(i, C4, q);
        int H, W, h, w;
        long row0;
        seg_decode(tab.s, m, H, W, row0, h, w);
        const int half = (K - 1) / 2;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int r = 0; r < K; ++r) {
            const int hi = h + (r - half) * dil;
            if ((unsigned)hi >= (unsigned)H) continue;
            for (int c = 0; c < K; ++c) {
                const int wi = w + (c - half) * dil;
                if ((unsigned)wi >= (unsigned)W) continue;
                const float4 v = *reinterpret_cast<const float4*>(x + (row0 + (long)hi * W + wi) * x_cs + x_co + 4 * q);
                const float4 k = *reinterpret_cast<const float4*>(wt + (r * K + c) * C + 4 * q);
                acc.x = fmaf(v.x, k.x, acc.x); acc.y = fmaf(v.y, k.y, acc.y);
                acc.z = fmaf(v.z, k.z, acc.z); acc.w = fmaf(v.w, k.w, acc.w);
            }
        }
        if (scale) {
            const float4 sc = *reinterpret_cast<const float4*>(scale + 4 * q);
            acc.x *= sc.x; acc.y *= sc.y; acc.z *= sc.z; acc.w *= sc.w;
        }
        if (shift) {
            const float4 sf = *reinterpret_cast<const float4*>(shift + 4 * q);
            acc.x += sf.x; acc.y += sf.y; acc.z += sf.z; acc.w += sf.w;
        }
        acc = fd_act4(acc, act, 0.f);
        *reinterpret_cast<float4*>(y + m * y_cs + y_co + 4 * q) = acc;
    }
}

extern "C" int32_t fd_dwconv_dilated_nhwc(const float* x, int32_t x_cs, int32_t x_co, const float* w, const float* scale,
                                          const float* shift, float* y, int32_t y_cs, int32_t y_co, int32_t C, int32_t K, int32_t dil,
                                          int32_t act, const fd_segs* segs, fd_stream_t stream) {
    FD_REQUIRE(fd_segs_ok(segs), FD_E_INVAL, "fd_dwconv_dilated: bad segment table");
    FD_REQUIRE(view_ok(x, x_cs, x_co, C) && view_ok(y, y_cs, y_co, C) && w && ((uintptr_t)w & 15) == 0, FD_E_INVAL,
               "fd_dwconv_dilated: channel views must be 4-aligned (C=%d)", C);
    FD_REQUIRE((K == 3 || K == 5 || K == 7) && dil >= 1 && dil <= 8, FD_E_UNSUPPORTED, "fd_dwconv_dilated: k in {3, 5, 7}, 1 <= dilation <= 8 (k=%d dil=%d)", K, dil);
    SegTab tab; tab.s = *segs;
    const long total = (long)segs->m_start[segs->nseg] * (C / 4);
    hipLaunchKernelGGL(dwconv_dilated_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, x_cs, x_co, w, scale, shift,
                       y, y_cs, y_co, C, K, dil, act, tab, total);
    FD_CHECK_LAUNCH("fd_dwconv_dilated_nhwc");
    return FD_OK;
}

// ---- depthwise 3x3 weight gradient: dw[t][c] = sum_m x[pix(m, t)][c] * dy[m][c]  (HBM-bound; x and dy read once)
// pass 1: one workgroup per row chunk; a thread owns one channel quad and every R-th row of the chunk, keeps the
//         9 tap sums in registers, lanes are combined through LDS in lane order -> partial[chunk][9][C]
// pass 2: fixed-order fp64 sum of the chunk partials -> dw[9][C]
__global__ __launch_bounds__(256) void dwconv3x3_wgrad_partial_kernel(const float* __restrict__ x, int x_cs, int x_co,
                                                                       const float* __restrict__ dy, int dy_cs, int dy_co,
                                                                       int C, int QW, long chunk_rows, SegTab tab,
                                                                       float* __restrict__ part) {
    __shared__ float4 red[256];
    const int tid = threadIdx.x;
    const int R = 256 / QW;                                  // row lanes
    const int ql = tid % QW, rl = tid / QW;
    const int q = blockIdx.y * QW + ql;                      // channel quad
    const long rows = tab.s.m_start[tab.s.nseg];
    const long m0 = (long)blockIdx.x * chunk_rows;
    const long m1 = min(rows, m0 + chunk_rows);
    float4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long m = m0 + rl; m < m1; m += R) {
        int H, W, h, w;
        long r0;
        seg_decode(tab.s, m, H, W, r0, h, w);
        const float4 g = *reinterpret_cast<const float4*>(dy + m * dy_cs + dy_co + 4 * q);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int hi = h + r - 1;
            if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int wi = w + c - 1;
                if ((unsigned)wi >= (unsigned)W) continue;
                const float4 u = *reinterpret_cast<const float4*>(x + (r0 + hi * W + wi) * x_cs + x_co + 4 * q);
                float4& a = acc[r * 3 + c];
                a.x = fmaf(u.x, g.x, a.x); a.y = fmaf(u.y, g.y, a.y);
                a.z = fmaf(u.z, g.z, a.z); a.w = fmaf(u.w, g.w, a.w);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        red[tid] = acc[t];
        __syncthreads();
        if (rl == 0) {
            float4 v = red[ql];
            for (int j = 1; j < R; ++j) {
                const float4 u = red[j * QW + ql];
                v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
            }
            *reinterpret_cast<float4*>(part + ((long)blockIdx.x * 9 + t) * C + 4 * q) = v;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void dwconv3x3_wgrad_final_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                     int n, int nchunk, const float* __restrict__ scale,
                                                                     int layout, int C) {
    // 16 outputs x 16 chunk lanes per workgroup: lane j adds chunks j, j+16, ... in order, lanes are combined in order
    __shared__ double red[256];
    const int o = threadIdx.x & 15, j = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + o;
    double s = 0.0;
    if (i < n)
        for (int k = j; k < nchunk; k += 16) s += (double)part[(long)k * n + i];
    red[threadIdx.x] = s;
    __syncthreads();
    if (j == 0 && i < n) {
        double t = 0.0;
        for (int q = 0; q < 16; ++q) t += red[q * 16 + o];
        const int tap = i / C, c = i - tap * C;
        float v = (float)t;
        if (scale) v *= scale[c];
        dw[layout ? c * 9 + tap : i] = v;
    }
}

static inline int dw_wgrad_chunks(long rows) {
    long n = (rows + 63) / 64;
    if (n > 1024) n = 1024;
    if (n < 1) n = 1;
    return (int)n;
}

extern "C" int64_t fd_dwconv3x3_wgrad_workspace_bytes(const fd_segs* segs, int32_t C) {
    if (!fd_segs_ok(segs) || C < 4) return -1;
    return (int64_t)dw_wgrad_chunks(segs->m_start[segs->nseg]) * 9 * C * 4;
}

extern "C" int32_t fd_dwconv3x3_bwd_weight_nhwc(const float* x, int32_t x_cs, int32_t x_co, const float* dy, int32_t dy_cs,
                                                int32_t dy_co, float* dw, int32_t C, const float* scale, int32_t layout,
                                                const fd_segs* segs, void* workspace, fd_stream_t stream) {
    FD_REQUIRE(fd_segs_ok(segs), FD_E_INVAL, "fd_dwconv3x3_bwd_weight: bad segment table");
    FD_REQUIRE(view_ok(x, x_cs, x_co, C) && view_ok(dy, dy_cs, dy_co, C) && dw && workspace && ((uintptr_t)workspace & 15) == 0,
               FD_E_INVAL, "fd_dwconv3x3_bwd_weight: channel views must be 4-aligned (C=%d)", C);
    const int C4 = C / 4;
    const int QW = C4 < 256 ? C4 : 256;
    FD_REQUIRE((C4 < 256 && 256 % C4 == 0) || C4 % 256 == 0, FD_E_UNSUPPORTED,
               "fd_dwconv3x3_bwd_weight: C/4 = %d must divide 256 or be a multiple of it", C4);
    const long rows = segs->m_start[segs->nseg];
    const int nchunk = dw_wgrad_chunks(rows);
    const long chunk_rows = (rows + nchunk - 1) / nchunk;
    SegTab tab; tab.s = *segs;
    hipLaunchKernelGGL(dwconv3x3_wgrad_partial_kernel, dim3(nchunk, C4 / QW), dim3(256), 0, (hipStream_t)stream, x, x_cs, x_co,
                       dy, dy_cs, dy_co, C, QW, chunk_rows, tab, (float*)workspace);
    FD_CHECK_LAUNCH("fd_dwconv3x3_bwd_weight_nhwc (partial)");
    hipLaunchKernelGGL(dwconv3x3_wgrad_final_kernel, dim3((9 * C + 15) / 16), dim3(256), 0, (hipStream_t)stream,
                       (const float*)workspace, dw, 9 * C, nchunk, scale, layout, C);
    FD_CHECK_LAUNCH("fd_dwconv3x3_bwd_weight_nhwc (final)");
    return FD_OK;
}

// ------------------------------------------------------------------------------ GroupNorm + activation
// pass 1: per (level, image, row-chunk) partial (sum, sumsq) per group, fp64, fixed order
// pass 2: per (level, image) finalise mean / rstd from the partials, normalise + affine + act
#define GN_MAXCHUNK 256   /* row chunks per (level, image): BatchNorm-as-GroupNorm runs the whole batch as ONE image */

__global__ __launch_bounds__(256) void gn_partial_kernel(const float* __restrict__ x, int x_cs, int x_co, int C, int G,
                                                          SegTab tab, double* __restrict__ part) {
    __shared__ double s_sum[256 * 4];
    __shared__ double s_sq[256 * 4];
    const int img = blockIdx.y;  // level-major: img = s * batch + n
    const int s = img / tab.s.batch, n = img - s * tab.s.batch;
    const int HW = tab.s.H[s] * tab.s.W[s];
    const int nchunk = min(GN_MAXCHUNK, (HW + 63) / 64);
    const int chunk = blockIdx.x;
    if (chunk >= nchunk) return;
    const int rows_per = (HW + nchunk - 1) / nchunk;
    const int r_begin = chunk * rows_per, r_end = min(HW, r_begin + rows_per);
    const int C4 = C >> 2, RT = 256 / C4;
    const int tid = threadIdx.x, q = tid % C4, rt = tid / C4;
    double su[4] = {0, 0, 0, 0}, sq[4] = {0, 0, 0, 0};
    if (rt < RT) {
        const float* base = x + ((long)tab.s.m_start[s] + (long)n * HW) * x_cs + x_co + 4 * q;
        // four rows' loads in flight per thread (one per iteration left the pass latency-bound at 3.5 TB/s); same summation order
        int r = r_begin + rt;
        for (; r + 3 * RT < r_end; r += 4 * RT) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(base + (long)(r + u * RT) * x_cs);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                su[0] += v[u].x; sq[0] += (double)v[u].x * v[u].x; su[1] += v[u].y; sq[1] += (double)v[u].y * v[u].y;
                su[2] += v[u].z; sq[2] += (double)v[u].z * v[u].z; su[3] += v[u].w; sq[3] += (double)v[u].w * v[u].w;
            }
        }
        for (; r < r_end; r += RT) {
            const float4 v = *reinterpret_cast<const float4*>(base + (long)r * x_cs);
            su[0] += v.x; sq[0] += (double)v.x * v.x; su[1] += v.y; sq[1] += (double)v.y * v.y;
            su[2] += v.z; sq[2] += (double)v.z * v.z; su[3] += v.w; sq[3] += (double)v.w * v.w;
        }
    }
    // s_*[rt][c], c = 4q + e
    if (rt < RT) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { s_sum[rt * C + 4 * q + e] = su[e]; s_sq[rt * C + 4 * q + e] = sq[e]; }
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        double a = 0, b = 0;
        for (int r = 0; r < RT; ++r) { a += s_sum[r * C + c]; b += s_sq[r * C + c]; }
        s_sum[c] = a; s_sq[c] = b;   // row 0 of the table: only thread c touches column c
    }
    __syncthreads();
    const int cg = C / G;
    for (int g = tid; g < G; g += 256) {
        double a = 0, b = 0;
        for (int c = 0; c < cg; ++c) { a += s_sum[g * cg + c]; b += s_sq[g * cg + c]; }
        double* o = part + (((long)img * GN_MAXCHUNK + chunk) * G + g) * 2;
        o[0] = a; o[1] = b;
    }
}

// pass 1b: chunk partials -> (mean, rstd) per (level, image, group), once (the apply workgroups used to redo this sum each):
// 256 threads = G-lanes x chunk-lanes, every chunk-lane adds its chunks in index order, lanes are combined in lane order
// coef != NULL: also the per-channel affine a consumer applies in its loader (the arithmetic of gn_apply_kernel) -- no launch of its own
__global__ __launch_bounds__(256) void gn_finalize_kernel(int C, int G, float eps, SegTab tab, const double* __restrict__ part,
                                                           double* __restrict__ gstat, const float* __restrict__ gamma = nullptr,
                                                           const float* __restrict__ beta = nullptr, float* __restrict__ coef = nullptr) {
    __shared__ double s_a[256], s_b[256];
    const int img = blockIdx.y;
    const int s = img / tab.s.batch;
    const int HW = tab.s.H[s] * tab.s.W[s];
    const int nchunk = min(GN_MAXCHUNK, (HW + 63) / 64);
    const int GL = 16, CL = 16;                      // 16 groups x 16 chunk lanes per workgroup
    const int gl = threadIdx.x % GL, cl = threadIdx.x / GL;
    const int g = blockIdx.x * GL + gl;
    double a = 0, b = 0;
    if (g < G)
        for (int k = cl; k < nchunk; k += CL) {
            const double* p = part + (((long)img * GN_MAXCHUNK + k) * G + g) * 2;
            a += p[0]; b += p[1];
        }
    s_a[threadIdx.x] = a; s_b[threadIdx.x] = b;
    __syncthreads();
    if (cl == 0 && g < G) {
        for (int j = 1; j < CL; ++j) { a += s_a[j * GL + gl]; b += s_b[j * GL + gl]; }
        const double cnt = (double)HW * (C / G);
        const double mean = a / cnt;
        double var = b / cnt - mean * mean;
        if (var < 0) var = 0;
        const double rstd = (double)(float)(1.0 / sqrt(var + (double)eps));
        gstat[((long)img * G + g) * 2] = mean;
        gstat[((long)img * G + g) * 2 + 1] = rstd;
        if (coef) {
            const int cg = C / G;
            for (int c = g * cg; c < (g + 1) * cg; ++c) {
                const float sc = (float)rstd * gamma[c];
                coef[((long)img * 2) * C + c] = sc;
                coef[((long)img * 2 + 1) * C + c] = beta[c] - (float)mean * sc;       // the arithmetic of gn_apply_kernel
            }
        }
    }
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, int x_cs, int x_co,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* __restrict__ y, int y_cs, int y_co, int C, int G, float eps,
                                                        int act, SegTab tab, const double* __restrict__ gstat) {
    __shared__ __attribute__((aligned(16))) float s_a[1024], s_b[1024];  // per-channel scale / bias
    const int img = blockIdx.y;
    const int s = img / tab.s.batch, n = img - s * tab.s.batch;
    const int HW = tab.s.H[s] * tab.s.W[s];
    const int nblk = gridDim.x;
    const int rows_per = (HW + nblk - 1) / nblk;
    const int r_begin = blockIdx.x * rows_per, r_end = min(HW, r_begin + rows_per);
    if (r_begin >= r_end) return;
    const int cg = C / G, tid = threadIdx.x;
    for (int c = tid; c < C; c += 256) {
        const double* p = gstat + ((long)img * G + c / cg) * 2;
        const float rstd = (float)p[1];
        const float sc = rstd * gamma[c];
        s_a[c] = sc;
        s_b[c] = beta[c] - (float)p[0] * sc;
    }
    __syncthreads();
    // C / 4 divides 256 (checked by the host): a thread keeps ONE channel quad (its scale / bias in registers) and walks rows, so the loop
    // has no index division (a 64-bit i / C4, i % C4 per float4 is ~100 VALU instructions on this ISA)
    const int C4 = C >> 2, q = tid % C4, rt = tid / C4, RT = 256 / C4;
    const long row0 = (long)tab.s.m_start[s] + (long)n * HW;
    const float4 sa = *reinterpret_cast<const float4*>(s_a + 4 * q), sb = *reinterpret_cast<const float4*>(s_b + 4 * q);
    const float* xp = x + (row0 + r_begin + rt) * x_cs + x_co + 4 * q;
    float* yp = y + (row0 + r_begin + rt) * y_cs + y_co + 4 * q;
    const long xs = (long)RT * x_cs, ys = (long)RT * y_cs;
    for (int r = r_begin + rt; r < r_end; r += RT, xp += xs, yp += ys) {
        const float4 v = *reinterpret_cast<const float4*>(xp);
        float4 o;
        o = fd_act4(make_float4(v.x * sa.x + sb.x, v.y * sa.y + sb.y, v.z * sa.z + sb.z, v.w * sa.w + sb.w), act, 0.f);
        *reinterpret_cast<float4*>(yp) = o;
    }
}

extern "C" int64_t fd_groupnorm_workspace_bytes(const fd_segs* segs, int32_t G) {
    if (!fd_segs_ok(segs) || G < 1) return -1;
    return (int64_t)segs->nseg * segs->batch * (GN_MAXCHUNK + 1) * G * 2 * (int64_t)sizeof(double);   // partials + (mean, rstd)
}

extern "C" int32_t fd_groupnorm_act_nhwc(const float* x, int32_t x_cs, int32_t x_co, const float* gamma,
                                         const float* beta, float* y, int32_t y_cs, int32_t y_co, int32_t C, int32_t G,
                                         float eps, int32_t act, const fd_segs* segs, void* workspace,
                                         fd_stream_t stream) {
    FD_REQUIRE(fd_segs_ok(segs), FD_E_INVAL, "fd_groupnorm: bad segment table");
    FD_REQUIRE(view_ok(x, x_cs, x_co, C) && view_ok(y, y_cs, y_co, C) && gamma && beta && workspace, FD_E_INVAL,
               "fd_groupnorm: bad pointer / channel view (C=%d)", C);
    FD_REQUIRE(G >= 1 && C % G == 0 && C <= 1024 && 256 % (C / 4) == 0, FD_E_UNSUPPORTED,
               "fd_groupnorm: C=%d G=%d unsupported (C/4 must divide 256, C <= 1024)", C, G);
    FD_REQUIRE((long)segs->nseg * segs->batch <= 65535, FD_E_UNSUPPORTED, "fd_groupnorm: too many (level, image) pairs");
    SegTab tab; tab.s = *segs;
    const int imgs = segs->nseg * segs->batch;
    int maxhw = 0;
    for (int s = 0; s < segs->nseg; ++s) maxhw = max(maxhw, segs->H[s] * segs->W[s]);
    const int nchunk = min(GN_MAXCHUNK, (maxhw + 63) / 64);
    hipLaunchKernelGGL(gn_partial_kernel, dim3(nchunk, imgs), dim3(256), 0, (hipStream_t)stream, x, x_cs, x_co, C, G, tab,
                       (double*)workspace);
    FD_CHECK_LAUNCH("fd_groupnorm (partial)");
    double* gstat = (double*)workspace + (long)imgs * GN_MAXCHUNK * G * 2;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((G + 15) / 16, imgs), dim3(256), 0, (hipStream_t)stream, C, G, eps, tab,
                       (const double*)workspace, gstat);
    FD_CHECK_LAUNCH("fd_groupnorm (finalize)");
    const int ablk = max(1, min(imgs >= 16 ? 64 : 512, (maxhw * (C / 4) + 2047) / 2048));
    hipLaunchKernelGGL(gn_apply_kernel, dim3(ablk, imgs), dim3(256), 0, (hipStream_t)stream, x, x_cs, x_co, gamma, beta, y,
                       y_cs, y_co, C, G, eps, act, tab, (const double*)gstat);
    FD_CHECK_LAUNCH("fd_groupnorm (apply)");
    return FD_OK;
}

// ---- GroupNorm statistics from the producer's row-group sums (fd_conv_params.gn_stats / fd_dwconv3x3_gn_nhwc): rgs[m][g] = (sum, sum of
// squares) of group g's channels in row m.  Same two steps as above with the 8x .. 32x smaller rgs in place of the map: chunk partials
// in fp64 (fixed order) in the layout gn_finalize_kernel reads, then (mean, rstd) per (level, image, group) -- so gn_apply_kernel, the
// backward and fd_batchnorm_update_running work on the result unchanged -- and optionally the per-(level, image, channel) affine
// coef[img][0][c] = rstd * gamma[c], coef[img][1][c] = beta[c] - mean * rstd * gamma[c] for a consumer that normalises in its loader
// (written by the finalise step itself).
__global__ __launch_bounds__(256) void gn_rowstats_partial_kernel(const float* __restrict__ rgs, int G, SegTab tab, double* __restrict__ part) {
    __shared__ double s_p[256];
    const int img = blockIdx.y;
    const int s = img / tab.s.batch, n = img - s * tab.s.batch;
    const int HW = tab.s.H[s] * tab.s.W[s];
    const int nchunk = min(GN_MAXCHUNK, (HW + 63) / 64);
    const int chunk = blockIdx.x;
    if (chunk >= nchunk) return;
    const int rows_per = (HW + nchunk - 1) / nchunk;
    const int r_begin = chunk * rows_per, r_end = min(HW, r_begin + rows_per);
    const int cols = 2 * G, RT = 256 / cols;                 // (host: 2 G divides 256)
    const int tid = threadIdx.x, col = tid % cols, rt = tid / cols;
    const float* base = rgs + ((long)tab.s.m_start[s] + (long)n * HW) * cols + col;
    double acc = 0;
    int r = r_begin + rt;
    for (; r + 3 * RT < r_end; r += 4 * RT) {
        const float v0 = base[(long)r * cols], v1 = base[(long)(r + RT) * cols], v2 = base[(long)(r + 2 * RT) * cols], v3 = base[(long)(r + 3 * RT) * cols];
        acc += v0; acc += v1; acc += v2; acc += v3;
    }
    for (; r < r_end; r += RT) acc += base[(long)r * cols];
    s_p[tid] = acc;
    __syncthreads();
    if (tid < cols) {
        double a = 0;
        for (int k = 0; k < RT; ++k) a += s_p[k * cols + tid];
        part[((long)img * GN_MAXCHUNK + chunk) * cols + tid] = a;      // = part[(img, chunk, g)][k], col = 2 g + k
    }
}

extern "C" int32_t fd_groupnorm_from_rowstats(const float* rowstats, int32_t C, int32_t G, float eps, const float* gamma, const float* beta,
                                              const fd_segs* segs, void* workspace, float* coef, fd_stream_t stream) {
    FD_REQUIRE(fd_segs_ok(segs) && rowstats && workspace, FD_E_INVAL, "fd_groupnorm_from_rowstats: bad argument");
    FD_REQUIRE(G >= 1 && C % G == 0 && 256 % (2 * G) == 0, FD_E_UNSUPPORTED, "fd_groupnorm_from_rowstats: 2 * groups must divide 256 (C=%d G=%d)", C, G);
    FD_REQUIRE(!coef || (gamma && beta && ((uintptr_t)coef & 15) == 0 && C % 4 == 0), FD_E_INVAL, "fd_groupnorm_from_rowstats: coef needs gamma, beta, 16-byte alignment");
    FD_REQUIRE((long)segs->nseg * segs->batch <= 65535, FD_E_UNSUPPORTED, "fd_groupnorm_from_rowstats: too many (level, image) pairs");
    SegTab tab; tab.s = *segs;
    const int imgs = segs->nseg * segs->batch;
    int maxhw = 0;
    for (int s = 0; s < segs->nseg; ++s) maxhw = max(maxhw, segs->H[s] * segs->W[s]);
    const int nchunk = min(GN_MAXCHUNK, (maxhw + 63) / 64);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gn_rowstats_partial_kernel, dim3(nchunk, imgs), dim3(256), 0, st, rowstats, G, tab, (double*)workspace);
    FD_CHECK_LAUNCH("fd_groupnorm_from_rowstats (partial)");
    double* gstat = (double*)workspace + (long)imgs * GN_MAXCHUNK * G * 2;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((G + 15) / 16, imgs), dim3(256), 0, st, C, G, eps, tab, (const double*)workspace, gstat, gamma, beta, coef);
    FD_CHECK_LAUNCH("fd_groupnorm_from_rowstats (finalize + coef)");
    return FD_OK;
}

// statistics only (the first two steps of fd_groupnorm_act_nhwc) + optionally the per-(level, image, channel) affine of fd_groupnorm_from_rowstats:
// for a map whose consumers normalise it themselves (a loader: fd_conv_params.gate_b) or in slices (fd_coef_apply_nhwc)
extern "C" int32_t fd_groupnorm_stats_nhwc(const float* x, int32_t x_cs, int32_t x_co, int32_t C, int32_t G, float eps, const float* gamma, const float* beta,
                                           const fd_segs* segs, void* workspace, float* coef, fd_stream_t stream) {
    FD_REQUIRE(fd_segs_ok(segs), FD_E_INVAL, "fd_groupnorm_stats: bad segment table");
    FD_REQUIRE(view_ok(x, x_cs, x_co, C) && workspace, FD_E_INVAL, "fd_groupnorm_stats: bad pointer / channel view (C=%d)", C);
    FD_REQUIRE(G >= 1 && C % G == 0 && C <= 1024 && 256 % (C / 4) == 0, FD_E_UNSUPPORTED,
               "fd_groupnorm_stats: C=%d G=%d unsupported (C/4 must divide 256, C <= 1024)", C, G);
    FD_REQUIRE(!coef || (gamma && beta && ((uintptr_t)coef & 15) == 0), FD_E_INVAL, "fd_groupnorm_stats: coef needs gamma, beta, 16-byte alignment");
    FD_REQUIRE((long)segs->nseg * segs->batch <= 65535, FD_E_UNSUPPORTED, "fd_groupnorm_stats: too many (level, image) pairs");
    SegTab tab; tab.s = *segs;
    const int imgs = segs->nseg * segs->batch;
    int maxhw = 0;
    for (int s = 0; s < segs->nseg; ++s) maxhw = max(maxhw, segs->H[s] * segs->W[s]);
    const int nchunk = min(GN_MAXCHUNK, (maxhw + 63) / 64);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gn_partial_kernel, dim3(nchunk, imgs), dim3(256), 0, st, x, x_cs, x_co, C, G, tab, (double*)workspace);
    FD_CHECK_LAUNCH("fd_groupnorm_stats (partial)");
    double* gstat = (double*)workspace + (long)imgs * GN_MAXCHUNK * G * 2;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((G + 15) / 16, imgs), dim3(256), 0, st, C, G, eps, tab, (const double*)workspace, gstat, gamma, beta, coef);
    FD_CHECK_LAUNCH("fd_groupnorm_stats (finalize + coef)");
    return FD_OK;
}

// normalise + affine + activation from statistics already in `workspace` (fd_groupnorm_from_rowstats, or a previous fd_groupnorm_act_nhwc)
extern "C" int32_t fd_groupnorm_apply_nhwc(const float* x, int32_t x_cs, int32_t x_co, const float* gamma, const float* beta, float* y,
                                           int32_t y_cs, int32_t y_co, int32_t C, int32_t G, float eps, int32_t act, const fd_segs* segs,
                                           const void* workspace, fd_stream_t stream) {
    FD_REQUIRE(fd_segs_ok(segs), FD_E_INVAL, "fd_groupnorm_apply: bad segment table");
    FD_REQUIRE(view_ok(x, x_cs, x_co, C) && view_ok(y, y_cs, y_co, C) && gamma && beta && workspace, FD_E_INVAL,
               "fd_groupnorm_apply: bad pointer / channel view (C=%d)", C);
    FD_REQUIRE(G >= 1 && C % G == 0 && C <= 1024 && 256 % (C / 4) == 0, FD_E_UNSUPPORTED,
               "fd_groupnorm_apply: C=%d G=%d unsupported (C/4 must divide 256, C <= 1024)", C, G);
    FD_REQUIRE((long)segs->nseg * segs->batch <= 65535, FD_E_UNSUPPORTED, "fd_groupnorm_apply: too many (level, image) pairs");
    SegTab tab; tab.s = *segs;
    const int imgs = segs->nseg * segs->batch;
    int maxhw = 0;
    for (int s = 0; s < segs->nseg; ++s) maxhw = max(maxhw, segs->H[s] * segs->W[s]);
    const double* gstat = (const double*)workspace + (long)imgs * GN_MAXCHUNK * G * 2;
    const int ablk = max(1, min(imgs >= 16 ? 64 : 512, (maxhw * (C / 4) + 2047) / 2048));
    hipLaunchKernelGGL(gn_apply_kernel, dim3(ablk, imgs), dim3(256), 0, (hipStream_t)stream, x, x_cs, x_co, gamma, beta, y, y_cs, y_co, C, G, eps,
                       act, tab, gstat);
    FD_CHECK_LAUNCH("fd_groupnorm_apply_nhwc");
    return FD_OK;
}

// y = act(x * coef[img][0][c] + coef[img][1][c]) over a CHANNEL SLICE of a map whose GroupNorm statistics were reduced over more channels than
// the slice (fd_groupnorm_from_rowstats' coef): the head tower's two GroupNorms are one 64-group reduction, its class half keeps a normalise
// pass (the F(4x4) predictor has no registers for an affine in its loader) while the box half is normalised inside the narrow predictor's loader.
__global__ __launch_bounds__(256) void coef_apply_kernel(const float* __restrict__ x, int x_cs, int x_co, const float* __restrict__ ca,
                                                          const float* __restrict__ cb, int coef_cs, float* __restrict__ y, int y_cs, int y_co,
                                                          int C, int act, SegTab tab) {
    const int img = blockIdx.y;
    const int s = img / tab.s.batch, n = img - s * tab.s.batch;
    const int HW = tab.s.H[s] * tab.s.W[s];
    const int nblk = gridDim.x;
    const int rows_per = (HW + nblk - 1) / nblk;
    const int r_begin = blockIdx.x * rows_per, r_end = min(HW, r_begin + rows_per);
    if (r_begin >= r_end) return;
    const int tid = threadIdx.x;
    const int C4 = C >> 2, q = tid % C4, rt = tid / C4, RT = 256 / C4;       // (host: C / 4 divides 256) a thread keeps one channel quad and walks rows
    const long row0 = (long)tab.s.m_start[s] + (long)n * HW;
    const float4 sa = *reinterpret_cast<const float4*>(ca + (long)img * coef_cs + 4 * q), sb = *reinterpret_cast<const float4*>(cb + (long)img * coef_cs + 4 * q);
    const float* xp = x + (row0 + r_begin + rt) * x_cs + x_co + 4 * q;
    float* yp = y + (row0 + r_begin + rt) * y_cs + y_co + 4 * q;
    const long xs = (long)RT * x_cs, ys = (long)RT * y_cs;
    for (int r = r_begin + rt; r < r_end; r += RT, xp += xs, yp += ys) {
        const float4 v = *reinterpret_cast<const float4*>(xp);
        float4 o;
        o = fd_act4(make_float4(v.x * sa.x + sb.x, v.y * sa.y + sb.y, v.z * sa.z + sb.z, v.w * sa.w + sb.w), act, 0.f);
        *reinterpret_cast<float4*>(yp) = o;
    }
}

extern "C" int32_t fd_coef_apply_nhwc(const float* x, int32_t x_cs, int32_t x_co, const float* coef_a, const float* coef_b, int32_t coef_cs, float* y,
                                      int32_t y_cs, int32_t y_co, int32_t C, int32_t act, const fd_segs* segs, fd_stream_t stream) {
    FD_REQUIRE(fd_segs_ok(segs), FD_E_INVAL, "fd_coef_apply: bad segment table");
    FD_REQUIRE(view_ok(x, x_cs, x_co, C) && view_ok(y, y_cs, y_co, C) && coef_a && coef_b && coef_cs >= C && coef_cs % 4 == 0 &&
                   ((uintptr_t)coef_a & 15) == 0 && ((uintptr_t)coef_b & 15) == 0,
               FD_E_INVAL, "fd_coef_apply: bad pointer / channel view / coefficient rows (C=%d)", C);
    FD_REQUIRE(C >= 4 && C <= 1024 && 256 % (C / 4) == 0, FD_E_UNSUPPORTED, "fd_coef_apply: C=%d unsupported (C/4 must divide 256, C <= 1024)", C);
    FD_REQUIRE((long)segs->nseg * segs->batch <= 65535, FD_E_UNSUPPORTED, "fd_coef_apply: too many (level, image) pairs");
    SegTab tab; tab.s = *segs;
    const int imgs = segs->nseg * segs->batch;
    int maxhw = 0;
    for (int s = 0; s < segs->nseg; ++s) maxhw = max(maxhw, segs->H[s] * segs->W[s]);
    const int ablk = max(1, min(imgs >= 16 ? 64 : 512, (maxhw * (C / 4) + 2047) / 2048));
    hipLaunchKernelGGL(coef_apply_kernel, dim3(ablk, imgs), dim3(256), 0, (hipStream_t)stream, x, x_cs, x_co, coef_a, coef_b, coef_cs, y, y_cs, y_co, C, act, tab);
    FD_CHECK_LAUNCH("fd_coef_apply_nhwc");
    return FD_OK;
}

// ---- GroupNorm + activation backward (train step): dz = dy * act'(z), z = xhat * gamma + beta,
//        dx = rstd * (dz * gamma - mean_g(dz * gamma) - xhat * mean_g(dz * gamma * xhat)),  dgamma = sum dz * xhat,  dbeta = sum dz
// pass 1: per (level, image, row-chunk): per-channel sums A = sum dz, B = sum dz * xhat (fp64, fixed order)
// pass 2: per (level, image): chunk sums -> group sums -> dx; block 0 also leaves the per-image channel sums
// pass 3: dgamma / dbeta = fixed-order sum over (level, image)
// mean / rstd come from the forward call's partial moments (its workspace), so x is not re-reduced.
__device__ __forceinline__ float fd_act_grad(float z, int act) {
    switch (act) {
        case FD_ACT_RELU: return z > 0.f ? 1.f : 0.f;
        case FD_ACT_SILU: { const float sg = fd_sigmoid(z); return sg * (1.f + z * (1.f - sg)); }
        default: return 1.f;
    }
}

__device__ __forceinline__ void gn_channel_stats(const double* __restrict__ gstat, int img, int G, int cg, int C, int tid,
                                                 float* s_mean, float* s_rstd) {
    for (int c = tid; c < C; c += 256) {
        const double* p = gstat + ((long)img * G + c / cg) * 2;
        s_mean[c] = (float)p[0];
        s_rstd[c] = (float)p[1];
    }
}

__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(const float* __restrict__ x, int x_cs, int x_co,
                                                              const float* __restrict__ dy, int dy_cs, int dy_co,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              int C, int G, float eps, int act, SegTab tab,
                                                              const double* __restrict__ fpart, double* __restrict__ part) {
    __shared__ double s_a[256 * 4];
    __shared__ double s_b[256 * 4];
    __shared__ float s_mean[1024], s_rstd[1024];
    const int img = blockIdx.y;
    const int s = img / tab.s.batch, n = img - s * tab.s.batch;
    const int HW = tab.s.H[s] * tab.s.W[s];
    const int nchunk = min(GN_MAXCHUNK, (HW + 63) / 64);
    const int chunk = blockIdx.x;
    if (chunk >= nchunk) return;
    const int tid = threadIdx.x, cg = C / G;
    gn_channel_stats(fpart, img, G, cg, C, tid, s_mean, s_rstd);
    __syncthreads();
    const int rows_per = (HW + nchunk - 1) / nchunk;
    const int r_begin = chunk * rows_per, r_end = min(HW, r_begin + rows_per);
    const int C4 = C >> 2, RT = 256 / C4;
    const int q = tid % C4, rt = tid / C4;
    double sa[4] = {0, 0, 0, 0}, sb[4] = {0, 0, 0, 0};
    if (rt < RT) {
        const long row0 = (long)tab.s.m_start[s] + (long)n * HW;
        float mean[4], rstd[4], gm[4], bt[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            mean[e] = s_mean[4 * q + e]; rstd[e] = s_rstd[4 * q + e]; gm[e] = gamma[4 * q + e]; bt[e] = beta[4 * q + e];
        }
        for (int r = r_begin + rt; r < r_end; r += RT) {
            const float4 v4 = *reinterpret_cast<const float4*>(x + (row0 + r) * x_cs + x_co + 4 * q);
            const float4 g4 = *reinterpret_cast<const float4*>(dy + (row0 + r) * dy_cs + dy_co + 4 * q);
            const float v[4] = {v4.x, v4.y, v4.z, v4.w}, g[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (v[e] - mean[e]) * rstd[e];
                const float dz = g[e] * fd_act_grad(xh * gm[e] + bt[e], act);
                sa[e] += dz; sb[e] += (double)dz * xh;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { s_a[rt * C + 4 * q + e] = sa[e]; s_b[rt * C + 4 * q + e] = sb[e]; }
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        double a = 0, b = 0;
        for (int r = 0; r < RT; ++r) { a += s_a[r * C + c]; b += s_b[r * C + c]; }
        double* o = part + ((long)img * GN_MAXCHUNK + chunk) * 2 * C;
        o[c] = a; o[C + c] = b;
    }
}

__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float* __restrict__ x, int x_cs, int x_co,
                                                            const float* __restrict__ dy, int dy_cs, int dy_co,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ dx, int dx_cs, int dx_co, int C, int G, float eps,
                                                            int act, SegTab tab, const double* __restrict__ fpart,
                                                            const double* __restrict__ img_sums, double cnt_ovr,
                                                            const double* __restrict__ cnt_dev = nullptr) {
    __shared__ float s_mean[1024], s_rstd[1024], s_k1[1024], s_p[1024], s_q[1024];
    __shared__ double s_ga[1024], s_gb[1024];
    const int img = blockIdx.y;
    const int s = img / tab.s.batch, n = img - s * tab.s.batch;
    const int HW = tab.s.H[s] * tab.s.W[s];
    const int nchunk = min(GN_MAXCHUNK, (HW + 63) / 64);
    const int nblk = gridDim.x;
    const int rows_per = (HW + nblk - 1) / nblk;
    const int r_begin = blockIdx.x * rows_per, r_end = min(HW, r_begin + rows_per);
    if (r_begin >= r_end) return;
    const int cg = C / G, tid = threadIdx.x;
    gn_channel_stats(fpart, img, G, cg, C, tid, s_mean, s_rstd);
    for (int c = tid; c < C; c += 256) {
        s_ga[c] = img_sums[(long)img * 2 * C + c] * (double)gamma[c];
        s_gb[c] = img_sums[(long)img * 2 * C + C + c] * (double)gamma[c];
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        const int g = c / cg;
        double s1 = 0, s2 = 0;
        for (int k = 0; k < cg; ++k) { s1 += s_ga[g * cg + k]; s2 += s_gb[g * cg + k]; }
        const double cnt = cnt_dev ? *cnt_dev : (cnt_ovr > 0.0 ? cnt_ovr : (double)HW * cg);     // (synchronised BatchNorm: the GLOBAL row count, host value or device word)
        const double rstd = (double)s_rstd[c], mean = (double)s_mean[c];
        s_k1[c] = (float)(rstd * (double)gamma[c]);
        s_p[c] = (float)(-rstd * rstd * s2 / cnt);
        s_q[c] = (float)(-rstd * s1 / cnt + rstd * rstd * mean * s2 / cnt);
    }
    __syncthreads();
    const int C4 = C >> 2;
    const long row0 = (long)tab.s.m_start[s] + (long)n * HW;
    const long total = (long)(r_end - r_begin) * C4;
    for (long i = tid; i < total; i += 256) {
        int q;
        const long m = row0 + r_begin + fd_div(i, C4, q);
        const float4 v4 = *reinterpret_cast<const float4*>(x + m * x_cs + x_co + 4 * q);
        const float4 g4 = *reinterpret_cast<const float4*>(dy + m * dy_cs + dy_co + 4 * q);
        const float v[4] = {v4.x, v4.y, v4.z, v4.w}, g[4] = {g4.x, g4.y, g4.z, g4.w};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 4 * q + e;
            const float xh = (v[e] - s_mean[c]) * s_rstd[c];
            const float dz = g[e] * fd_act_grad(xh * gamma[c] + beta[c], act);
            o[e] = s_k1[c] * dz + s_p[c] * v[e] + s_q[c];
        }
        *reinterpret_cast<float4*>(dx + m * dx_cs + dx_co + 4 * q) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const double* __restrict__ part, double* __restrict__ img_sums, int C,
                                                             SegTab tab) {
    // 64 columns (of the 2C per-channel sums) x 4 chunk lanes per workgroup; chunk lanes add in index order, then lane order
    __shared__ double s_p[256];
    const int img = blockIdx.y;
    const int s = img / tab.s.batch;
    const int HW = tab.s.H[s] * tab.s.W[s];
    const int nchunk = min(GN_MAXCHUNK, (HW + 63) / 64);
    const int cl = threadIdx.x & 63, kl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    double a = 0;
    if (c < 2 * C)
        for (int k = kl; k < nchunk; k += 4) a += part[((long)img * GN_MAXCHUNK + k) * 2 * C + c];
    s_p[threadIdx.x] = a;
    __syncthreads();
    if (kl == 0 && c < 2 * C) img_sums[(long)img * 2 * C + c] = ((a + s_p[64 + cl]) + s_p[128 + cl]) + s_p[192 + cl];
}

__global__ __launch_bounds__(256) void gn_bwd_param_kernel(const double* __restrict__ img_sums, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int C, int imgs) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double a = 0, b = 0;
    for (int i = 0; i < imgs; ++i) { a += img_sums[(long)i * 2 * C + c]; b += img_sums[(long)i * 2 * C + C + c]; }
    dbeta[c] = (float)a;
    dgamma[c] = (float)b;
}

extern "C" int64_t fd_groupnorm_bwd_workspace_bytes(const fd_segs* segs, int32_t C) {
    if (!fd_segs_ok(segs) || C < 4) return -1;
    const int64_t imgs = (int64_t)segs->nseg * segs->batch;
    return (imgs * GN_MAXCHUNK + imgs) * 2 * C * (int64_t)sizeof(double);
}

extern "C" int32_t fd_groupnorm_act_bwd_nhwc(const float* x, int32_t x_cs, int32_t x_co, const float* dy, int32_t dy_cs,
                                             int32_t dy_co, const float* gamma, const float* beta, float* dx, int32_t dx_cs,
                                             int32_t dx_co, float* dgamma, float* dbeta, int32_t C, int32_t G, float eps,
                                             int32_t act, const fd_segs* segs, const void* fwd_workspace, void* workspace,
                                             fd_stream_t stream) {
    FD_REQUIRE(fd_segs_ok(segs), FD_E_INVAL, "fd_groupnorm_bwd: bad segment table");
    FD_REQUIRE(view_ok(x, x_cs, x_co, C) && view_ok(dy, dy_cs, dy_co, C) && view_ok(dx, dx_cs, dx_co, C) && gamma && beta &&
                   dgamma && dbeta && fwd_workspace && workspace,
               FD_E_INVAL, "fd_groupnorm_bwd: bad pointer / channel view (C=%d)", C);
    FD_REQUIRE(G >= 1 && C % G == 0 && C <= 1024 && 256 % (C / 4) == 0, FD_E_UNSUPPORTED,
               "fd_groupnorm_bwd: C=%d G=%d unsupported (C/4 must divide 256, C <= 1024)", C, G);
    FD_REQUIRE(act == FD_ACT_NONE || act == FD_ACT_RELU || act == FD_ACT_SILU, FD_E_UNSUPPORTED,
               "fd_groupnorm_bwd: activation %d has no backward", act);
    FD_REQUIRE((long)segs->nseg * segs->batch <= 65535, FD_E_UNSUPPORTED, "fd_groupnorm_bwd: too many (level, image) pairs");
    SegTab tab; tab.s = *segs;
    const int imgs = segs->nseg * segs->batch;
    int maxhw = 0;
    for (int s = 0; s < segs->nseg; ++s) maxhw = max(maxhw, segs->H[s] * segs->W[s]);
    const int nchunk = min(GN_MAXCHUNK, (maxhw + 63) / 64);
    double* part = (double*)workspace;
    double* img_sums = part + (long)imgs * GN_MAXCHUNK * 2 * C;
    const double* gstat = (const double*)fwd_workspace + (long)imgs * GN_MAXCHUNK * G * 2;   // (mean, rstd) left by the forward
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gn_bwd_partial_kernel, dim3(nchunk, imgs), dim3(256), 0, st, x, x_cs, x_co, dy, dy_cs, dy_co, gamma, beta, C,
                       G, eps, act, tab, gstat, part);
    FD_CHECK_LAUNCH("fd_groupnorm_bwd (partial)");
    hipLaunchKernelGGL(gn_bwd_reduce_kernel, dim3((2 * C + 63) / 64, imgs), dim3(256), 0, st, (const double*)part, img_sums, C, tab);
    FD_CHECK_LAUNCH("fd_groupnorm_bwd (reduce)");
    const int ablk = max(1, min(imgs >= 16 ? 64 : 512, (maxhw * (C / 4) + 2047) / 2048));
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(ablk, imgs), dim3(256), 0, st, x, x_cs, x_co, dy, dy_cs, dy_co, gamma, beta, dx,
                       dx_cs, dx_co, C, G, eps, act, tab, gstat, (const double*)img_sums, 0.0);
    FD_CHECK_LAUNCH("fd_groupnorm_bwd (apply)");
    hipLaunchKernelGGL(gn_bwd_param_kernel, dim3((C + 255) / 256), dim3(256), 0, st, (const double*)img_sums, dgamma, dbeta, C,
                       imgs);
    FD_CHECK_LAUNCH("fd_groupnorm_bwd (params)");
    return FD_OK;
}

// ------------------------------------------------------------------------------ squeeze-excitation
// Any C % 4 == 0 up to SE_MAXC (HisBlock: 128 -> 32; EfficientNet MBConv: expanded width up to 3840 -> block_in / 4).
#define SE_MAXCHUNK 64
#define SE_MAXC 4096
#define SE_MAXCR 1024

// partial channel sums: grid (row chunk, image, channel tile of QW quads); 256 threads = QW quads x (256 / QW) row lanes
__global__ __launch_bounds__(256) void se_gap_kernel(const float* __restrict__ x, int x_cs, int x_co, int HW, int C,
                                                      int nchunk, int QW, double* __restrict__ part) {
    __shared__ double s_sum[256 * 4];
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int rows_per = (HW + nchunk - 1) / nchunk;
    const int r_begin = chunk * rows_per, r_end = min(HW, r_begin + rows_per);
    const int C4 = C >> 2, RT = 256 / QW, CW = 4 * QW;
    const int tid = threadIdx.x, ql = tid % QW, rt = tid / QW;
    const int q = blockIdx.z * QW + ql;
    double su[4] = {0, 0, 0, 0};
    if (q < C4) {
        const float* base = x + (long)n * HW * x_cs + x_co + 4 * q;
        for (int r = r_begin + rt; r < r_end; r += RT) {
            const float4 v = *reinterpret_cast<const float4*>(base + (long)r * x_cs);
            su[0] += v.x; su[1] += v.y; su[2] += v.z; su[3] += v.w;
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) s_sum[rt * CW + 4 * ql + e] = su[e];
    __syncthreads();
    for (int cl = tid; cl < CW; cl += 256) {
        const int c = blockIdx.z * CW + cl;
        if (c >= C) continue;
        double a = 0;
        for (int r = 0; r < RT; ++r) a += s_sum[r * CW + cl];
        part[((long)n * SE_MAXCHUNK + chunk) * C + c] = a;
    }
}

// gate = sigmoid(W2 silu(W1 mean + b1) + b2), one workgroup per image.  The squeeze dot products (length C) are spread
// over a wave's lanes and combined with a fixed shuffle tree; the excite ones (length Cr) run one output per thread.
__global__ __launch_bounds__(256) void se_fc_kernel(const double* __restrict__ part, int nchunk, int HW, int C, int Cr,
                                                     const float* __restrict__ w1, const float* __restrict__ b1,
                                                     const float* __restrict__ w2, const float* __restrict__ b2,
                                                     float* __restrict__ gate) {
    __shared__ __attribute__((aligned(16))) float s_mean[SE_MAXC];
    __shared__ float s_h[SE_MAXCR];
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // (every loop below keeps several independent loads in flight: with one dependent load per iteration these small loops cost
    //  0.1-0.2 ms per MBConv block on latency alone)
    for (int c = tid; c < C; c += 256) {
        const double* pp = part + (long)n * SE_MAXCHUNK * C + c;
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        int k = 0;
        for (; k + 4 <= nchunk; k += 4) {
            a0 += pp[(long)k * C]; a1 += pp[(long)(k + 1) * C]; a2 += pp[(long)(k + 2) * C]; a3 += pp[(long)(k + 3) * C];
        }
        for (; k < nchunk; ++k) a0 += pp[(long)k * C];
        s_mean[c] = (float)(((a0 + a1) + (a2 + a3)) / (double)HW);
    }
    __syncthreads();
    // squeeze layer: a wave owns rows j = wv, wv + 4, ...; two rows at a time, 4 channels per lane and load (C % 4 == 0)
    const int C4 = C >> 2;
    for (int j = wv; j < Cr; j += 8) {
        const int j2 = j + 4;
        const bool two = j2 < Cr;
        const float4* wa = reinterpret_cast<const float4*>(w1 + (long)j * C);
        const float4* wb = reinterpret_cast<const float4*>(w1 + (long)(two ? j2 : j) * C);
        float a = 0.f, b = 0.f;
        for (int c4 = lane; c4 < C4; c4 += 64) {
            const float4 m = *reinterpret_cast<const float4*>(s_mean + 4 * c4);
            const float4 u = wa[c4], v = wb[c4];
            a = fmaf(u.x, m.x, a); a = fmaf(u.y, m.y, a); a = fmaf(u.z, m.z, a); a = fmaf(u.w, m.w, a);
            b = fmaf(v.x, m.x, b); b = fmaf(v.y, m.y, b); b = fmaf(v.z, m.z, b); b = fmaf(v.w, m.w, b);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
        if (lane == 0) {
            s_h[j] = fd_act(a + (b1 ? b1[j] : 0.f), FD_ACT_SILU, 0.f);
            if (two) s_h[j2] = fd_act(b + (b1 ? b1[j2] : 0.f), FD_ACT_SILU, 0.f);
        }
    }
    __syncthreads();
    // the expand layer: blockIdx.y owns 256 output channels (a thread's w2 row is Cr scattered loads: with one workgroup per image the
    // wide EfficientNet blocks -- C up to 2304 -- spent 0.2-0.4 ms here; the squeeze layer above is recomputed per workgroup, it is cheap)
    const int c = blockIdx.y * 256 + tid;
    if (c < C) {
        const float* wr = w2 + (long)c * Cr;
        float a0 = b2 ? b2[c] : 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int j = 0;
        for (; j + 4 <= Cr; j += 4) {
            a0 = fmaf(wr[j], s_h[j], a0); a1 = fmaf(wr[j + 1], s_h[j + 1], a1);
            a2 = fmaf(wr[j + 2], s_h[j + 2], a2); a3 = fmaf(wr[j + 3], s_h[j + 3], a3);
        }
        for (; j < Cr; ++j) a0 = fmaf(wr[j], s_h[j], a0);
        gate[(long)n * C + c] = fd_sigmoid((a0 + a1) + (a2 + a3));
    }
}

__global__ __launch_bounds__(256) void se_scale_kernel(const float* __restrict__ x, int x_cs, int x_co,
                                                        const float* __restrict__ gate, float* __restrict__ y, int y_cs,
                                                        int y_co, int HW, int C4, long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int q;
        const long m = fd_div(i, C4, q);
        int hw_rem_; const long n = fd_div(m, HW, hw_rem_);
        const float4 v = *reinterpret_cast<const float4*>(x + m * x_cs + x_co + 4 * q);
        const float4 g = *reinterpret_cast<const float4*>(gate + n * C4 * 4 + 4 * q);
        *reinterpret_cast<float4*>(y + m * y_cs + y_co + 4 * q) = make_float4(v.x * g.x, v.y * g.y, v.z * g.z, v.w * g.w);
    }
}

extern "C" int64_t fd_se_workspace_bytes(int32_t N, int32_t HW, int32_t C) {
    if (N < 1 || HW < 1 || C < 1) return -1;
    return (int64_t)N * SE_MAXCHUNK * C * (int64_t)sizeof(double) + (int64_t)N * C * (int64_t)sizeof(float);
}

extern "C" int32_t fd_se_scale_nhwc(const float* x, int32_t x_cs, int32_t x_co, const float* w1, const float* b1,
                                    const float* w2, const float* b2, float* y, int32_t y_cs, int32_t y_co, int32_t N,
                                    int32_t HW, int32_t C, int32_t Cr, void* workspace, fd_stream_t stream) {
    FD_REQUIRE(view_ok(x, x_cs, x_co, C) && (!y || view_ok(y, y_cs, y_co, C)) && w1 && w2 && workspace, FD_E_INVAL,
               "fd_se_scale: bad pointer / channel view (C=%d)", C);
    FD_REQUIRE(N >= 1 && N <= 65535 && HW >= 1 && Cr >= 1 && Cr <= SE_MAXCR && C <= SE_MAXC, FD_E_UNSUPPORTED,
               "fd_se_scale: C=%d Cr=%d unsupported (C <= %d, Cr <= %d)", C, Cr, SE_MAXC, SE_MAXCR);
    FD_REQUIRE(((uintptr_t)workspace & 15) == 0, FD_E_INVAL, "fd_se_scale: workspace not 16-byte aligned");
    double* part = (double*)workspace;
    float* gate = (float*)((char*)workspace + (size_t)N * SE_MAXCHUNK * C * sizeof(double));
    const int nchunk = min(SE_MAXCHUNK, (HW + 63) / 64);
    int QW = 1;                                      // quads per channel tile: a power of two, at most 64
    while (QW < 64 && QW < C / 4) QW <<= 1;
    hipLaunchKernelGGL(se_gap_kernel, dim3(nchunk, N, (C / 4 + QW - 1) / QW), dim3(256), 0, (hipStream_t)stream, x, x_cs, x_co, HW, C,
                       nchunk, QW, part);
    FD_CHECK_LAUNCH("fd_se_scale (gap)");
    hipLaunchKernelGGL(se_fc_kernel, dim3(N, (C + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const double*)part, nchunk, HW, C, Cr, w1,
                       b1, w2, b2, gate);
    FD_CHECK_LAUNCH("fd_se_scale (fc)");
    if (!y) return FD_OK;                            // gates only (left in the workspace for fd_conv_params.gate)
    const long total = (long)N * HW * (C / 4);
    hipLaunchKernelGGL(se_scale_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, x_cs, x_co,
                       (const float*)gate, y, y_cs, y_co, HW, C / 4, total);
    FD_CHECK_LAUNCH("fd_se_scale (scale)");
    return FD_OK;
}


// The squeeze-excitation gates of an MBConv block whose depthwise output came out of fd_mbconv_expand_dw_nhwc (fd_mbconv.hip): the pooled sums are the sums of that
// kernel's per-tile partials -- pool[n][t][c], T tiles, added in tile order: deterministic -- so no pass over the map is needed; then the two FC layers as above.
// The gates are left where fd_se_scale_nhwc(y = NULL) leaves them (the project conv's loader multiplies them in: fd_conv_params.gate).
// chunk k of an image = tiles [k * tpc, (k + 1) * tpc): up to SE_MAXCHUNK workgroups per image and channel slice (one workgroup per image over 1 440 tiles was a
// 0.17 ms latency chain on B3's first fused block); se_fc_kernel then adds the chunks in index order
__global__ __launch_bounds__(256) void se_pool_reduce_kernel(const float* __restrict__ pool, int T, int tpc, int C, double* __restrict__ part) {
    const int n = blockIdx.y, k = blockIdx.z;
    const int t0 = k * tpc, t1 = min(T, t0 + tpc);
    for (int c = blockIdx.x * 256 + threadIdx.x; c < C; c += gridDim.x * 256) {
        const float* p = pool + (size_t)n * T * C + c;
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        int t = t0;
        for (; t + 4 <= t1; t += 4) { a0 += p[(size_t)t * C]; a1 += p[(size_t)(t + 1) * C]; a2 += p[(size_t)(t + 2) * C]; a3 += p[(size_t)(t + 3) * C]; }
        for (; t < t1; ++t) a0 += p[(size_t)t * C];
        part[((size_t)n * SE_MAXCHUNK + k) * C + c] = (a0 + a1) + (a2 + a3);
    }
}

extern "C" int32_t fd_se_gate_from_pool(const float* pool, int32_t T, const float* w1, const float* b1, const float* w2, const float* b2, int32_t N, int32_t HW,
                                        int32_t C, int32_t Cr, void* workspace, fd_stream_t stream) {
    FD_REQUIRE(pool && w1 && w2 && workspace && T >= 1 && C % 4 == 0, FD_E_INVAL, "fd_se_gate_from_pool: bad arguments (C=%d T=%d)", C, T);
    FD_REQUIRE(N >= 1 && N <= 65535 && HW >= 1 && Cr >= 1 && Cr <= SE_MAXCR && C <= SE_MAXC, FD_E_UNSUPPORTED,
               "fd_se_gate_from_pool: C=%d Cr=%d unsupported (C <= %d, Cr <= %d)", C, Cr, SE_MAXC, SE_MAXCR);
    FD_REQUIRE(((uintptr_t)workspace & 15) == 0, FD_E_INVAL, "fd_se_gate_from_pool: workspace not 16-byte aligned");
    double* part = (double*)workspace;
    float* gate = (float*)((char*)workspace + (size_t)N * SE_MAXCHUNK * C * sizeof(double));
    const int tpc = (T + SE_MAXCHUNK - 1) / SE_MAXCHUNK, nchunk = (T + tpc - 1) / tpc;
    hipLaunchKernelGGL(se_pool_reduce_kernel, dim3((C + 255) / 256, N, nchunk), dim3(256), 0, (hipStream_t)stream, pool, T, tpc, C, part);
    FD_CHECK_LAUNCH("fd_se_gate_from_pool (reduce)");
    hipLaunchKernelGGL(se_fc_kernel, dim3(N, (C + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const double*)part, nchunk, HW, C, Cr, w1, b1, w2, b2, gate);
    FD_CHECK_LAUNCH("fd_se_gate_from_pool (fc)");
    return FD_OK;
}

// ---- squeeze-excitation backward (train step; HisBlock.conv1_2, HISFcos.py:104, modules.py:107-121)
//   y = x * g,  g = sigmoid(W2 s + b2),  s = silu(h),  h = W1 m + b1,  m = mean_hw(x)
//   dgate[n][c] = sum_hw dy * x                     (pass 1, row chunks like the forward's pooling)
//   one workgroup: for every image in order  dz = dgate g (1 - g), ds = W2^T dz, dh = ds silu'(h), dm = W1^T dh  and the
//   parameter gradients dW2 += dz s^T, db2 += dz, dW1 += dh m^T, db1 += dh   (images added in index order: deterministic)
//   dx = dy * g + dm / HW                            (pass 3)
// `fwd_workspace` is the forward call's workspace, untouched since (pooled sums and gates).
__global__ __launch_bounds__(256) void se_dgate_kernel(const float* __restrict__ x, int x_cs, int x_co, const float* __restrict__ dy,
                                                        int dy_cs, int dy_co, int HW, int C, int nchunk, int QW, double* __restrict__ part) {
    __shared__ double s_sum[256 * 4];
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int rows_per = (HW + nchunk - 1) / nchunk;
    const int r_begin = chunk * rows_per, r_end = min(HW, r_begin + rows_per);
    const int C4 = C >> 2, RT = 256 / QW, CW = 4 * QW;
    const int tid = threadIdx.x, ql = tid % QW, rt = tid / QW;
    const int q = blockIdx.z * QW + ql;
    double su[4] = {0, 0, 0, 0};
    if (q < C4) {
        const float* bx = x + (long)n * HW * x_cs + x_co + 4 * q;
        const float* bg = dy + (long)n * HW * dy_cs + dy_co + 4 * q;
        for (int r = r_begin + rt; r < r_end; r += RT) {
            const float4 v = *reinterpret_cast<const float4*>(bx + (long)r * x_cs);
            const float4 g = *reinterpret_cast<const float4*>(bg + (long)r * dy_cs);
            su[0] += (double)v.x * g.x; su[1] += (double)v.y * g.y; su[2] += (double)v.z * g.z; su[3] += (double)v.w * g.w;
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) s_sum[rt * CW + 4 * ql + e] = su[e];
    __syncthreads();
    for (int cl = tid; cl < CW; cl += 256) {
        const int c = blockIdx.z * CW + cl;
        if (c >= C) continue;
        double a = 0;
        for (int r = 0; r < RT; ++r) a += s_sum[r * CW + cl];
        part[((long)n * SE_MAXCHUNK + chunk) * C + c] = a;
    }
}

// per image: dz, s = silu(h), dh, dm (one workgroup each) ...
__global__ __launch_bounds__(256) void se_bwd_fc_kernel(const double* __restrict__ fpart, const float* __restrict__ gate,
                                                         const double* __restrict__ dpart, int nchunk, int HW, int C, int Cr,
                                                         const float* __restrict__ w1, const float* __restrict__ b1,
                                                         const float* __restrict__ w2, float* __restrict__ dmean,
                                                         float* __restrict__ vec) {   // vec[n]: m[C] | dz[C] | s[Cr] | dh[Cr]
    __shared__ float s_m[SE_MAXC], s_dz[SE_MAXC];
    __shared__ float s_h[SE_MAXCR], s_dh[SE_MAXCR];
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float* vm = vec + (long)n * (2 * C + 2 * Cr);
    float* vdz = vm + C; float* vs = vdz + C; float* vdh = vs + Cr;
    for (int c = tid; c < C; c += 256) {
        double a = 0, d = 0;
        for (int k = 0; k < nchunk; ++k) {
            a += fpart[((long)n * SE_MAXCHUNK + k) * C + c];
            d += dpart[((long)n * SE_MAXCHUNK + k) * C + c];
        }
        const float m = (float)(a / (double)HW);
        const float g = gate[(long)n * C + c];
        const float dz = (float)d * g * (1.f - g);
        s_m[c] = m; s_dz[c] = dz; vm[c] = m; vdz[c] = dz;
    }
    __syncthreads();
    for (int j = wv; j < Cr; j += 4) {       // h = W1 m + b1 (same lane split and shuffle tree as the forward)
        float a = 0.f;
        const float* wr = w1 + (long)j * C;
        for (int c = lane; c < C; c += 64) a = fmaf(wr[c], s_m[c], a);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
        float ds = 0.f;                      // ds[j] = sum_c W2[c][j] dz[c]
        for (int c = lane; c < C; c += 64) ds = fmaf(w2[(long)c * Cr + j], s_dz[c], ds);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ds += __shfl_xor(ds, o);
        if (lane == 0) {
            const float h = a + (b1 ? b1[j] : 0.f), sg = fd_sigmoid(h);
            const float dh = ds * (sg * (1.f + h * (1.f - sg)));
            s_h[j] = h; s_dh[j] = dh;
            vs[j] = fd_act(h, FD_ACT_SILU, 0.f); vdh[j] = dh;
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {     // dm[c] = sum_j W1[j][c] dh[j]
        float a = 0.f;
        for (int j = 0; j < Cr; ++j) a = fmaf(w1[(long)j * C + c], s_dh[j], a);
        dmean[(long)n * C + c] = a;
    }
}

// ... then the parameter gradients, images added in index order: dW2[c][j] = sum_n dz s^T, dW1[j][c] = sum_n dh m^T, db2, db1
__global__ __launch_bounds__(256) void se_bwd_param_kernel(const float* __restrict__ vec, int N, int C, int Cr, float* __restrict__ dw1,
                                                            float* __restrict__ db1, float* __restrict__ dw2, float* __restrict__ db2) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int stride = 2 * C + 2 * Cr;
    if (i < C * Cr) {
        const int c2 = i / Cr, j2 = i - c2 * Cr;
        const int j1 = i / C, c1 = i - j1 * C;
        float a2 = 0.f, a1 = 0.f;
        for (int n = 0; n < N; ++n) {
            const float* v = vec + (long)n * stride;
            a2 = fmaf(v[C + c2], v[2 * C + j2], a2);
            a1 = fmaf(v[2 * C + Cr + j1], v[c1], a1);
        }
        dw2[i] = a2; dw1[i] = a1;
    }
    if (i < C) { float a = 0.f; for (int n = 0; n < N; ++n) a += vec[(long)n * stride + C + i]; db2[i] = a; }
    if (i < Cr) { float a = 0.f; for (int n = 0; n < N; ++n) a += vec[(long)n * stride + 2 * C + Cr + i]; db1[i] = a; }
}

__global__ __launch_bounds__(256) void se_bwd_apply_kernel(const float* __restrict__ dy, int dy_cs, int dy_co, const float* __restrict__ gate,
                                                            const float* __restrict__ dmean, float* __restrict__ dx, int dx_cs, int dx_co,
                                                            int HW, int C4, float inv_hw, long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int q;
        const long m = fd_div(i, C4, q);
        int hw_rem_; const long n = fd_div(m, HW, hw_rem_);
        const float4 v = *reinterpret_cast<const float4*>(dy + m * dy_cs + dy_co + 4 * q);
        const float4 g = *reinterpret_cast<const float4*>(gate + n * C4 * 4 + 4 * q);
        const float4 d = *reinterpret_cast<const float4*>(dmean + n * C4 * 4 + 4 * q);
        *reinterpret_cast<float4*>(dx + m * dx_cs + dx_co + 4 * q) =
            make_float4(fmaf(d.x, inv_hw, v.x * g.x), fmaf(d.y, inv_hw, v.y * g.y), fmaf(d.z, inv_hw, v.z * g.z), fmaf(d.w, inv_hw, v.w * g.w));
    }
}

extern "C" int64_t fd_se_bwd_workspace_bytes(int32_t N, int32_t HW, int32_t C) {
    if (N < 1 || HW < 1 || C < 1) return -1;       // chunk sums of dy*x | dmean[N][C] | per-image vectors m, dz, s, dh (<= 4C floats)
    return (int64_t)N * SE_MAXCHUNK * C * (int64_t)sizeof(double) + (int64_t)N * C * 5 * (int64_t)sizeof(float);
}

extern "C" int32_t fd_se_scale_bwd_nhwc(const float* x, int32_t x_cs, int32_t x_co, const float* dy, int32_t dy_cs, int32_t dy_co,
                                        const float* w1, const float* b1, const float* w2, const float* b2, float* dx, int32_t dx_cs,
                                        int32_t dx_co, float* dw1, float* db1, float* dw2, float* db2, int32_t N, int32_t HW, int32_t C,
                                        int32_t Cr, const void* fwd_workspace, void* workspace, fd_stream_t stream) {
    FD_REQUIRE(view_ok(x, x_cs, x_co, C) && view_ok(dy, dy_cs, dy_co, C) && view_ok(dx, dx_cs, dx_co, C) && w1 && w2 && dw1 && db1 && dw2 &&
                   db2 && fwd_workspace && workspace,
               FD_E_INVAL, "fd_se_scale_bwd: bad pointer / channel view (C=%d)", C);
    FD_REQUIRE(N >= 1 && N <= 65535 && HW >= 1 && Cr >= 1 && Cr <= SE_MAXCR && C <= SE_MAXC, FD_E_UNSUPPORTED,
               "fd_se_scale_bwd: C=%d Cr=%d unsupported", C, Cr);
    FD_REQUIRE(((uintptr_t)workspace & 15) == 0 && ((uintptr_t)fwd_workspace & 15) == 0, FD_E_INVAL, "fd_se_scale_bwd: workspace not 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const double* fpart = (const double*)fwd_workspace;
    const float* gate = (const float*)((const char*)fwd_workspace + (size_t)N * SE_MAXCHUNK * C * sizeof(double));
    double* dpart = (double*)workspace;
    float* dmean = (float*)((char*)workspace + (size_t)N * SE_MAXCHUNK * C * sizeof(double));
    const int nchunk = min(SE_MAXCHUNK, (HW + 63) / 64);
    int QW = 1;
    while (QW < 64 && QW < C / 4) QW <<= 1;
    hipLaunchKernelGGL(se_dgate_kernel, dim3(nchunk, N, (C / 4 + QW - 1) / QW), dim3(256), 0, st, x, x_cs, x_co, dy, dy_cs, dy_co, HW, C, nchunk,
                       QW, dpart);
    FD_CHECK_LAUNCH("fd_se_scale_bwd (dgate)");
    FD_REQUIRE(Cr <= C, FD_E_UNSUPPORTED, "fd_se_scale_bwd: Cr=%d > C=%d", Cr, C);
    float* vec = dmean + (size_t)N * C;
    hipLaunchKernelGGL(se_bwd_fc_kernel, dim3(N), dim3(256), 0, st, fpart, gate, (const double*)dpart, nchunk, HW, C, Cr, w1, b1, w2, dmean, vec);
    FD_CHECK_LAUNCH("fd_se_scale_bwd (fc)");
    hipLaunchKernelGGL(se_bwd_param_kernel, dim3((C * Cr + 255) / 256), dim3(256), 0, st, (const float*)vec, N, C, Cr, dw1, db1, dw2, db2);
    FD_CHECK_LAUNCH("fd_se_scale_bwd (params)");
    (void)b2;
    const long total = (long)N * HW * (C / 4);
    hipLaunchKernelGGL(se_bwd_apply_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, dy, dy_cs, dy_co, gate, (const float*)dmean, dx, dx_cs,
                       dx_co, HW, C / 4, 1.0f / (float)HW, total);
    FD_CHECK_LAUNCH("fd_se_scale_bwd (apply)");
    return FD_OK;
}

// ---- BatchNorm2d in TRAINING mode on the GroupNorm kernels: statistics per channel over the whole batch are GroupNorm
// statistics of ONE image of batch*H rows with G = C groups.  This entry point folds the batch statistics the forward left in
// its workspace into the running statistics the way nn.BatchNorm2d does (momentum update, unbiased variance; HISFcos.py FPN
// BatchNorms under the reference's model.train(), train.py:151).
__global__ __launch_bounds__(256) void bn_running_kernel(const double* __restrict__ gstat, float* __restrict__ rmean, float* __restrict__ rvar,
                                                          int C, double count, float momentum, float eps, const double* __restrict__ count_dev = nullptr) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    if (count_dev) count = *count_dev;
    const double mean = gstat[2 * c], rstd = gstat[2 * c + 1];
    double var = 1.0 / (rstd * rstd) - (double)eps;          // biased batch variance the forward normalised with
    if (var < 0) var = 0;
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    rmean[c] = (float)((1.0 - (double)momentum) * (double)rmean[c] + (double)momentum * mean);
    rvar[c] = (float)((1.0 - (double)momentum) * (double)rvar[c] + (double)momentum * unbiased);
}

extern "C" int32_t fd_batchnorm_update_running(const void* gn_workspace, int64_t rows, int32_t C, float momentum, float eps,
                                               float* running_mean, float* running_var, fd_stream_t stream) {
    FD_REQUIRE(gn_workspace && running_mean && running_var && rows >= 1 && C >= 1, FD_E_INVAL, "fd_batchnorm_update_running: bad argument");
    const double* gstat = (const double*)gn_workspace + (long)GN_MAXCHUNK * C * 2;     // imgs = 1, G = C
    hipLaunchKernelGGL(bn_running_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gstat, running_mean, running_var, C,
                       (double)rows, momentum, eps);
    FD_CHECK_LAUNCH("fd_batchnorm_update_running");
    return FD_OK;
}

extern "C" int32_t fd_batchnorm_update_running_dev(const void* gn_workspace, const double* count_dev, int32_t C, float momentum, float eps,
                                                   float* running_mean, float* running_var, fd_stream_t stream) {
    FD_REQUIRE(gn_workspace && count_dev && running_mean && running_var && C >= 1, FD_E_INVAL, "fd_batchnorm_update_running_dev: bad argument");
    const double* gstat = (const double*)gn_workspace + (long)GN_MAXCHUNK * C * 2;
    hipLaunchKernelGGL(bn_running_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gstat, running_mean, running_var, C,
                       0.0, momentum, eps, count_dev);
    FD_CHECK_LAUNCH("fd_batchnorm_update_running_dev");
    return FD_OK;
}

// ---- nn.SyncBatchNorm (train.py:103 converts the model; collective C3 of SURVEY 2.1) on the same kernels.  Batch statistics
// are sums over ALL ranks' rows, so forward and backward are cut in two around ONE all-reduce each, issued by the caller
// (torch.distributed over RCCL): phase 1 leaves this rank's per-channel fp64 sums (fixed summation order) in `sums`,
// the caller all-reduces them (and the row count), phase 2 finishes from the global sums.
//   forward  sums = [sum x | sum x^2]            (2C doubles, channel-major pairs: sums[2c], sums[2c+1])
//   backward sums = [sum dz | sum dz * xhat]     (2C doubles: sums[c], sums[C + c]), dz = dy * act'(z)
__global__ __launch_bounds__(256) void bn_sync_sums_kernel(const double* __restrict__ part, int nchunk, int C, double* __restrict__ sums) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double a = 0, b = 0;
    for (int k = 0; k < nchunk; ++k) { a += part[((long)k * C + c) * 2]; b += part[((long)k * C + c) * 2 + 1]; }
    sums[2 * c] = a; sums[2 * c + 1] = b;
}

__global__ __launch_bounds__(256) void bn_sync_finalize_kernel(const double* __restrict__ sums, double count, float eps, int C,
                                                                double* __restrict__ gstat) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    if (count <= 0.0) count = sums[2 * C];           // the all-reduced row count riding behind the 2C sums (uneven shards: no host round trip)
    const double mean = sums[2 * c] / count;
    double var = sums[2 * c + 1] / count - mean * mean;
    if (var < 0) var = 0;
    gstat[2 * c] = mean;
    gstat[2 * c + 1] = (double)(float)(1.0 / sqrt(var + (double)eps));        // rounded like gn_finalize_kernel
}

static bool bn_rows_ok(int64_t rows, int32_t C) { return rows >= 1 && rows < (1L << 31) && C >= 4 && C <= 1024 && C % 4 == 0 && 256 % (C / 4) == 0; }

extern "C" int32_t fd_batchnorm_sync_fwd_nhwc(const float* x, int32_t x_cs, int32_t x_co, const float* gamma, const float* beta, float* y,
                                              int32_t y_cs, int32_t y_co, int64_t rows, int32_t C, float eps, int32_t act, int32_t phase,
                                              double* sums, double total_rows, void* workspace, fd_stream_t stream) {
    FD_REQUIRE(bn_rows_ok(rows, C), FD_E_UNSUPPORTED, "fd_batchnorm_sync_fwd: rows=%ld C=%d unsupported (C/4 must divide 256, C <= 1024)", (long)rows, C);
    FD_REQUIRE(view_ok(x, x_cs, x_co, C) && sums && workspace && (phase == 1 || phase == 2), FD_E_INVAL, "fd_batchnorm_sync_fwd: bad argument");
    SegTab tab;
    tab = SegTab{};
    tab.s.nseg = 1; tab.s.batch = 1; tab.s.H[0] = (int)rows; tab.s.W[0] = 1; tab.s.m_start[0] = 0; tab.s.m_start[1] = (int)rows;
    const int nchunk = (int)min((int64_t)GN_MAXCHUNK, (rows + 63) / 64);
    hipStream_t st = (hipStream_t)stream;
    double* gstat = (double*)workspace + (long)GN_MAXCHUNK * C * 2;          // where fd_groupnorm_act_nhwc (imgs = 1, G = C) leaves (mean, rstd)
    if (phase == 1) {
        hipLaunchKernelGGL(gn_partial_kernel, dim3(nchunk, 1), dim3(256), 0, st, x, x_cs, x_co, C, C, tab, (double*)workspace);
        FD_CHECK_LAUNCH("fd_batchnorm_sync_fwd (partial)");
        hipLaunchKernelGGL(bn_sync_sums_kernel, dim3((C + 255) / 256), dim3(256), 0, st, (const double*)workspace, nchunk, C, sums);
        FD_CHECK_LAUNCH("fd_batchnorm_sync_fwd (sums)");
        return FD_OK;
    }
    FD_REQUIRE(view_ok(y, y_cs, y_co, C) && gamma && beta && (total_rows <= 0.0 || total_rows >= (double)rows), FD_E_INVAL,
               "fd_batchnorm_sync_fwd: phase 2 needs y, gamma, beta and the global row count (host value >= rows, or <= 0: read from sums[2C] on the device)");
    hipLaunchKernelGGL(bn_sync_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, st, (const double*)sums, total_rows, eps, C, gstat);
    FD_CHECK_LAUNCH("fd_batchnorm_sync_fwd (finalize)");
    const int ablk = (int)max((int64_t)1, min((int64_t)512, (rows * (C / 4) + 2047) / 2048));
    hipLaunchKernelGGL(gn_apply_kernel, dim3(ablk, 1), dim3(256), 0, st, x, x_cs, x_co, gamma, beta, y, y_cs, y_co, C, C, eps, act, tab,
                       (const double*)gstat);
    FD_CHECK_LAUNCH("fd_batchnorm_sync_fwd (apply)");
    return FD_OK;
}

extern "C" int32_t fd_batchnorm_sync_bwd_nhwc(const float* x, int32_t x_cs, int32_t x_co, const float* dy, int32_t dy_cs, int32_t dy_co,
                                              const float* gamma, const float* beta, float* dx, int32_t dx_cs, int32_t dx_co, float* dgamma,
                                              float* dbeta, int64_t rows, int32_t C, float eps, int32_t act, int32_t phase, double* sums,
                                              double total_rows, const void* fwd_workspace, void* workspace, fd_stream_t stream) {
    FD_REQUIRE(bn_rows_ok(rows, C), FD_E_UNSUPPORTED, "fd_batchnorm_sync_bwd: rows=%ld C=%d unsupported", (long)rows, C);
    FD_REQUIRE(view_ok(x, x_cs, x_co, C) && view_ok(dy, dy_cs, dy_co, C) && gamma && beta && sums && fwd_workspace && (phase == 1 || phase == 2),
               FD_E_INVAL, "fd_batchnorm_sync_bwd: bad argument");
    FD_REQUIRE(act == FD_ACT_NONE || act == FD_ACT_RELU || act == FD_ACT_SILU, FD_E_UNSUPPORTED, "fd_batchnorm_sync_bwd: activation %d has no backward", act);
    SegTab tab;
    tab = SegTab{};
    tab.s.nseg = 1; tab.s.batch = 1; tab.s.H[0] = (int)rows; tab.s.W[0] = 1; tab.s.m_start[0] = 0; tab.s.m_start[1] = (int)rows;
    const int nchunk = (int)min((int64_t)GN_MAXCHUNK, (rows + 63) / 64);
    hipStream_t st = (hipStream_t)stream;
    const double* gstat = (const double*)fwd_workspace + (long)GN_MAXCHUNK * C * 2;   // GLOBAL (mean, rstd) left by the forward's phase 2
    if (phase == 1) {
        FD_REQUIRE(workspace && dgamma && dbeta, FD_E_INVAL, "fd_batchnorm_sync_bwd: phase 1 needs the workspace, dgamma, dbeta");
        hipLaunchKernelGGL(gn_bwd_partial_kernel, dim3(nchunk, 1), dim3(256), 0, st, x, x_cs, x_co, dy, dy_cs, dy_co, gamma, beta, C, C, eps, act, tab,
                           gstat, (double*)workspace);
        FD_CHECK_LAUNCH("fd_batchnorm_sync_bwd (partial)");
        hipLaunchKernelGGL(gn_bwd_reduce_kernel, dim3((2 * C + 63) / 64, 1), dim3(256), 0, st, (const double*)workspace, sums, C, tab);
        FD_CHECK_LAUNCH("fd_batchnorm_sync_bwd (reduce)");
        // the affine gradients come from THIS rank's sums (DistributedDataParallel averages them like every other gradient)
        hipLaunchKernelGGL(gn_bwd_param_kernel, dim3((C + 255) / 256), dim3(256), 0, st, (const double*)sums, dgamma, dbeta, C, 1);
        FD_CHECK_LAUNCH("fd_batchnorm_sync_bwd (params)");
        return FD_OK;
    }
    FD_REQUIRE(view_ok(dx, dx_cs, dx_co, C) && (total_rows <= 0.0 || total_rows >= (double)rows), FD_E_INVAL,
               "fd_batchnorm_sync_bwd: phase 2 needs dx and the global row count (host value >= rows, or <= 0: read from sums[2C] on the device)");
    const int ablk = (int)max((int64_t)1, min((int64_t)512, (rows * (C / 4) + 2047) / 2048));
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(ablk, 1), dim3(256), 0, st, x, x_cs, x_co, dy, dy_cs, dy_co, gamma, beta, dx, dx_cs, dx_co, C, C, eps,
                       act, tab, gstat, (const double*)sums, total_rows, total_rows <= 0.0 ? (const double*)sums + 2 * C : nullptr);
    FD_CHECK_LAUNCH("fd_batchnorm_sync_bwd (apply)");
    return FD_OK;
}
