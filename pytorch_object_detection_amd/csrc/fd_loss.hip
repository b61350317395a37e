// fd_loss.hip — fused masked LTRB IoU / GIoU regression loss, forward + backward (gfx950).
// Restates reference model/loss.py:116-177 without the boolean-mask gathers / per-image Python loop:
// one pass over [B][L][4] pred/target + [B][L] positive mask.  Sub-gradient conventions follow torch autograd
// (minimum/maximum split ties evenly, clamp(min) passes the gradient at equality).
#include "fd_common.h"

struct LtrbTerms {
    float wi, hi, ov, a1, a2, U, iou, wg, hg, G;
};

__device__ __forceinline__ LtrbTerms ltrb_terms(const float4 p, const float4 t) {
    LtrbTerms k;
    k.wi = fmaxf(fminf(p.z, t.z) + fminf(p.x, t.x), 0.f);
    k.hi = fmaxf(fminf(p.w, t.w) + fminf(p.y, t.y), 0.f);
    k.ov = k.wi * k.hi;
    k.a1 = (p.z + p.x) * (p.w + p.y);
    k.a2 = (t.z + t.x) * (t.w + t.y);
    k.U = (k.a1 + k.a2) - k.ov;
    k.iou = k.ov / k.U;
    k.wg = fmaxf(fmaxf(p.z, t.z) + fmaxf(p.x, t.x), 0.f);
    k.hg = fmaxf(fmaxf(p.w, t.w) + fmaxf(p.y, t.y), 0.f);
    k.G = k.wg * k.hg;
    return k;
}

__device__ __forceinline__ float ltrb_loss(const LtrbTerms& k, int mode) {
    if (mode == 0) return -logf(fmaxf(k.iou, 1e-6f));
    const float giou = k.iou - (k.G - k.U) / fmaxf(k.G, 1e-10f);
    return 1.0f - giou;
}

__global__ __launch_bounds__(256) void ltrb_loss_fwd_kernel(const float4* __restrict__ pred, const float4* __restrict__ tgt,
                                                             const unsigned char* __restrict__ mask, int L, int mode,
                                                             float* __restrict__ loss, int* __restrict__ num_pos) {
    __shared__ double s_sum[256];
    __shared__ int s_cnt[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    double acc = 0.0;
    int cnt = 0;
    for (int i = tid; i < L; i += 256) {
        const long o = (long)b * L + i;
        if (mask[o]) {
            acc += (double)ltrb_loss(ltrb_terms(pred[o], tgt[o]), mode);
            ++cnt;
        }
    }
    s_sum[tid] = acc; s_cnt[tid] = cnt;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) { s_sum[tid] += s_sum[tid + s]; s_cnt[tid] += s_cnt[tid + s]; }
        __syncthreads();
    }
    if (tid == 0) { loss[b] = (float)s_sum[0]; num_pos[b] = s_cnt[0]; }
}

// d min(a,b)/da and d max(a,b)/da with torch's even split on ties
__device__ __forceinline__ float pick_min(float a, float b) { return a < b ? 1.f : (a == b ? 0.5f : 0.f); }
__device__ __forceinline__ float pick_max(float a, float b) { return a > b ? 1.f : (a == b ? 0.5f : 0.f); }

__global__ __launch_bounds__(256) void ltrb_loss_bwd_kernel(const float4* __restrict__ pred, const float4* __restrict__ tgt,
                                                             const unsigned char* __restrict__ mask,
                                                             const float* __restrict__ gscale, int L, int mode, long total,
                                                             float4* __restrict__ grad) {
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long)gridDim.x * 256) {
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (mask[o]) {
            const float4 p = pred[o], t = tgt[o];
            const LtrbTerms k = ltrb_terms(p, t);
            const float gs = gscale[o / L];
            // d(ov), d(a1) w.r.t. (l, t, r, b) = (x, y, z, w)
            const float wpass = (fminf(p.z, t.z) + fminf(p.x, t.x)) >= 0.f ? 1.f : 0.f;
            const float hpass = (fminf(p.w, t.w) + fminf(p.y, t.y)) >= 0.f ? 1.f : 0.f;
            const float dov[4] = {k.hi * wpass * pick_min(p.x, t.x), k.wi * hpass * pick_min(p.y, t.y),
                                  k.hi * wpass * pick_min(p.z, t.z), k.wi * hpass * pick_min(p.w, t.w)};
            const float da1[4] = {p.w + p.y, p.z + p.x, p.w + p.y, p.z + p.x};
            float dG[4] = {0.f, 0.f, 0.f, 0.f};
            if (mode == 1) {
                const float wgp = (fmaxf(p.z, t.z) + fmaxf(p.x, t.x)) >= 0.f ? 1.f : 0.f;
                const float hgp = (fmaxf(p.w, t.w) + fmaxf(p.y, t.y)) >= 0.f ? 1.f : 0.f;
                dG[0] = k.hg * wgp * pick_max(p.x, t.x); dG[1] = k.wg * hgp * pick_max(p.y, t.y);
                dG[2] = k.hg * wgp * pick_max(p.z, t.z); dG[3] = k.wg * hgp * pick_max(p.w, t.w);
            }
            float out[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dU = da1[e] - dov[e];
                const float diou = (dov[e] * k.U - k.ov * dU) / (k.U * k.U);
                float dl;
                if (mode == 0) {
                    dl = (k.iou >= 1e-6f) ? -diou / k.iou : 0.f;
                } else {
                    const float Gc = fmaxf(k.G, 1e-10f);
                    const float dGc = (k.G >= 1e-10f) ? dG[e] : 0.f;
                    const float dterm = (dG[e] - dU) / Gc - (k.G - k.U) / (Gc * Gc) * dGc;
                    dl = -diou + dterm;
                }
                out[e] = dl * gs;
            }
            g = make_float4(out[0], out[1], out[2], out[3]);
        }
        grad[o] = g;
    }
}

extern "C" int32_t fd_ltrb_iou_loss_fwd(const float* pred, const float* target, const uint8_t* mask, int32_t B, int32_t L,
                                        int32_t mode, float* loss_per_image, int32_t* num_pos, fd_stream_t stream) {
    FD_REQUIRE(pred && target && mask && loss_per_image && num_pos, FD_E_INVAL, "fd_ltrb_iou_loss_fwd: null pointer");
    FD_REQUIRE(B >= 1 && L >= 1 && (mode == 0 || mode == 1), FD_E_INVAL, "fd_ltrb_iou_loss_fwd: bad argument");
    FD_REQUIRE((((uintptr_t)pred | (uintptr_t)target) & 15) == 0, FD_E_INVAL, "fd_ltrb_iou_loss_fwd: not 16-byte aligned");
    hipLaunchKernelGGL(ltrb_loss_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, (const float4*)pred,
                       (const float4*)target, mask, L, mode, loss_per_image, num_pos);
    FD_CHECK_LAUNCH("fd_ltrb_iou_loss_fwd");
    return FD_OK;
}

extern "C" int32_t fd_ltrb_iou_loss_bwd(const float* pred, const float* target, const uint8_t* mask, const float* gscale,
                                        int32_t B, int32_t L, int32_t mode, float* grad_pred, fd_stream_t stream) {
    FD_REQUIRE(pred && target && mask && gscale && grad_pred, FD_E_INVAL, "fd_ltrb_iou_loss_bwd: null pointer");
    FD_REQUIRE(B >= 1 && L >= 1 && (mode == 0 || mode == 1), FD_E_INVAL, "fd_ltrb_iou_loss_bwd: bad argument");
    FD_REQUIRE((((uintptr_t)pred | (uintptr_t)target | (uintptr_t)grad_pred) & 15) == 0, FD_E_INVAL,
               "fd_ltrb_iou_loss_bwd: not 16-byte aligned");
    const long total = (long)B * L;
    long g = (total + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(ltrb_loss_bwd_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const float4*)pred,
                       (const float4*)target, mask, gscale, L, mode, total, (float4*)grad_pred);
    FD_CHECK_LAUNCH("fd_ltrb_iou_loss_bwd");
    return FD_OK;
}

// ==============================================================================================
// Focal classification loss from logits (reference model/loss.py:180-193, called per image by
// compute_cls_loss :6-28): p = clip(sigmoid(x), 5e-6, 1) [the 0.99999999995 upper clip is 1.0 in fp32],
// pt = p*t + (1-p)*(1-t), w = a*t + (1-a)*(1-t), loss = -w * (1-pt)^2 * log(pt), summed over [L][C].
// t = one-hot(label) with labels 1..C (0 = background).  Same fp32 operation order as the torch graph.
// ==============================================================================================
#define FOCAL_MAXCHUNK 64

__device__ __forceinline__ float focal_term(float x, float t, float alpha, float* dldx) {
    const float s = fd_sigmoid(x);
    const float p = fminf(fmaxf(s, 0.000005f), 1.0f);
    const float pt = p * t + (1.0f - p) * (1.0f - t);
    const float w = alpha * t + (1.0f - alpha) * (1.0f - t);
    const float om = 1.0f - pt;
    const float lg = logf(pt);
    const float loss = -w * (om * om) * lg;
    if (dldx) {
        // d/dpt of -w*(1-pt)^2*log(pt); dpt/dp = 2t-1; clip passes the gradient on [5e-6, 1]; dp/dx = s(1-s)
        const float dpt = -w * (-2.0f * om * lg + (om * om) / pt);
        const float pass = (s >= 0.000005f && s <= 1.0f) ? 1.0f : 0.0f;
        *dldx = dpt * (2.0f * t - 1.0f) * pass * (s * (1.0f - s));
    }
    return loss;
}

__global__ __launch_bounds__(256) void focal_fwd_kernel(const float* __restrict__ x, const long long* __restrict__ labels,
                                                         int L, int C, float alpha, int nchunk, double* __restrict__ part) {
    __shared__ double s_sum[256];
    const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const long total = (long)L * C;
    const long per = (total + nchunk - 1) / nchunk;
    const long lo = chunk * per, hi = min(total, lo + per);
    double acc = 0.0;
    for (long i = lo + tid; i < hi; i += 256) {
        const long loc = i / C;
        const int c = (int)(i - loc * C);
        const float t = (labels[(long)b * L + loc] == (long long)(c + 1)) ? 1.0f : 0.0f;
        acc += (double)focal_term(x[(long)b * total + i], t, alpha, nullptr);
    }
    s_sum[tid] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) s_sum[tid] += s_sum[tid + s];
        __syncthreads();
    }
    if (tid == 0) part[(long)b * FOCAL_MAXCHUNK + chunk] = s_sum[0];
}

__global__ void focal_final_kernel(const double* __restrict__ part, int nchunk, float* __restrict__ loss) {
    const int b = blockIdx.x;
    if (threadIdx.x == 0) {
        double a = 0.0;
        for (int k = 0; k < nchunk; ++k) a += part[(long)b * FOCAL_MAXCHUNK + k];
        loss[b] = (float)a;
    }
}

__global__ __launch_bounds__(256) void focal_bwd_kernel(const float* __restrict__ x, const long long* __restrict__ labels,
                                                         const float* __restrict__ gscale, int L, int C, float alpha,
                                                         long total, float* __restrict__ grad) {
    const long per_img = (long)L * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long b = i / per_img;
        const long r = i - b * per_img;
        const long loc = r / C;
        const int c = (int)(r - loc * C);
        const float t = (labels[b * L + loc] == (long long)(c + 1)) ? 1.0f : 0.0f;
        float d;
        focal_term(x[i], t, alpha, &d);
        grad[i] = d * gscale[b];
    }
}

extern "C" int64_t fd_focal_workspace_bytes(int32_t B) { return B < 1 ? -1 : (int64_t)B * FOCAL_MAXCHUNK * (int64_t)sizeof(double); }

extern "C" int32_t fd_focal_loss_fwd(const float* logits, const int64_t* labels, int32_t B, int32_t L, int32_t C, float alpha,
                                     float gamma, float* loss_per_image, void* workspace, fd_stream_t stream) {
    FD_REQUIRE(logits && labels && loss_per_image && workspace, FD_E_INVAL, "fd_focal_loss_fwd: null pointer");
    FD_REQUIRE(B >= 1 && B <= 65535 && L >= 1 && C >= 1, FD_E_INVAL, "fd_focal_loss_fwd: bad shape");
    FD_REQUIRE(gamma == 2.0f, FD_E_UNSUPPORTED, "fd_focal_loss_fwd: only gamma = 2 (the reference's value) is built");
    const long total = (long)L * C;
    const int nchunk = (int)max(1L, min((long)FOCAL_MAXCHUNK, (total + 16383) / 16384));
    hipLaunchKernelGGL(focal_fwd_kernel, dim3(nchunk, B), dim3(256), 0, (hipStream_t)stream, logits, (const long long*)labels, L,
                       C, alpha, nchunk, (double*)workspace);
    FD_CHECK_LAUNCH("fd_focal_loss_fwd");
    hipLaunchKernelGGL(focal_final_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, (const double*)workspace, nchunk,
                       loss_per_image);
    FD_CHECK_LAUNCH("fd_focal_loss_fwd (final)");
    return FD_OK;
}

extern "C" int32_t fd_focal_loss_bwd(const float* logits, const int64_t* labels, const float* gscale, int32_t B, int32_t L,
                                     int32_t C, float alpha, float gamma, float* grad_logits, fd_stream_t stream) {
    FD_REQUIRE(logits && labels && gscale && grad_logits, FD_E_INVAL, "fd_focal_loss_bwd: null pointer");
    FD_REQUIRE(B >= 1 && L >= 1 && C >= 1, FD_E_INVAL, "fd_focal_loss_bwd: bad shape");
    FD_REQUIRE(gamma == 2.0f, FD_E_UNSUPPORTED, "fd_focal_loss_bwd: only gamma = 2 is built");
    const long total = (long)B * L * C;
    long g = (total + 255) / 256;
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(focal_bwd_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, logits, (const long long*)labels,
                       gscale, L, C, alpha, total, grad_logits);
    FD_CHECK_LAUNCH("fd_focal_loss_bwd");
    return FD_OK;
}

// ==============================================================================================
// Centerness loss: BCE-with-logits summed over positives (reference model/loss.py:31-57)
// ==============================================================================================
__global__ __launch_bounds__(256) void bce_fwd_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                       const unsigned char* __restrict__ mask, int L, float* __restrict__ loss,
                                                       int* __restrict__ num_pos) {
    __shared__ double s_sum[256];
    __shared__ int s_cnt[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    double acc = 0.0;
    int cnt = 0;
    for (int i = tid; i < L; i += 256) {
        const long o = (long)b * L + i;
        if (mask[o]) {
            const float v = x[o], tt = t[o];
            const float mx = fmaxf(-v, 0.0f);
            acc += (double)((1.0f - tt) * v + mx + logf(expf(-mx) + expf(-v - mx)));
            ++cnt;
        }
    }
    s_sum[tid] = acc; s_cnt[tid] = cnt;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) { s_sum[tid] += s_sum[tid + s]; s_cnt[tid] += s_cnt[tid + s]; }
        __syncthreads();
    }
    if (tid == 0) { loss[b] = (float)s_sum[0]; num_pos[b] = s_cnt[0]; }
}

__global__ __launch_bounds__(256) void bce_bwd_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                       const unsigned char* __restrict__ mask, const float* __restrict__ gscale,
                                                       int L, long total, float* __restrict__ grad) {
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long)gridDim.x * 256)
        grad[o] = mask[o] ? (fd_sigmoid(x[o]) - t[o]) * gscale[o / L] : 0.0f;
}

extern "C" int32_t fd_bce_logits_loss_fwd(const float* x, const float* target, const uint8_t* mask, int32_t B, int32_t L,
                                          float* loss_per_image, int32_t* num_pos, fd_stream_t stream) {
    FD_REQUIRE(x && target && mask && loss_per_image && num_pos, FD_E_INVAL, "fd_bce_logits_loss_fwd: null pointer");
    FD_REQUIRE(B >= 1 && L >= 1, FD_E_INVAL, "fd_bce_logits_loss_fwd: bad shape");
    hipLaunchKernelGGL(bce_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, target, mask, L, loss_per_image, num_pos);
    FD_CHECK_LAUNCH("fd_bce_logits_loss_fwd");
    return FD_OK;
}

extern "C" int32_t fd_bce_logits_loss_bwd(const float* x, const float* target, const uint8_t* mask, const float* gscale,
                                          int32_t B, int32_t L, float* grad, fd_stream_t stream) {
    FD_REQUIRE(x && target && mask && gscale && grad, FD_E_INVAL, "fd_bce_logits_loss_bwd: null pointer");
    FD_REQUIRE(B >= 1 && L >= 1, FD_E_INVAL, "fd_bce_logits_loss_bwd: bad shape");
    const long total = (long)B * L;
    long g = (total + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(bce_bwd_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, target, mask, gscale, L, total, grad);
    FD_CHECK_LAUNCH("fd_bce_logits_loss_bwd");
    return FD_OK;
}

// ==============================================================================================
// FCOS target assignment (reference model/modules/head.py:211-316 FCOSGenTargets), one thread per
// (image, location), GT loop in registers — no [B][HW][M][4] temporaries.
// ==============================================================================================
struct TargetArgs {
    const float* gt; const long long* labels;
    int B, M, L, nseg;
    int H[FD_MAX_SEG], W[FD_MAX_SEG], stride[FD_MAX_SEG], lo[FD_MAX_SEG], hi[FD_MAX_SEG], loc_start[FD_MAX_SEG + 1];
    float radius;
    long long* cls_t; float* cnt_t; float* reg_t;
};

__global__ __launch_bounds__(256) void gen_targets_kernel(TargetArgs a) {
    const int b = blockIdx.y;
    const int loc = blockIdx.x * 256 + threadIdx.x;
    if (loc >= a.L) return;
    int s = 0;
#pragma unroll
    for (int t = 1; t < FD_MAX_SEG; ++t)
        if (t < a.nseg && loc >= a.loc_start[t]) s = t;
    const int pix = loc - a.loc_start[s];
    const int W = a.W[s];
    const int py = pix / W, px = pix - py * W;
    const int st = a.stride[s];
    const float x = (float)(px * st) + (float)(st / 2), y = (float)(py * st) + (float)(st / 2);
    const float lo = (float)a.lo[s], hi = (float)a.hi[s];
    const float ratio = (float)st * a.radius;

    float best_area = 0.f;
    int best = 0;
    bool any_pos = false;
    float4 best_off = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int m = 0; m < a.M; ++m) {
        const float4 g = reinterpret_cast<const float4*>(a.gt)[(long)b * a.M + m];
        const float4 off = make_float4(x - g.x, y - g.y, g.z - x, g.w - y);
        float area = (off.x + off.z) * (off.y + off.w);
        const float omin = fminf(fminf(off.x, off.y), fminf(off.z, off.w));
        const float omax = fmaxf(fmaxf(off.x, off.y), fmaxf(off.z, off.w));
        const float cx = (g.x + g.z) / 2.0f, cy = (g.y + g.w) / 2.0f;
        const float cmax = fmaxf(fmaxf(x - cx, y - cy), fmaxf(cx - x, cy - y));
        const bool pos = (omin > 0.f) && (omax > lo) && (omax <= hi) && (cmax < ratio);
        if (!pos) area = 99999999.0f;
        any_pos = any_pos || pos;
        if (m == 0 || area < best_area) { best_area = area; best = m; best_off = off; }  // first minimum wins
    }
    const long o = (long)b * a.L + loc;
    if (any_pos) {
        const float lr_min = fminf(best_off.x, best_off.z), lr_max = fmaxf(best_off.x, best_off.z);
        const float tb_min = fminf(best_off.y, best_off.w), tb_max = fmaxf(best_off.y, best_off.w);
        a.cls_t[o] = a.labels[(long)b * a.M + best];
        a.cnt_t[o] = sqrtf((lr_min * tb_min) / (lr_max * tb_max + 1e-10f));
        reinterpret_cast<float4*>(a.reg_t)[o] = best_off;
    } else {
        a.cls_t[o] = 0;
        a.cnt_t[o] = -1.0f;
        reinterpret_cast<float4*>(a.reg_t)[o] = make_float4(-1.f, -1.f, -1.f, -1.f);
    }
}

extern "C" int32_t fd_fcos_gen_targets(const float* gt_boxes, const int64_t* labels, int32_t M, const fd_segs* segs,
                                       const int32_t* strides, const int32_t* range_lo, const int32_t* range_hi,
                                       float radius_ratio, int64_t* cls_target, float* cnt_target, float* reg_target,
                                       fd_stream_t stream) {
    FD_REQUIRE(gt_boxes && labels && strides && range_lo && range_hi && cls_target && cnt_target && reg_target, FD_E_INVAL,
               "fd_fcos_gen_targets: null pointer");
    FD_REQUIRE(fd_segs_ok(segs) && M >= 1, FD_E_INVAL, "fd_fcos_gen_targets: bad level table / M");
    FD_REQUIRE((((uintptr_t)gt_boxes | (uintptr_t)reg_target) & 15) == 0, FD_E_INVAL, "fd_fcos_gen_targets: not 16-byte aligned");
    TargetArgs a;
    a.gt = gt_boxes; a.labels = (const long long*)labels; a.B = segs->batch; a.M = M; a.nseg = segs->nseg;
    int L = 0;
    for (int s = 0; s < FD_MAX_SEG; ++s) {
        a.loc_start[s] = L;
        if (s < segs->nseg) {
            a.H[s] = segs->H[s]; a.W[s] = segs->W[s]; a.stride[s] = strides[s]; a.lo[s] = range_lo[s]; a.hi[s] = range_hi[s];
            L += segs->H[s] * segs->W[s];
        } else { a.H[s] = a.W[s] = a.stride[s] = 1; a.lo[s] = a.hi[s] = 0; }
    }
    a.loc_start[FD_MAX_SEG] = L; a.L = L; a.radius = radius_ratio;
    a.cls_t = (long long*)cls_target; a.cnt_t = cnt_target; a.reg_t = reg_target;
    hipLaunchKernelGGL(gen_targets_kernel, dim3((L + 255) / 256, segs->batch), dim3(256), 0, (hipStream_t)stream, a);
    FD_CHECK_LAUNCH("fd_fcos_gen_targets");
    return FD_OK;
}
