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
