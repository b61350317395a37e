// fd_common.h — shared host/device helpers for libfcosdet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/fcosdet.h"

void fd_set_error(const char* fmt, ...);

#define FD_REQUIRE(cond, code, ...)      \
    do {                                 \
        if (!(cond)) {                   \
            fd_set_error(__VA_ARGS__);   \
            return (code);               \
        }                                \
    } while (0)

#define FD_CHECK_LAUNCH(name)                                                      \
    do {                                                                           \
        hipError_t e__ = hipGetLastError();                                        \
        if (e__ != hipSuccess) {                                                   \
            fd_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));   \
            return FD_E_LAUNCH;                                                    \
        }                                                                          \
    } while (0)

// One-time per-device kernel attribute (dynamic LDS above 64 KiB): `mask` is a function-local static std::atomic<unsigned>,
// one bit per device ordinal; a second thread racing here merely sets the same attribute twice.  Keeps the entry points
// re-entrant and correct with several devices in one process.
#include <atomic>
static inline void fd_set_max_lds_once(std::atomic<unsigned>& mask, const void* kern, int bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const unsigned bit = 1u << (dev & 31);
    if (!(mask.load(std::memory_order_acquire) & bit)) {
        hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        mask.fetch_or(bit, std::memory_order_release);
    }
}

static inline int fd_segs_ok(const fd_segs* s) {
    if (!s || s->nseg < 1 || s->nseg > FD_MAX_SEG || s->batch < 1 || s->m_start[0] != 0) return 0;
    for (int i = 0; i < s->nseg; ++i) {
        if (s->H[i] < 1 || s->W[i] < 1) return 0;
        if (s->m_start[i + 1] - s->m_start[i] != s->batch * s->H[i] * s->W[i]) return 0;
    }
    return 1;
}

// i / d (returned) and i % d (rem) of a non-negative 64-bit element index by a positive 32-bit divisor.  Every launch in practice has fewer than 2^32
// elements: the unsigned 32-bit division is ~5x fewer instructions than the 64-bit one (this ISA has no integer divider), and the elementwise
// kernels pay it once per 16 bytes.
__device__ __forceinline__ long fd_div(long i, int d, int& rem) {
    if ((unsigned long)i <= 0xFFFFFFFFul) {
        const unsigned u = (unsigned)i, q = u / (unsigned)d;
        rem = (int)(u - q * (unsigned)d);
        return (long)q;
    }
    const long q = i / d;
    rem = (int)(i - q * d);
    return q;
}

__device__ __forceinline__ float fd_sigmoid(float v) { return 1.0f / (1.0f + expf(-v)); }

__device__ __forceinline__ float fd_act(float v, int act, float p) {
    switch (act) {
        case FD_ACT_RELU: return v > 0.f ? v : 0.f;
        // SiLU on the hardware exp / reciprocal (v_exp_f32, v_rcp_f32: ~2 ulp): the libm expf + IEEE divide cost ~25 VALU instructions per
        // element, which made the MBConv expand / depthwise epilogues (2.6 GB maps) VALU-bound; parity bar of these layers: 1e-4
        case FD_ACT_SILU: return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));   // (HIP's __fdividef is a plain IEEE divide)
        case FD_ACT_EXP: return expf(v * p);
        case FD_ACT_SIGMOID: return fd_sigmoid(v);
        default: return v;
    }
}
// fd_act with the common cases (`act` is uniform in every caller) decided by compares BEFORE the switch: at a site inside an unrolled epilogue the switch's exp /
// sigmoid bodies and the branches around them cost instruction fetch and scalar work (40 % of the AMP conv kernel, 8-12 % of the F(4x4) ReLU layers)
__device__ __forceinline__ float fd_act1(float v, int act, float p) {
    if (act == FD_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == FD_ACT_NONE) return v;
    if (act == FD_ACT_SILU) return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
    return fd_act(v, act, p);
}
__device__ __forceinline__ float4 fd_act4(const float4& v, int act, float p) {
    if (act == FD_ACT_RELU) return make_float4(v.x > 0.f ? v.x : 0.f, v.y > 0.f ? v.y : 0.f, v.z > 0.f ? v.z : 0.f, v.w > 0.f ? v.w : 0.f);
    if (act == FD_ACT_NONE) return v;
    if (act == FD_ACT_SILU)
        return make_float4(v.x * __builtin_amdgcn_rcpf(1.0f + __expf(-v.x)), v.y * __builtin_amdgcn_rcpf(1.0f + __expf(-v.y)),
                           v.z * __builtin_amdgcn_rcpf(1.0f + __expf(-v.z)), v.w * __builtin_amdgcn_rcpf(1.0f + __expf(-v.w)));
    return make_float4(fd_act(v.x, act, p), fd_act(v.y, act, p), fd_act(v.z, act, p), fd_act(v.w, act, p));
}
