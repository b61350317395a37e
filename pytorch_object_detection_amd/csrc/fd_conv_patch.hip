// fd_conv_patch.hip — 3x3 stride-1 'same' convolution (dilation 1 or 2) on gfx950 with the INPUT PATCH staged in LDS.
//
// The generic implicit-GEMM kernel (fd_conv.hip) walks K as (32-channel chunk, tap) and loads a fresh 128 x 32 A tile from
// global memory for each of the 9 taps: 9 x 16 KB per chunk and workgroup, i.e. every input pixel travels L2 -> LDS nine
// times (rocprof: the head tower fetches 4.8x its input).  Here a workgroup (128 output pixels x 128 channels) stages, per
// chunk, the pixels its tile can touch ONCE: output rows are consecutive pixels of an NHWC map, so with stride 1 / 'same'
// padding tap (dr, dq) of output row m is input row m + dr*W + dq and the whole tile needs the contiguous row range
// [m0 - halo, m0 + 128 + halo), halo = dil*(W + 1): 290 rows (37 KB) at W = 80 instead of 9 x 128.  The MFMA feed reads its
// A fragments from that patch at per-lane row offsets; taps that fall outside the image (or rows past M) are pointed at a
// zero row.  Weights stream per tap through a double-buffered 128 x 32 tile exactly as in the generic kernel; the epilogue is
// shared (fd_conv_epilogue.inc).  LDS: patch 41 KB + weights 32 KB -> 2 workgroups per CU: one computes while the other
// refills its patch.  SPLIT (FD_PREC_F16X3): the patch holds f16 hi / lo planes, so the fp32 -> (hi, lo) conversion runs once per
// staged element instead of once per tap.
#include "fd_conv_common.h"

template <int TAG, bool SPLIT>
__global__ __launch_bounds__(256, 2) void conv3x3_patch_kernel(ConvArgs a) {
    constexpr int TM = 2, TN = 2, WGN = 2, BM = FD_PATCH_BM, BN = 128, RPP = 32;
    constexpr int BP = BN / RPP;                         // weight rows per thread and tap
    constexpr int PP = FD_PATCH_MAXROWS / RPP;           // patch rows per thread and chunk (10)
    constexpr int PZ = FD_PATCH_MAXROWS;                 // index of the zero row
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // f32:   Ps [PZ + 1][32] floats | Bs [2][BN * 32] floats
    // SPLIT: Phi [PZ + 1][32] halves | Plo [PZ + 1][32] halves | per buffer Bhi [BN][32] | Blo [BN][32] halves
    float* Ps = reinterpret_cast<float*>(smem);
    float* Bs = Ps + (PZ + 1) * 32;
    _Float16* Phi = reinterpret_cast<_Float16*>(smem);
    _Float16* Plo = Phi + (PZ + 1) * 32;
    _Float16* Bh = Plo + (PZ + 1) * 32;                  // [2][2 * BN * 32]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int l31 = lane & 31, lh = lane >> 5;

    const int nblk = a.mtiles * a.ntiles;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int mt = bid / a.ntiles, nt = bid - mt * a.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const int halo = a.p_halo, prows = BM + 2 * halo;    // <= FD_PATCH_MAXROWS (host)
    const int dil = a.dil;

    const int lrow = tid >> 3, chunk = tid & 7;
    constexpr unsigned OOB = 0xC0000000u;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, (short)0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, (short)0, (int)a.w_bytes, 0x00020000);

    // ---- MFMA-side rows of this lane: patch index of the centre pixel, row pitch, 9-bit tap validity ----
    int pc[TM], wd[TM];
    unsigned vm[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = (wm * TM + i) * 32 + l31;
        const int m = m0 + r;
        int s = 0;
#pragma unroll
        for (int t = 1; t < FD_MAX_SEG; ++t)
            if (t < a.nseg && m >= a.m_out[t]) s = t;
        const int H = a.H[s], W = a.W[s];
        const int local = m - a.m_out[s];
        const int rem = local % (H * W);
        const int h = rem / W, w = rem - h * W;
        pc[i] = halo + r;
        wd[i] = dil * W;
        unsigned v = 0;
        if (m < a.M) {
#pragma unroll
            for (int tr = 0; tr < 3; ++tr)
#pragma unroll
                for (int tq = 0; tq < 3; ++tq) {
                    const int hh = h + (tr - 1) * dil, ww = w + (tq - 1) * dil;
                    if ((unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W) v |= 1u << (tr * 3 + tq);
                }
        }
        vm[i] = v;
    }

    // ---- loaders ----
    // patch row t <-> input row m0 - halo + t (input rows == output rows: stride 1, 'same'); rows before the buffer or past its
    // end are out of the raw buffer's range and read as zero (they are only ever addressed by taps the masks above disable)
    unsigned b_off[BP];
#pragma unroll
    for (int j = 0; j < BP; ++j) {
        const int n = n0 + lrow + RPP * j;
        b_off[j] = (n < a.Cout) ? ((unsigned)n * (unsigned)a.Kpacked + (unsigned)(chunk * 4)) * 4u : OOB;
    }
    const long prow0 = (long)(m0 - halo + lrow) * a.x_cs + a.x_co + chunk * 4;     // element index of patch row `lrow`, chunk 0
    float4 pr[PP], rb[BP];
    auto load_patch = [&](int cc) {
        const bool c_ok = cc * 32 + chunk * 4 < a.Cin;
#pragma unroll
        for (int j = 0; j < PP; ++j) {
            const int t = lrow + RPP * j;
            const long e = prow0 + (long)(RPP * j) * a.x_cs + cc * 32;
            const bool ok = c_ok && t < prows && e >= 0 && e * 4 < (long)a.x_bytes;
            pr[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)(ok ? (unsigned)(e * 4) : OOB), 0, 0));
        }
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int j = 0; j < PP; ++j) {
            const int t = lrow + RPP * j;
            if (t >= prows) continue;
            if constexpr (SPLIT) {
                const f32x4 v = {pr[j].x, pr[j].y, pr[j].z, pr[j].w};
                const h4 hi = __builtin_convertvector(v, h4);
                const f32x4 rem = (v - __builtin_convertvector(hi, f32x4)) * FD_SPLIT_SCALE;
                const h4 lo = __builtin_convertvector(rem, h4);
                const int off = t * 32 + ((((chunk >> 1) ^ ((t >> 2) & 3)) << 3) | ((chunk & 1) << 2));
                *reinterpret_cast<h4*>(Phi + off) = hi;
                *reinterpret_cast<h4*>(Plo + off) = lo;
            } else {
                *reinterpret_cast<float4*>(Ps + lds_off(t, chunk)) = pr[j];
            }
        }
    };
    auto load_b = [&](int kt) {
        const unsigned kb = (unsigned)kt * 128u;
#pragma unroll
        for (int j = 0; j < BP; ++j)
            rb[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (int)(b_off[j] + kb), 0, 0));
    };
    auto store_b = [&](int buf) {
#pragma unroll
        for (int j = 0; j < BP; ++j) {
            const int row = lrow + RPP * j;
            if constexpr (SPLIT) {   // weights arrive pre-split: 16-B chunks 0-3 = hi, 4-7 = lo of this K-tile
                _Float16* Bhi = Bh + buf * (2 * BN * 32);
                _Float16* Blo = Bhi + BN * 32;
                *reinterpret_cast<float4*>(((chunk & 4) ? Blo : Bhi) + lds_off_h(row, chunk & 3)) = rb[j];
            } else {
                *reinterpret_cast<float4*>(Bs + buf * BN * 32 + lds_off(row, chunk)) = rb[j];
            }
        }
    };

    f32x16 acc[TM][TN];
    f32x16 cor[SPLIT ? TM : 1][SPLIT ? TN : 1];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc[i][j][e] = 0.f;
                if constexpr (SPLIT) cor[i][j][e] = 0.f;
            }

    // zero row (read by disabled taps)
    if (tid < 8) {
        if constexpr (SPLIT) {       // 64-byte rows: 4 x 16 B per plane
            *reinterpret_cast<float4*>(((tid & 4) ? Plo : Phi) + PZ * 32 + (tid & 3) * 8) = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            *reinterpret_cast<float4*>(Ps + PZ * 32 + tid * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    float* const ybase = a.y;
    const int NC = (a.Cin + 31) >> 5;
    load_patch(0);
    load_b(0);
    store_patch();
    store_b(0);
    __syncthreads();
    int kt = 0, buf = 0;
    for (int cc = 0; cc < NC; ++cc) {
        int tr = 0, tq = 0;
        for (int tap = 0; tap < 9; ++tap) {
            const bool last = (cc == NC - 1) && (tap == 8);
            const bool refill = (tap == 8) && (cc + 1 < NC);
            if (!last) load_b(kt + 1);
            // the next chunk's patch: fp32 prefetches it into registers under this tap's MFMAs; the split-f16 variant (two
            // accumulator sets: no registers to spare, it spilled) issues the loads right after its MFMAs instead -- they fly while
            // the other waves finish and the CU's second workgroup computes
            if (refill && !SPLIT) load_patch(cc + 1);
            int pe[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                pe[i] = ((vm[i] >> tap) & 1u) ? pc[i] + (tr - 1) * wd[i] + (tq - 1) * dil : PZ;
            if constexpr (SPLIT) {
                const _Float16* Bhi = Bh + buf * (2 * BN * 32) + (wn * TN * 32) * 32;
                const _Float16* Blo = Bhi + BN * 32;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    h8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        ah[i] = *reinterpret_cast<const h8*>(Phi + lds_off_h(pe[i], 2 * ks + lh));
                        al[i] = *reinterpret_cast<const h8*>(Plo + lds_off_h(pe[i], 2 * ks + lh));
                    }
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        bh[j] = *reinterpret_cast<const h8*>(Bhi + lds_off_h(j * 32 + l31, 2 * ks + lh));
                        bl[j] = *reinterpret_cast<const h8*>(Blo + lds_off_h(j * 32 + l31, 2 * ks + lh));
                    }
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                            cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], cor[i][j], 0, 0, 0);
                            cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], cor[i][j], 0, 0, 0);
                        }
                }
            } else {
                const float* Bb = Bs + buf * BN * 32 + (wn * TN * 32) * 32;
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    float4 fa[TM], fb[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const float4*>(Ps + lds_off(pe[i], 2 * s + lh));
#pragma unroll
                    for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const float4*>(Bb + lds_off(j * 32 + l31, 2 * s + lh));
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
                        }
                }
                __builtin_amdgcn_s_setprio(0);
            }
            if (!last) store_b(buf ^ 1);
            if (refill) {
                if constexpr (SPLIT) load_patch(cc + 1);
                __syncthreads();           // every wave is done with this chunk's patch
                store_patch();
            }
            __syncthreads();
            buf ^= 1;
            ++kt;
            if (++tq == 3) { tq = 0; ++tr; }
        }
    }
    constexpr bool GNS = false;
    constexpr bool RUP = false;       // (no upsampled-residual form of this kernel)
    constexpr bool H1 = false;        // (no f16-storage form either)
#include "fd_conv_epilogue.inc"
}

template <int TAG, bool SPLIT>
static int launch_patch(const ConvArgs& a, hipStream_t stream) {
    constexpr int lds = SPLIT ? 2 * (FD_PATCH_MAXROWS + 1) * 32 * 2 + 2 * 2 * 128 * 32 * 2
                              : (FD_PATCH_MAXROWS + 1) * 32 * 4 + 2 * 128 * 32 * 4;
    ConvArgs b = a;
    b.mtiles = (a.M + FD_PATCH_BM - 1) / FD_PATCH_BM;
    b.ntiles = (a.Cout + 127) / 128;
    auto kern = conv3x3_patch_kernel<TAG, SPLIT>;
    static std::atomic<unsigned> attr_mask{0};
    fd_set_max_lds_once(attr_mask, reinterpret_cast<const void*>(kern), lds);
    hipLaunchKernelGGL(kern, dim3(b.mtiles * b.ntiles), dim3(256), lds, stream, b);
    FD_CHECK_LAUNCH("fd_conv2d_nhwc_f32 (patch)");
    return FD_OK;
}

int fd_launch_conv_patch(const ConvArgs& a, int tag, int split, hipStream_t stream) {
    if (split) return tag ? launch_patch<1, true>(a, stream) : launch_patch<0, true>(a, stream);
    return tag ? launch_patch<1, false>(a, stream) : launch_patch<0, false>(a, stream);
}
