// fd_postproc.hip — FCOS detection post-processing on gfx950 (wave64):
//   decode (sigmoid / max / sqrt / LTRB->xyxy)  ->  per-image radix-select top-k + bitonic sort
//   ->  score mask + class-offset batched NMS (bitmask in LDS, wave-level serial scan)  ->  clip.
// Restates reference model/modules/head.py:8-102,152-162 and utill/utills.py:58-73,201-255.
// Compiled with -ffp-contract=off: IoU arithmetic must round exactly like the CPU reference.
#include "fd_common.h"

// --------------------------------------------------------------------------------------------
// decode
// --------------------------------------------------------------------------------------------
struct DecodeArgs {
    const float* cls; const float* cnt; const float* reg;
    int cls_cs, cls_co, cnt_cs, cnt_co, reg_cs, reg_co;
    int C, L;
    int loc_start[FD_MAX_SEG + 1];
    int stride[FD_MAX_SEG];
    fd_segs segs;
    float* scores; int* classes; float* boxes;
};

__global__ __launch_bounds__(256) void decode_kernel(DecodeArgs a) {
    const int n = blockIdx.y;
    const int loc = blockIdx.x * 256 + threadIdx.x;
    if (loc >= a.L) return;
    int s = 0;
#pragma unroll
    for (int t = 1; t < FD_MAX_SEG; ++t)
        if (t < a.segs.nseg && loc >= a.loc_start[t]) s = t;
    const int pix = loc - a.loc_start[s];
    const int W = a.segs.W[s];
    const int hw = a.segs.H[s] * W;
    const long m = (long)a.segs.m_start[s] + (long)n * hw + pix;
    const int py = pix / W, px = pix - py * W;

    // max_c sigmoid(cls), first maximal index (torch.max semantics, head.py:62)
    const float* cp = a.cls + m * a.cls_cs + a.cls_co;
    float best = -1.0f;
    int besti = 0;
    const bool vec = ((a.cls_cs | a.cls_co | a.C) & 3) == 0;
    if (vec) {
        const float4* c4 = reinterpret_cast<const float4*>(cp);
        for (int c = 0; c < a.C / 4; ++c) {
            const float4 v = c4[c];
            const float s0 = fd_sigmoid(v.x), s1 = fd_sigmoid(v.y), s2 = fd_sigmoid(v.z), s3 = fd_sigmoid(v.w);
            if (s0 > best) { best = s0; besti = 4 * c; }
            if (s1 > best) { best = s1; besti = 4 * c + 1; }
            if (s2 > best) { best = s2; besti = 4 * c + 2; }
            if (s3 > best) { best = s3; besti = 4 * c + 3; }
        }
    } else {
        for (int c = 0; c < a.C; ++c) {
            const float sv = fd_sigmoid(cp[c]);
            if (sv > best) { best = sv; besti = c; }
        }
    }
    const float cen = fd_sigmoid(a.cnt[m * a.cnt_cs + a.cnt_co]);
    const float score = sqrtf(best * cen);

    const float* rp = a.reg + m * a.reg_cs + a.reg_co;
    const float l = rp[0], t = rp[1], r = rp[2], b = rp[3];
    const int st = a.stride[s];
    const float cx = (float)(px * st) + (float)(st / 2);
    const float cy = (float)(py * st) + (float)(st / 2);
    const long o = (long)n * a.L + loc;
    a.scores[o] = score;
    a.classes[o] = besti + 1;
    reinterpret_cast<float4*>(a.boxes)[o] = make_float4(cx - l, cy - t, cx + r, cy + b);
}

// Coalesced form (C % 4 == 0, C <= 256).  The one-thread-per-location kernel above is bound twice over: its loads touch 64
// cache lines per instruction (lanes 320 B apart), and it evaluates 80 sigmoids (expf + IEEE divide) per location --
// 10.9 M per 16-image batch, more VALU time than the 50 MB take to stream.  Here a wave owns DEC_LPW consecutive locations:
//   1. their class logits are copied global -> LDS as ONE flat run of float4 (lane l takes quads l, l + 64, ...: 1 KiB per
//      load instruction when rows are contiguous, all loads issued before the first LDS write);
//   2. one lane per location scans its row in LDS: L = max logit (compare-only), then sigmoid only for the classes whose
//      computed sigmoid could tie with or exceed the computed sigmoid(L).  sigmoid is monotonic and fd_sigmoid good to a few
//      ulp, so those are the logits within ~1e-6 (1 + e^L) of L: normally just the maximum itself (1 sigmoid per location
//      instead of C); equal or nearly equal logits, or a saturating L, widen the set -- up to every class -- and the first
//      maximal SIGMOID value wins exactly as in the plain kernel (torch.max semantics, head.py:61-62).
#define DEC_MAXQ 64
#define DEC_LPW 16     /* locations per wave: 64 / DEC_LPW lanes share one location in the scan */
__global__ __launch_bounds__(256) void decode_coalesced_kernel(DecodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) char dec_smem[];
    const int Q = a.C >> 2, QP = Q | 1;             // row stride in float4 (odd: conflict-free ds_read_b128 across lanes)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float4* rows = reinterpret_cast<float4*>(dec_smem) + wv * DEC_LPW * QP;
    const int n = blockIdx.y;
    const int loc0 = (blockIdx.x * 4 + wv) * DEC_LPW;
    if (loc0 >= a.L) return;                        // whole wave out of range (no workgroup barrier below)
    const int nloc = min(DEC_LPW, a.L - loc0);
    // lanes l, l + 16, l + 32, l + 48 share location l (each scans a quarter of its row below); lanes past nloc shadow the
    // last location and store nothing.  Centre-ness / regression loads are issued before the class sweep.
    constexpr int PARTS = 64 / DEC_LPW;
    const int lid = lane % DEC_LPW, part = lane / DEC_LPW;
    const int myloc = loc0 + min(lid, nloc - 1);
    int ms = 0;
#pragma unroll
    for (int t = 1; t < FD_MAX_SEG; ++t)
        if (t < a.segs.nseg && myloc >= a.loc_start[t]) ms = t;
    const int mypix = myloc - a.loc_start[ms];
    const int mW = a.segs.W[ms];
    const long mym = (long)a.segs.m_start[ms] + (long)n * (a.segs.H[ms] * mW) + mypix;
    const float cnt_logit = a.cnt[mym * a.cnt_cs + a.cnt_co];
    const float* rp = a.reg + mym * a.reg_cs + a.reg_co;
    const float rl = rp[0], rt = rp[1], rr = rp[2], rb = rp[3];

    const int items = nloc * Q;
    // DEC_U loads are issued back to back before the first LDS write (a plain loop waits for every load in turn: latency-
    // bound).  Item i = lane + 64*k -> (location i / Q, quad i % Q) advances without divisions: (lo, q) += (64 / Q, 64 % Q)
    // with a carry (integer divisions and 64-bit multiplies per item made this phase VALU-bound).
    constexpr int DEC_U = 5;
    const int d64 = 64 / Q, r64 = 64 - d64 * Q;
    int lo = lane / Q, q = lane - lo * Q;
    // the wave's locations normally sit in ONE pyramid level: its row base is wave-uniform (scalar), a location costs one add
    int s_first = 0, s_last = 0;
#pragma unroll
    for (int t = 1; t < FD_MAX_SEG; ++t) {
        if (t < a.segs.nseg && loc0 >= a.loc_start[t]) s_first = t;
        if (t < a.segs.nseg && loc0 + nloc - 1 >= a.loc_start[t]) s_last = t;
    }
    const bool one_level = s_first == s_last;
    const int row_base = a.segs.m_start[s_first] + n * (a.segs.H[s_first] * a.segs.W[s_first]) - a.loc_start[s_first] + loc0;
    for (int i0 = lane; i0 < items; i0 += 64 * DEC_U) {
        float4 v[DEC_U];
        int slot[DEC_U];
#pragma unroll
        for (int u = 0; u < DEC_U; ++u) {
            const bool ok = i0 + 64 * u < items;
            const int lc = ok ? lo : nloc - 1, qc = ok ? q : 0;     // (clamped: the load stays in range, the store is predicated)
            int row = row_base + lc;                                                                     // < 2^31 rows
            if (!one_level) {                                    // (uniform branch: a wave that straddles a level boundary)
                const int loc = loc0 + lc;
                int s = 0;
#pragma unroll
                for (int t = 1; t < FD_MAX_SEG; ++t)
                    if (t < a.segs.nseg && loc >= a.loc_start[t]) s = t;
                row = a.segs.m_start[s] + n * (a.segs.H[s] * a.segs.W[s]) + (loc - a.loc_start[s]);
            }
            v[u] = *reinterpret_cast<const float4*>(a.cls + (long)row * a.cls_cs + a.cls_co + 4 * qc);
            slot[u] = ok ? lc * QP + qc : -1;
            q += r64; lo += d64;
            if (q >= Q) { q -= Q; ++lo; }
        }
#pragma unroll
        for (int u = 0; u < DEC_U; ++u)
            if (slot[u] >= 0) rows[slot[u]] = v[u];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): this wave's LDS writes have landed
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
        // one pass over the row, split between the two lanes of a location: largest logit L with its FIRST index, and the
        // second largest value (compare-only)
        const float4* row = rows + lid * QP;
        const int q0 = part * Q / PARTS, q1 = (part + 1) * Q / PARTS;
        float L = -INFINITY, L2 = -INFINITY;
        int li = 0;
        for (int q = q0; q < q1; ++q) {
            const float4 v = row[q];
            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (e[c] > L) { L2 = L; L = e[c]; li = 4 * q + c; }
                else L2 = fmaxf(L2, e[c]);
            }
        }
#pragma unroll
        for (int o = DEC_LPW; o < 64; o <<= 1) {         // combine the parts; the part with the lower class indices wins ties
            const float oL = __shfl_xor(L, o), oL2 = __shfl_xor(L2, o);
            const int oli = __shfl_xor(li, o);
            const bool upper = (lane & o) != 0;
            const bool take_other = upper ? (oL >= L) : (oL > L);
            L2 = fmaxf(fmaxf(L2, oL2), take_other ? L : oL);
            if (take_other) { L = oL; li = oli; }
        }
        // classes whose COMPUTED sigmoid could reach the computed sigmoid(L): fd_sigmoid is good to a few ulp, so only logits
        // within ~8 ulp(s) / s'(L) = 9.5e-7 (1 + e^L) of L (doubled below) -- just the maximum unless logits are (nearly) equal
        const float cut = (L < 80.0f) ? L - 2.0e-6f * (1.0f + expf(L)) : -INFINITY;   // (e^L overflows: every class is a candidate)
        float best = -1.0f;
        int besti = 0;
        if (L2 < cut) {
            best = fd_sigmoid(L);
            besti = li;
        } else if (part == 0 && lid < nloc) {           // rare: evaluate every candidate, first maximal sigmoid wins (head.py:62)
            for (int q = 0; q < Q; ++q) {
                const float4 v = row[q];
                if (v.x >= cut) { const float sv = fd_sigmoid(v.x); if (sv > best) { best = sv; besti = 4 * q; } }
                if (v.y >= cut) { const float sv = fd_sigmoid(v.y); if (sv > best) { best = sv; besti = 4 * q + 1; } }
                if (v.z >= cut) { const float sv = fd_sigmoid(v.z); if (sv > best) { best = sv; besti = 4 * q + 2; } }
                if (v.w >= cut) { const float sv = fd_sigmoid(v.w); if (sv > best) { best = sv; besti = 4 * q + 3; } }
            }
        }                                               // a row of NaNs: L = -inf, nothing >= cut -> best = -1, class 1 (plain kernel)
        if (part != 0 || lid >= nloc) return;
        const float score = sqrtf(best * fd_sigmoid(cnt_logit));
        const int py = mypix / mW, px = mypix - py * mW;
        const int st = a.stride[ms];
        const float cx = (float)(px * st) + (float)(st / 2);
        const float cy = (float)(py * st) + (float)(st / 2);
        const long o = (long)n * a.L + myloc;
        a.scores[o] = score;
        a.classes[o] = besti + 1;
        reinterpret_cast<float4*>(a.boxes)[o] = make_float4(cx - rl, cy - rt, cx + rr, cy + rb);
    }
}

extern "C" int32_t fd_fcos_decode(const float* cls, int32_t cls_cs, int32_t cls_co, const float* cnt,
                                  int32_t cnt_cs, int32_t cnt_co, const float* reg, int32_t reg_cs,
                                  int32_t reg_co, int32_t num_classes, const fd_segs* segs,
                                  const int32_t* strides, float* scores, int32_t* classes, float* boxes,
                                  fd_stream_t stream) {
    FD_REQUIRE(cls && cnt && reg && scores && classes && boxes && strides, FD_E_INVAL, "fd_fcos_decode: null pointer");
    FD_REQUIRE(fd_segs_ok(segs), FD_E_INVAL, "fd_fcos_decode: bad segment table");
    FD_REQUIRE(num_classes >= 1, FD_E_INVAL, "fd_fcos_decode: num_classes < 1");
    FD_REQUIRE(((uintptr_t)boxes & 15) == 0, FD_E_INVAL, "fd_fcos_decode: boxes not 16-byte aligned");
    DecodeArgs a;
    a.cls = cls; a.cnt = cnt; a.reg = reg;
    a.cls_cs = cls_cs; a.cls_co = cls_co; a.cnt_cs = cnt_cs; a.cnt_co = cnt_co; a.reg_cs = reg_cs; a.reg_co = reg_co;
    a.C = num_classes;
    a.segs = *segs;
    int L = 0;
    for (int s = 0; s < segs->nseg; ++s) {
        a.loc_start[s] = L;
        a.stride[s] = strides[s];
        L += segs->H[s] * segs->W[s];
    }
    for (int s = segs->nseg; s < FD_MAX_SEG; ++s) { a.loc_start[s] = L; a.stride[s] = 1; }
    a.loc_start[FD_MAX_SEG] = L;
    a.L = L;
    a.scores = scores; a.classes = classes; a.boxes = boxes;
    if (num_classes % 4 == 0 && num_classes <= 4 * DEC_MAXQ && cls_cs % 4 == 0 && cls_co % 4 == 0 && ((uintptr_t)cls & 15) == 0) {
        const int QP = (num_classes / 4) | 1;
        const int lds = 4 * DEC_LPW * QP * 16;       // 4 waves x DEC_LPW locations x QP float4
        static std::atomic<unsigned> attr_mask{0};
        fd_set_max_lds_once(attr_mask, reinterpret_cast<const void*>(decode_coalesced_kernel), 4 * DEC_LPW * 65 * 16);
        dim3 grid((L + 4 * DEC_LPW - 1) / (4 * DEC_LPW), segs->batch);
        hipLaunchKernelGGL(decode_coalesced_kernel, grid, dim3(256), lds, (hipStream_t)stream, a);
    } else {
        dim3 grid((L + 255) / 256, segs->batch);
        hipLaunchKernelGGL(decode_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    }
    FD_CHECK_LAUNCH("fd_fcos_decode");
    return FD_OK;
}

// --------------------------------------------------------------------------------------------
// top-k: radix select of the K-th largest key, ordered compaction of ties, bitonic sort in LDS
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned fd_order_key(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// descending bitonic sort of 1024 u64 in LDS by 1024 threads
__device__ __forceinline__ void bitonic_desc_1024(unsigned long long* c, int tid) {
    for (int k = 2; k <= 1024; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int ixj = tid ^ j;
            if (ixj > tid) {
                const unsigned long long x = c[tid], y = c[ixj];
                const bool desc = (tid & k) == 0;
                if (desc ? (x < y) : (x > y)) { c[tid] = y; c[ixj] = x; }
            }
            __syncthreads();
        }
    }
}

// descending bitonic sort of 1024 u64, ONE element per thread in a register: exchanges with a partner in the same wave
// (j < 64) go through __shfl_xor, only the 10 steps with j >= 64 go through LDS (the all-LDS form above takes 55 barriers)
__device__ __forceinline__ unsigned long long bitonic_desc_1024_reg(unsigned long long x, unsigned long long* c, int tid) {
    for (int k = 2; k <= 1024; k <<= 1) {
        const bool desc = (tid & k) == 0;
        for (int j = k >> 1; j > 0; j >>= 1) {
            unsigned long long y;
            if (j >= 64) {
                c[tid] = x;
                __syncthreads();
                y = c[tid ^ j];
                __syncthreads();
            } else {
                const unsigned lo = __shfl_xor((unsigned)x, j), hi = __shfl_xor((unsigned)(x >> 32), j);
                y = ((unsigned long long)hi << 32) | lo;
            }
            const bool lower = (tid & j) == 0;
            x = (lower == desc) ? (x > y ? x : y) : (x < y ? x : y);
        }
    }
    return x;
}

// Radix select of the K-th largest key in THREE passes (11 + 11 + 10 bits, 2048-bin LDS histogram; the digit is found by a
// parallel suffix scan, not a serial walk), keys held in registers across the passes when L <= TOPK_MAXE * 1024 (one global
// read of the scores instead of five), ordered compaction (ties: lower index first), register bitonic sort.
#define TOPK_MAXE 24
template <bool REG>
__global__ __launch_bounds__(1024) void topk_kernel(const float* __restrict__ scores, const int* __restrict__ classes,
                                                     const float* __restrict__ boxes, int L, int K,
                                                     float* top_scores, long long* top_classes, float* top_boxes,
                                                     int* top_idx) {
    __shared__ unsigned hist[2048];
    __shared__ unsigned wsum[16];
    __shared__ unsigned sh_prefix, sh_need, sh_gt, sh_eqbase;
    __shared__ unsigned wave_cnt[16];
    __shared__ unsigned long long cand[1024];
    const int n = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* s = scores + (long)n * L;
    const int nE = (L + 1023) >> 10;

    unsigned key[REG ? TOPK_MAXE : 1];
    if constexpr (REG) {
#pragma unroll
        for (int e = 0; e < TOPK_MAXE; ++e) {
            const int i = e * 1024 + tid;
            key[e] = (e < nE && i < L) ? fd_order_key(s[i]) : 0u;
        }
    }
    auto key_at = [&](int e, int i) -> unsigned { if constexpr (REG) return key[e]; else return fd_order_key(s[i]); };

    unsigned prefix = 0, maskbits = 0, need = (unsigned)K;
    const int shifts[3] = {21, 10, 0}, widths[3] = {11, 11, 10};
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
        const int sh = shifts[pass];
        const unsigned dmask = (1u << widths[pass]) - 1u;
        hist[tid] = 0; hist[tid + 1024] = 0;
        __syncthreads();
        if constexpr (REG) {
#pragma unroll
            for (int e = 0; e < TOPK_MAXE; ++e) {
                const int i = e * 1024 + tid;
                if (e < nE && i < L && (key[e] & maskbits) == prefix) atomicAdd(&hist[(key[e] >> sh) & dmask], 1u);
            }
        } else {
            for (int i = tid; i < L; i += 1024) {
                const unsigned kk = fd_order_key(s[i]);
                if ((kk & maskbits) == prefix) atomicAdd(&hist[(kk >> sh) & dmask], 1u);
            }
        }
        __syncthreads();
        // thread t owns digits d0 = 2047 - 2t and d0 - 1 (descending); inclusive scan of the pair sums over threads
        const int d0 = 2047 - 2 * tid;
        const unsigned h0 = hist[d0], h1 = hist[d0 - 1];
        unsigned incl = h0 + h1;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        unsigned before = 0;
        for (int w = 0; w < wv; ++w) before += wsum[w];
        const unsigned excl = before + incl - (h0 + h1);
        if (excl < need && need <= excl + h0 + h1) {          // exactly one thread: the K-th key's digit is here
            if (need <= excl + h0) { sh_prefix = prefix | ((unsigned)d0 << sh); sh_need = need - excl; }
            else { sh_prefix = prefix | ((unsigned)(d0 - 1) << sh); sh_need = need - excl - h0; }
        }
        __syncthreads();
        prefix = sh_prefix;
        need = sh_need;
        maskbits |= dmask << sh;
        __syncthreads();
    }
    const unsigned T = prefix;                 // K-th largest key
    const unsigned n_gt = (unsigned)K - need;  // keys strictly above T; 'need' ties are taken lowest-index-first

    // ---- compaction ----
    cand[tid] = 0ull;
    if (tid == 0) { sh_gt = 0; sh_eqbase = 0; }
    __syncthreads();
    for (int e = 0; e < nE; ++e) {
        const int i = e * 1024 + tid;
        unsigned kk = 0;
        bool gt = false, eq = false;
        if (i < L) {
            kk = REG ? key_at(e < TOPK_MAXE ? e : 0, i) : fd_order_key(s[i]);
            gt = kk > T;
            eq = kk == T;
        }
        if (gt) {
            const unsigned slot = atomicAdd(&sh_gt, 1u);
            cand[slot] = ((unsigned long long)kk << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
        }
        const unsigned long long bal = __ballot(eq);
        if (lane == 0) wave_cnt[wv] = (unsigned)__popcll(bal);
        __syncthreads();
        unsigned before = sh_eqbase;
        for (int w = 0; w < wv; ++w) before += wave_cnt[w];
        const unsigned rank = before + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
        if (eq && rank < need)
            cand[n_gt + rank] = ((unsigned long long)kk << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
        __syncthreads();
        if (tid == 0) {
            unsigned tot = 0;
            for (int w = 0; w < 16; ++w) tot += wave_cnt[w];
            sh_eqbase += tot;
        }
        __syncthreads();
    }

    // ---- sort (score desc, index asc) ----
    const unsigned long long mine = bitonic_desc_1024_reg(cand[tid], cand, tid);

    if (tid < K) {
        const unsigned idx = 0xFFFFFFFFu - (unsigned)(mine & 0xFFFFFFFFull);
        const long o = (long)n * K + tid;
        const long src = (long)n * L + idx;
        top_scores[o] = s[idx];
        top_classes[o] = (long long)classes[src];
        reinterpret_cast<float4*>(top_boxes)[o] = reinterpret_cast<const float4*>(boxes)[src];
        if (top_idx) top_idx[o] = (int)idx;
    }
}

// ---- K > 1024 (FCOSHead(max_detection_box > 1024): the reference accepts any value, model/modules/head.py:41-50): the same radix
// select and the same order (score descending, ties by lower index), with the candidate list and its bitonic sort in a global
// scratch of Kpad = next power of two >= K entries per image instead of LDS / registers.  One 1024-thread workgroup per image.
__global__ __launch_bounds__(1024) void topk_large_kernel(const float* __restrict__ scores, const int* __restrict__ classes,
                                                           const float* __restrict__ boxes, int L, int K, int Kpad,
                                                           unsigned long long* __restrict__ cand_ws, float* top_scores,
                                                           long long* top_classes, float* top_boxes, int* top_idx) {
    __shared__ unsigned hist[2048];
    __shared__ unsigned wsum[16];
    __shared__ unsigned sh_prefix, sh_need, sh_gt, sh_eqbase;
    __shared__ unsigned wave_cnt[16];
    const int n = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* s = scores + (long)n * L;
    unsigned long long* cand = cand_ws + (long)n * Kpad;
    const int nE = (L + 1023) >> 10;

    unsigned prefix = 0, maskbits = 0, need = (unsigned)K;
    const int shifts[3] = {21, 10, 0}, widths[3] = {11, 11, 10};
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
        const int sh = shifts[pass];
        const unsigned dmask = (1u << widths[pass]) - 1u;
        hist[tid] = 0; hist[tid + 1024] = 0;
        __syncthreads();
        for (int i = tid; i < L; i += 1024) {
            const unsigned kk = fd_order_key(s[i]);
            if ((kk & maskbits) == prefix) atomicAdd(&hist[(kk >> sh) & dmask], 1u);
        }
        __syncthreads();
        const int d0 = 2047 - 2 * tid;
        const unsigned h0 = hist[d0], h1 = hist[d0 - 1];
        unsigned incl = h0 + h1;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        unsigned before = 0;
        for (int w = 0; w < wv; ++w) before += wsum[w];
        const unsigned excl = before + incl - (h0 + h1);
        if (excl < need && need <= excl + h0 + h1) {
            if (need <= excl + h0) { sh_prefix = prefix | ((unsigned)d0 << sh); sh_need = need - excl; }
            else { sh_prefix = prefix | ((unsigned)(d0 - 1) << sh); sh_need = need - excl - h0; }
        }
        __syncthreads();
        prefix = sh_prefix;
        need = sh_need;
        maskbits |= dmask << sh;
        __syncthreads();
    }
    const unsigned T = prefix;
    const unsigned n_gt = (unsigned)K - need;

    for (int i = tid; i < Kpad; i += 1024) cand[i] = 0ull;
    if (tid == 0) { sh_gt = 0; sh_eqbase = 0; }
    __syncthreads();
    for (int e = 0; e < nE; ++e) {
        const int i = e * 1024 + tid;
        unsigned kk = 0;
        bool gt = false, eq = false;
        if (i < L) { kk = fd_order_key(s[i]); gt = kk > T; eq = kk == T; }
        if (gt) {
            const unsigned slot = atomicAdd(&sh_gt, 1u);
            cand[slot] = ((unsigned long long)kk << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
        }
        const unsigned long long bal = __ballot(eq);
        if (lane == 0) wave_cnt[wv] = (unsigned)__popcll(bal);
        __syncthreads();
        unsigned before = sh_eqbase;
        for (int w = 0; w < wv; ++w) before += wave_cnt[w];
        const unsigned rank = before + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
        if (eq && rank < need)
            cand[n_gt + rank] = ((unsigned long long)kk << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
        __syncthreads();
        if (tid == 0) {
            unsigned tot = 0;
            for (int w = 0; w < 16; ++w) tot += wave_cnt[w];
            sh_eqbase += tot;
        }
        __syncthreads();
    }
    // bitonic sort, descending, in the global scratch (writes of a workgroup are visible to it after its barrier)
    for (int k = 2; k <= Kpad; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < Kpad; i += 1024) {
                const int p = i ^ j;
                if (p > i) {
                    const unsigned long long a = cand[i], b = cand[p];
                    const bool desc = (i & k) == 0;
                    if (desc ? (a < b) : (a > b)) { cand[i] = b; cand[p] = a; }
                }
            }
            __syncthreads();
        }
    for (int t = tid; t < K; t += 1024) {
        const unsigned idx = 0xFFFFFFFFu - (unsigned)(cand[t] & 0xFFFFFFFFull);
        const long o = (long)n * K + t;
        const long src = (long)n * L + idx;
        top_scores[o] = s[idx];
        top_classes[o] = (long long)classes[src];
        reinterpret_cast<float4*>(top_boxes)[o] = reinterpret_cast<const float4*>(boxes)[src];
        if (top_idx) top_idx[o] = (int)idx;
    }
}

static int topk_kpad(int K) { int p = 1; while (p < K) p <<= 1; return p; }

extern "C" int64_t fd_topk_workspace_bytes(int32_t N, int32_t L, int32_t K) {
    if (N < 1 || L < 1 || K < 1 || K > L) return -1;
    return K <= 1024 ? 0 : (int64_t)N * topk_kpad(K) * (int64_t)sizeof(unsigned long long);
}

extern "C" int32_t fd_fcos_topk(const float* scores, const int32_t* classes, const float* boxes, int32_t N,
                                int32_t L, int32_t K, float* top_scores, int64_t* top_classes, float* top_boxes,
                                int32_t* top_idx, void* workspace, fd_stream_t stream) {
    FD_REQUIRE(scores && classes && boxes && top_scores && top_classes && top_boxes, FD_E_INVAL,
               "fd_fcos_topk: null pointer");
    FD_REQUIRE(N >= 1 && L >= 1 && K >= 1 && K <= L, FD_E_INVAL, "fd_fcos_topk: need 1 <= K <= L (K=%d L=%d)", K, L);
    FD_REQUIRE((((uintptr_t)boxes | (uintptr_t)top_boxes) & 15) == 0, FD_E_INVAL, "fd_fcos_topk: boxes not 16-byte aligned");
    if (K > 1024) {
        FD_REQUIRE(workspace && ((uintptr_t)workspace & 7) == 0, FD_E_INVAL, "fd_fcos_topk: K=%d > 1024 needs a workspace of fd_topk_workspace_bytes() bytes", K);
        hipLaunchKernelGGL(topk_large_kernel, dim3(N), dim3(1024), 0, (hipStream_t)stream, scores, classes, boxes, L, K, topk_kpad(K),
                           (unsigned long long*)workspace, top_scores, (long long*)top_classes, top_boxes, top_idx);
        FD_CHECK_LAUNCH("fd_fcos_topk (K > 1024)");
        return FD_OK;
    }
    if (L <= TOPK_MAXE * 1024)
        hipLaunchKernelGGL(topk_kernel<true>, dim3(N), dim3(1024), 0, (hipStream_t)stream, scores, classes, boxes, L, K,
                           top_scores, (long long*)top_classes, top_boxes, top_idx);
    else
        hipLaunchKernelGGL(topk_kernel<false>, dim3(N), dim3(1024), 0, (hipStream_t)stream, scores, classes, boxes, L, K,
                           top_scores, (long long*)top_classes, top_boxes, top_idx);
    FD_CHECK_LAUNCH("fd_fcos_topk");
    return FD_OK;
}

// --------------------------------------------------------------------------------------------
// greedy NMS: suppression bitmask (upper triangle) in LDS, then a wave-level serial scan
// --------------------------------------------------------------------------------------------
#define NMS_MAXK 1024
#define NMS_NW (NMS_MAXK / 64)

struct NmsShared {
    unsigned long long mask[NMS_MAXK * NMS_NW];  // 128 KiB
    float4 box[NMS_MAXK];                        // 16 KiB (class-offset boxes)
    float area[NMS_MAXK];                        // 4 KiB
    unsigned long long remv[NMS_NW];
    unsigned long long kept[NMS_NW];
    float red[16];
    int n_valid;
};

// PLUS1 = false: torchvision nms rule, suppress when (double)ovr > thr_d
// PLUS1 = true : DataEncoder._box_nms rule (utills.py:221-255), suppress when ovr > thr_f ; mode 1 = 'min'
// greedy pass over the LDS bitmask: wave 0 resolves each 64-box block's dependency chain on SGPR state, then the
// other waves OR the kept rows into the later words.  Needs sh.mask filled and sh.remv / sh.kept zeroed.
__device__ __forceinline__ void nms_scan(NmsShared& sh, int n, int tid) {
    const int nw = (n + 63) >> 6;
    // ---- serial scan, one 64-box block at a time ----
    const int lane = tid & 63, wv = tid >> 6;
    for (int blk = 0; blk < nw; ++blk) {
        if (wv == 0) {
            const int row = (blk << 6) + lane;
            const unsigned long long diag = (row < n) ? sh.mask[row * NMS_NW + blk] : 0ull;
            const unsigned dlo = (unsigned)diag, dhi = (unsigned)(diag >> 32);
            const unsigned long long cur0 = sh.remv[blk];
            unsigned clo = __builtin_amdgcn_readfirstlane((unsigned)cur0);
            unsigned chi = __builtin_amdgcn_readfirstlane((unsigned)(cur0 >> 32));
            const int nb = min(64, n - (blk << 6));
            unsigned klo = 0, khi = 0;
#pragma unroll
            for (int b = 0; b < 32; ++b) {
                if (b < nb && !((clo >> b) & 1u)) {
                    klo |= 1u << b;
                    clo |= __builtin_amdgcn_readlane(dlo, b);
                    chi |= __builtin_amdgcn_readlane(dhi, b);
                }
            }
#pragma unroll
            for (int b = 0; b < 32; ++b) {
                if (b + 32 < nb && !((chi >> b) & 1u)) {
                    khi |= 1u << b;
                    chi |= __builtin_amdgcn_readlane(dhi, b + 32);
                }
            }
            if (lane == 0) sh.kept[blk] = ((unsigned long long)khi << 32) | klo;
        }
        __syncthreads();
        // OR the rows of the kept boxes of this block into the later words: wave w handles word w
        if (wv > blk && wv < nw) {
            const unsigned long long k = sh.kept[blk];
            const int row = (blk << 6) + lane;
            unsigned long long v = ((k >> lane) & 1ull) ? sh.mask[row * NMS_NW + wv] : 0ull;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const unsigned lo = __shfl_xor((unsigned)v, o);
                const unsigned hi = __shfl_xor((unsigned)(v >> 32), o);
                v |= ((unsigned long long)hi << 32) | lo;
            }
            if (lane == 0) sh.remv[wv] |= v;
        }
        __syncthreads();
    }
}

template <bool PLUS1>
__device__ __forceinline__ void nms_core(NmsShared& sh, int n, double thr_d, float thr_f, int mode, int tid) {
    const int nw = (n + 63) >> 6;
    // ---- suppression mask, upper triangle ----
    for (int idx = tid; idx < n * nw; idx += 1024) {
        const int i = idx / nw, w = idx - i * nw;
        if (w < (i >> 6)) continue;
        const float4 bi = sh.box[i];
        const float ai = sh.area[i];
        unsigned long long bits = 0ull;
        const int j0 = w << 6;
        const int jend = min(64, n - j0);
        for (int b = 0; b < jend; ++b) {
            const int j = j0 + b;
            if (j <= i) continue;
            const float4 bj = sh.box[j];
            const float xx1 = fmaxf(bi.x, bj.x), yy1 = fmaxf(bi.y, bj.y);
            const float xx2 = fminf(bi.z, bj.z), yy2 = fminf(bi.w, bj.w);
            float ww, hh;
            if (PLUS1) { ww = fmaxf((xx2 - xx1) + 1.0f, 0.0f); hh = fmaxf((yy2 - yy1) + 1.0f, 0.0f); }
            else       { ww = fmaxf(0.0f, xx2 - xx1);          hh = fmaxf(0.0f, yy2 - yy1); }
            const float inter = ww * hh;
            float ovr;
            if (PLUS1 && mode == 1) ovr = inter / fminf(sh.area[j], ai);
            else                    ovr = inter / ((ai + sh.area[j]) - inter);
            const bool sup = PLUS1 ? (ovr > thr_f) : ((double)ovr > thr_d);
            if (sup) bits |= 1ull << b;
        }
        sh.mask[i * NMS_NW + w] = bits;
    }
    if (tid < NMS_NW) { sh.remv[tid] = 0ull; sh.kept[tid] = 0ull; }
    __syncthreads();

    nms_scan(sh, n, tid);
}

__device__ __forceinline__ int nms_rank(const NmsShared& sh, int r) {
    int rank = 0;
    const int w = r >> 6;
    for (int q = 0; q < w; ++q) rank += __popcll(sh.kept[q]);
    rank += __popcll(sh.kept[w] & ((1ull << (r & 63)) - 1ull));
    return rank;
}

// ---- batched NMS, multi-block: (1) every (image, 64-row block) builds its slice of the suppression bitmask on its
// own CU and writes it to a global workspace, (2) one workgroup per image pulls the mask into LDS and runs the scan.
struct NmsPrepShared {
    float4 box[NMS_MAXK];
    float area[NMS_MAXK];
    float red[16];
    int n_valid;
};

// valid prefix {score >= thr} = [0, n) (rows are score-descending), boxes.max() over it, class-offset boxes + areas
// (torchvision _batched_nms_coordinate_trick).  blockDim.x threads cooperate; returns n.
template <typename SH>
__device__ __forceinline__ int nms_prep(SH& sh, const float* __restrict__ scores, const long long* __restrict__ classes,
                                        const float* __restrict__ boxes, long base, int K, float score_thr, int tid, int nthr) {
    if (tid == 0) sh.n_valid = 0;
    if (tid < 16) sh.red[tid] = -INFINITY;
    __syncthreads();
    int cnt = 0;
    for (int i = tid; i < K; i += nthr) cnt += (scores[base + i] >= score_thr) ? 1 : 0;
    if (cnt) atomicAdd(&sh.n_valid, cnt);
    __syncthreads();
    const int n = sh.n_valid;
    float mx = -INFINITY;
    for (int i = tid; i < n; i += nthr) {
        const float4 bx = reinterpret_cast<const float4*>(boxes)[base + i];
        mx = fmaxf(mx, fmaxf(fmaxf(bx.x, bx.y), fmaxf(bx.z, bx.w)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((tid & 63) == 0) sh.red[tid >> 6] = mx;
    __syncthreads();
    float maxc = sh.red[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) maxc = fmaxf(maxc, sh.red[w]);
    for (int i = tid; i < n; i += nthr) {
        const float4 bx = reinterpret_cast<const float4*>(boxes)[base + i];
        const float off = (float)classes[base + i] * (maxc + 1.0f);
        const float4 ob = make_float4(bx.x + off, bx.y + off, bx.z + off, bx.w + off);
        sh.box[i] = ob;
        sh.area[i] = (ob.z - ob.x) * (ob.w - ob.y);
    }
    __syncthreads();
    return n;
}

__device__ __forceinline__ unsigned long long nms_mask_word(const float4* box, const float* area, int i, int w, int n,
                                                             double thr_d) {
    const float4 bi = box[i];
    const float ai = area[i];
    unsigned long long bits = 0ull;
    const int j0 = w << 6;
    const int jend = min(64, n - j0);
    for (int b = 0; b < jend; ++b) {
        const int j = j0 + b;
        if (j <= i) continue;
        const float4 bj = box[j];
        const float xx1 = fmaxf(bi.x, bj.x), yy1 = fmaxf(bi.y, bj.y);
        const float xx2 = fminf(bi.z, bj.z), yy2 = fminf(bi.w, bj.w);
        const float ww = fmaxf(0.0f, xx2 - xx1), hh = fmaxf(0.0f, yy2 - yy1);
        const float inter = ww * hh;
        const float ovr = inter / ((ai + area[j]) - inter);
        if ((double)ovr > thr_d) bits |= 1ull << b;
    }
    return bits;
}

__global__ __launch_bounds__(256) void nms_mask_kernel(const float* __restrict__ scores, const long long* __restrict__ classes,
                                                        const float* __restrict__ boxes, int K, float score_thr, double iou_thr,
                                                        unsigned long long* __restrict__ mask_ws) {
    __shared__ NmsPrepShared sh;
    const int img = blockIdx.y, rb = blockIdx.x, tid = threadIdx.x;
    const long base = (long)img * K;
    const int n = nms_prep(sh, scores, classes, boxes, base, K, score_thr, tid, 256);
    const int r0 = rb << 6;
    if (r0 >= n) return;
    const int nw = (n + 63) >> 6;
    const int nwork = 64 * (nw - rb);  // (row, word >= rb) pairs of this row block
    unsigned long long* out = mask_ws + (long)img * NMS_MAXK * NMS_NW;
    for (int idx = tid; idx < nwork; idx += 256) {
        const int w = rb + idx / 64, i = r0 + (idx & 63);   // a wave shares the word, lanes take consecutive rows
        if (i < n) out[i * NMS_NW + w] = nms_mask_word(sh.box, sh.area, i, w, n, iou_thr);
    }
}

__global__ __launch_bounds__(1024) void nms_scan_kernel(const float* __restrict__ scores, const long long* __restrict__ classes,
                                                         const float* __restrict__ boxes, int K, float score_thr,
                                                         const unsigned long long* __restrict__ mask_ws, float* out_scores,
                                                         long long* out_classes, float* out_boxes, int* keep_idx, int* counts) {
    __shared__ NmsShared sh;
    const int img = blockIdx.x, tid = threadIdx.x;
    const long base = (long)img * K;
    const int lane = tid & 63, wv = tid >> 6;
    if (tid == 0) sh.n_valid = 0;
    __syncthreads();
    float sc = 0.f;
    bool ok = false;
    if (tid < K) { sc = scores[base + tid]; ok = sc >= score_thr; }
    const unsigned long long bal = __ballot(ok);
    if (lane == 0 && bal) atomicAdd(&sh.n_valid, __popcll(bal));
    __syncthreads();
    const int n = sh.n_valid;
    const int nw = (n + 63) >> 6;
    // pull the upper-triangular mask into LDS (coalesced 8-byte loads)
    const unsigned long long* src = mask_ws + (long)img * NMS_MAXK * NMS_NW;
    for (int idx = tid; idx < n * NMS_NW; idx += 1024) {
        const int i = idx >> 4, w = idx & 15;
        sh.mask[idx] = (w >= (i >> 6) && w < nw) ? src[idx] : 0ull;
    }
    if (tid < NMS_NW) { sh.remv[tid] = 0ull; sh.kept[tid] = 0ull; }
    __syncthreads();
    if (n > 0) nms_scan(sh, n, tid);

    int total = 0;
    for (int q = 0; q < nw; ++q) total += __popcll(sh.kept[q]);
    if (tid < n && ((sh.kept[tid >> 6] >> (tid & 63)) & 1ull)) {
        const int r = nms_rank(sh, tid);
        out_scores[base + r] = sc;
        out_classes[base + r] = classes[base + tid];
        reinterpret_cast<float4*>(out_boxes)[base + r] = reinterpret_cast<const float4*>(boxes)[base + tid];
        keep_idx[base + r] = tid;
    }
    if (tid < K && tid >= total) {
        out_scores[base + tid] = 0.f;
        out_classes[base + tid] = 0;
        reinterpret_cast<float4*>(out_boxes)[base + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
        keep_idx[base + tid] = -1;
    }
    if (tid == 0) counts[img] = total;
}

// ---- K > 1024: the same greedy rule with everything that lived in LDS moved to a global workspace.  Per image:
//   prep   valid prefix n, boxes.max(), class-offset boxes + areas                           -> ws.box / ws.area / ws.n
//   mask   (row block, image): rows x words of the upper-triangular suppression bitmask        -> ws.mask [K][nwK]
//   scan   one workgroup: per 64-box block wave 0 resolves the in-block chain on the diagonal word, then the 16 waves OR the kept
//          rows into the later words of `remv` (LDS, nwK <= NMSL_MAXNW words)
#define NMSL_MAXNW 4096            /* K <= 262144 */
struct NmsLargeLayout { long mask_off, box_off, area_off, n_off, per_img; int nw; };
static NmsLargeLayout nms_large_layout(int K) {
    NmsLargeLayout l;
    l.nw = (K + 63) / 64;
    l.mask_off = 0;
    l.box_off = (long)K * l.nw * 8;
    l.area_off = l.box_off + (long)K * 16;
    l.n_off = l.area_off + (((long)K * 4 + 15) & ~15L);
    l.per_img = l.n_off + 16;
    return l;
}

__global__ __launch_bounds__(1024) void nms_large_prep_kernel(const float* __restrict__ scores, const long long* __restrict__ classes,
                                                               const float* __restrict__ boxes, int K, float score_thr, char* __restrict__ ws,
                                                               NmsLargeLayout lay) {
    __shared__ float red[16];
    __shared__ int n_valid;
    const int img = blockIdx.x, tid = threadIdx.x;
    const long base = (long)img * K;
    char* w = ws + (long)img * lay.per_img;
    float4* obox = reinterpret_cast<float4*>(w + lay.box_off);
    float* oarea = reinterpret_cast<float*>(w + lay.area_off);
    if (tid == 0) n_valid = 0;
    if (tid < 16) red[tid] = -INFINITY;
    __syncthreads();
    int cnt = 0;
    for (int i = tid; i < K; i += 1024) cnt += (scores[base + i] >= score_thr) ? 1 : 0;
    if (cnt) atomicAdd(&n_valid, cnt);
    __syncthreads();
    const int n = n_valid;
    float mx = -INFINITY;
    for (int i = tid; i < n; i += 1024) {
        const float4 bx = reinterpret_cast<const float4*>(boxes)[base + i];
        mx = fmaxf(mx, fmaxf(fmaxf(bx.x, bx.y), fmaxf(bx.z, bx.w)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    float maxc = red[0];
#pragma unroll
    for (int q = 1; q < 16; ++q) maxc = fmaxf(maxc, red[q]);
    for (int i = tid; i < n; i += 1024) {
        const float4 bx = reinterpret_cast<const float4*>(boxes)[base + i];
        const float off = (float)classes[base + i] * (maxc + 1.0f);
        const float4 ob = make_float4(bx.x + off, bx.y + off, bx.z + off, bx.w + off);
        obox[i] = ob;
        oarea[i] = (ob.z - ob.x) * (ob.w - ob.y);
    }
    if (tid == 0) *reinterpret_cast<int*>(w + lay.n_off) = n;
}

__global__ __launch_bounds__(256) void nms_large_mask_kernel(double iou_thr, char* __restrict__ ws, NmsLargeLayout lay) {
    const int img = blockIdx.y, rb = blockIdx.x, tid = threadIdx.x;
    char* w = ws + (long)img * lay.per_img;
    const int n = *reinterpret_cast<const int*>(w + lay.n_off);
    const int r0 = rb << 6;
    if (r0 >= n) return;
    const float4* box = reinterpret_cast<const float4*>(w + lay.box_off);
    const float* area = reinterpret_cast<const float*>(w + lay.area_off);
    unsigned long long* out = reinterpret_cast<unsigned long long*>(w + lay.mask_off);
    const int nw = (n + 63) >> 6;
    const int nwork = 64 * (nw - rb);
    for (int idx = tid; idx < nwork; idx += 256) {
        const int wd = rb + idx / 64, i = r0 + (idx & 63);
        if (i < n) out[(long)i * lay.nw + wd] = nms_mask_word(box, area, i, wd, n, iou_thr);
    }
}

__global__ __launch_bounds__(1024) void nms_large_scan_kernel(const float* __restrict__ scores, const long long* __restrict__ classes,
                                                               const float* __restrict__ boxes, int K, const char* __restrict__ ws,
                                                               NmsLargeLayout lay, float* out_scores, long long* out_classes,
                                                               float* out_boxes, int* keep_idx, int* counts) {
    __shared__ unsigned long long remv[NMSL_MAXNW], kept[NMSL_MAXNW];
    __shared__ int prefix[NMSL_MAXNW + 1];
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long base = (long)img * K;
    const char* w = ws + (long)img * lay.per_img;
    const int n = *reinterpret_cast<const int*>(w + lay.n_off);
    const unsigned long long* mask = reinterpret_cast<const unsigned long long*>(w + lay.mask_off);
    const int nw = (n + 63) >> 6, ld = lay.nw;
    for (int q = tid; q < nw; q += 1024) { remv[q] = 0ull; kept[q] = 0ull; }
    __syncthreads();
    for (int blk = 0; blk < nw; ++blk) {
        if (wv == 0) {
            const int row = (blk << 6) + lane;
            const unsigned long long diag = (row < n) ? mask[(long)row * ld + blk] : 0ull;
            const unsigned dlo = (unsigned)diag, dhi = (unsigned)(diag >> 32);
            const unsigned long long cur0 = remv[blk];
            unsigned clo = __builtin_amdgcn_readfirstlane((unsigned)cur0);
            unsigned chi = __builtin_amdgcn_readfirstlane((unsigned)(cur0 >> 32));
            const int nb = min(64, n - (blk << 6));
            unsigned klo = 0, khi = 0;
#pragma unroll
            for (int b = 0; b < 32; ++b) {
                if (b < nb && !((clo >> b) & 1u)) {
                    klo |= 1u << b;
                    clo |= __builtin_amdgcn_readlane(dlo, b);
                    chi |= __builtin_amdgcn_readlane(dhi, b);
                }
            }
#pragma unroll
            for (int b = 0; b < 32; ++b) {
                if (b + 32 < nb && !((chi >> b) & 1u)) {
                    khi |= 1u << b;
                    chi |= __builtin_amdgcn_readlane(dhi, b + 32);
                }
            }
            if (lane == 0) kept[blk] = ((unsigned long long)khi << 32) | klo;
        }
        __syncthreads();
        const unsigned long long k = kept[blk];
        const int row = (blk << 6) + lane;
        for (int wd = blk + 1 + wv; wd < nw; wd += 16) {      // wave wv takes the later words wd = blk + 1 + wv, + 16, ...
            unsigned long long v = (((k >> lane) & 1ull) && row < n) ? mask[(long)row * ld + wd] : 0ull;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const unsigned lo = __shfl_xor((unsigned)v, o);
                const unsigned hi = __shfl_xor((unsigned)(v >> 32), o);
                v |= ((unsigned long long)hi << 32) | lo;
            }
            if (lane == 0) remv[wd] |= v;
        }
        __syncthreads();
    }
    if (tid == 0) {
        int acc = 0;
        for (int q = 0; q < nw; ++q) { prefix[q] = acc; acc += __popcll(kept[q]); }
        prefix[nw] = acc;
    }
    __syncthreads();
    const int total = prefix[nw];
    for (int t = tid; t < n; t += 1024) {
        const unsigned long long kw = kept[t >> 6];
        if ((kw >> (t & 63)) & 1ull) {
            const int r = prefix[t >> 6] + __popcll(kw & ((1ull << (t & 63)) - 1ull));
            out_scores[base + r] = scores[base + t];
            out_classes[base + r] = classes[base + t];
            reinterpret_cast<float4*>(out_boxes)[base + r] = reinterpret_cast<const float4*>(boxes)[base + t];
            keep_idx[base + r] = t;
        }
    }
    for (int t = total + tid; t < K; t += 1024) {
        out_scores[base + t] = 0.f;
        out_classes[base + t] = 0;
        reinterpret_cast<float4*>(out_boxes)[base + t] = make_float4(0.f, 0.f, 0.f, 0.f);
        keep_idx[base + t] = -1;
    }
    if (tid == 0) counts[img] = total;
}

extern "C" int64_t fd_nms_workspace_bytes(int32_t N, int32_t K) {
    if (N < 1 || K < 1 || K > 64 * NMSL_MAXNW) return -1;
    if (K <= NMS_MAXK) return (int64_t)N * NMS_MAXK * NMS_NW * (int64_t)sizeof(unsigned long long);
    return (int64_t)N * nms_large_layout(K).per_img;
}

extern "C" int32_t fd_batched_nms(const float* scores, const int64_t* classes, const float* boxes, int32_t N,
                                  int32_t K, float score_thr, double iou_thr, float* out_scores,
                                  int64_t* out_classes, float* out_boxes, int32_t* keep_idx, int32_t* counts,
                                  void* workspace, fd_stream_t stream) {
    FD_REQUIRE(scores && classes && boxes && out_scores && out_classes && out_boxes && keep_idx && counts && workspace,
               FD_E_INVAL, "fd_batched_nms: null pointer");
    FD_REQUIRE(N >= 1 && N <= 65535 && K >= 1, FD_E_INVAL, "fd_batched_nms: N=%d K=%d", N, K);
    FD_REQUIRE(K <= 64 * NMSL_MAXNW, FD_E_UNSUPPORTED, "fd_batched_nms: K=%d > %d not supported", K, 64 * NMSL_MAXNW);
    FD_REQUIRE((((uintptr_t)boxes | (uintptr_t)out_boxes | (uintptr_t)workspace) & 15) == 0, FD_E_INVAL,
               "fd_batched_nms: boxes / workspace not 16-byte aligned");
    FD_REQUIRE(out_scores != scores && out_boxes != boxes && (const void*)out_classes != (const void*)classes, FD_E_INVAL,
               "fd_batched_nms: outputs must not alias inputs");
    if (K > NMS_MAXK) {      // FCOSHead(max_detection_box > 1024): bitmask rows of ceil(K/64) words in the global workspace
        const NmsLargeLayout lay = nms_large_layout(K);
        hipStream_t st = (hipStream_t)stream;
        hipLaunchKernelGGL(nms_large_prep_kernel, dim3(N), dim3(1024), 0, st, scores, (const long long*)classes, boxes, K, score_thr, (char*)workspace, lay);
        FD_CHECK_LAUNCH("fd_batched_nms (K > 1024: prep)");
        hipLaunchKernelGGL(nms_large_mask_kernel, dim3(lay.nw, N), dim3(256), 0, st, iou_thr, (char*)workspace, lay);
        FD_CHECK_LAUNCH("fd_batched_nms (K > 1024: mask)");
        hipLaunchKernelGGL(nms_large_scan_kernel, dim3(N), dim3(1024), 0, st, scores, (const long long*)classes, boxes, K, (const char*)workspace, lay,
                           out_scores, (long long*)out_classes, out_boxes, keep_idx, counts);
        FD_CHECK_LAUNCH("fd_batched_nms (K > 1024: scan)");
        return FD_OK;
    }
    hipLaunchKernelGGL(nms_mask_kernel, dim3((K + 63) / 64, N), dim3(256), 0, (hipStream_t)stream, scores,
                       (const long long*)classes, boxes, K, score_thr, iou_thr, (unsigned long long*)workspace);
    FD_CHECK_LAUNCH("fd_batched_nms (mask)");
    hipLaunchKernelGGL(nms_scan_kernel, dim3(N), dim3(1024), 0, (hipStream_t)stream, scores, (const long long*)classes, boxes, K,
                       score_thr, (const unsigned long long*)workspace, out_scores, (long long*)out_classes, out_boxes,
                       keep_idx, counts);
    FD_CHECK_LAUNCH("fd_batched_nms (scan)");
    return FD_OK;
}

// DataEncoder._box_nms: sort by score (desc, ties lower index), "+1" areas, keep while ovr <= thr
__global__ __launch_bounds__(1024) void box_nms_plus1_kernel(const float* __restrict__ boxes,
                                                              const float* __restrict__ scores,
                                                              const int* __restrict__ n_valid, int K, float thr,
                                                              int mode, int* keep_idx, int* counts) {
    __shared__ NmsShared sh;
    __shared__ unsigned long long order[1024];
    const int img = blockIdx.x, tid = threadIdx.x;
    const long base = (long)img * K;
    int n = n_valid ? n_valid[img] : K;
    n = max(0, min(n, K));
    order[tid] = (tid < n) ? (((unsigned long long)fd_order_key(scores[base + tid]) << 32) |
                              (unsigned long long)(0xFFFFFFFFu - (unsigned)tid))
                           : 0ull;
    __syncthreads();
    bitonic_desc_1024(order, tid);
    int src = -1;
    if (tid < n) {
        src = (int)(0xFFFFFFFFu - (unsigned)(order[tid] & 0xFFFFFFFFull));
        const float4 b = reinterpret_cast<const float4*>(boxes)[base + src];
        sh.box[tid] = b;
        sh.area[tid] = ((b.z - b.x) + 1.0f) * ((b.w - b.y) + 1.0f);
    }
    __syncthreads();
    if (n > 0) nms_core<true>(sh, n, 0.0, thr, mode, tid);
    int total = 0;
    if (n > 0) {
        const int nw = (n + 63) >> 6;
        for (int q = 0; q < nw; ++q) total += __popcll(sh.kept[q]);
    }
    if (tid < n && ((sh.kept[tid >> 6] >> (tid & 63)) & 1ull)) keep_idx[base + nms_rank(sh, tid)] = src;
    if (tid < K && tid >= total) keep_idx[base + tid] = -1;
    if (tid == 0) counts[img] = total;
}

extern "C" int32_t fd_box_nms_plus1(const float* boxes, const float* scores, const int32_t* n_valid, int32_t N,
                                    int32_t K, float thr, int32_t mode, int32_t* keep_idx, int32_t* counts,
                                    fd_stream_t stream) {
    FD_REQUIRE(boxes && scores && keep_idx && counts, FD_E_INVAL, "fd_box_nms_plus1: null pointer");
    FD_REQUIRE(N >= 1 && K >= 1, FD_E_INVAL, "fd_box_nms_plus1: N=%d K=%d", N, K);
    FD_REQUIRE(K <= NMS_MAXK, FD_E_UNSUPPORTED, "fd_box_nms_plus1: K=%d > %d not supported", K, NMS_MAXK);
    FD_REQUIRE(mode == 0 || mode == 1, FD_E_INVAL, "fd_box_nms_plus1: mode must be 0 ('union') or 1 ('min')");
    FD_REQUIRE(((uintptr_t)boxes & 15) == 0, FD_E_INVAL, "fd_box_nms_plus1: boxes not 16-byte aligned");
    hipLaunchKernelGGL(box_nms_plus1_kernel, dim3(N), dim3(1024), 0, (hipStream_t)stream, boxes, scores, n_valid,
                       K, thr, mode, keep_idx, counts);
    FD_CHECK_LAUNCH("fd_box_nms_plus1");
    return FD_OK;
}

// --------------------------------------------------------------------------------------------
// pairwise IoU, clip
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pairwise_iou_kernel(const float4* __restrict__ a, const float4* __restrict__ b,
                                                            int Na, int Nb, int plus_one, float* out) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= Nb) return;
    const float p = plus_one ? 1.0f : 0.0f;
    const float4 x = a[i], y = b[j];
    const float ltx = fmaxf(x.x, y.x), lty = fmaxf(x.y, y.y);
    const float rbx = fminf(x.z, y.z), rby = fminf(x.w, y.w);
    const float w = fmaxf((rbx - ltx) + p, 0.0f), h = fmaxf((rby - lty) + p, 0.0f);
    const float inter = w * h;
    const float a1 = ((x.z - x.x) + p) * ((x.w - x.y) + p);
    const float a2 = ((y.z - y.x) + p) * ((y.w - y.y) + p);
    out[(long)i * Nb + j] = inter / ((a1 + a2) - inter);
}

extern "C" int32_t fd_pairwise_iou(const float* a, const float* b, int32_t Na, int32_t Nb, int32_t plus_one,
                                   float* out, fd_stream_t stream) {
    FD_REQUIRE(a && b && out, FD_E_INVAL, "fd_pairwise_iou: null pointer");
    FD_REQUIRE(Na >= 1 && Nb >= 1 && Na <= 65535, FD_E_INVAL, "fd_pairwise_iou: Na=%d Nb=%d", Na, Nb);
    FD_REQUIRE((((uintptr_t)a | (uintptr_t)b) & 15) == 0, FD_E_INVAL, "fd_pairwise_iou: boxes not 16-byte aligned");
    hipLaunchKernelGGL(pairwise_iou_kernel, dim3((Nb + 255) / 256, Na), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)a, (const float4*)b, Na, Nb, plus_one, out);
    FD_CHECK_LAUNCH("fd_pairwise_iou");
    return FD_OK;
}

__global__ __launch_bounds__(256) void clip_kernel(float4* boxes, long n, float xmax, float ymax) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float4 b = boxes[i];
    b.x = fminf(fmaxf(b.x, 0.f), xmax);
    b.y = fminf(fmaxf(b.y, 0.f), ymax);
    b.z = fminf(fmaxf(b.z, 0.f), xmax);
    b.w = fminf(fmaxf(b.w, 0.f), ymax);
    boxes[i] = b;
}

extern "C" int32_t fd_clip_boxes(float* boxes, int64_t n_boxes, int32_t img_h, int32_t img_w, fd_stream_t stream) {
    FD_REQUIRE(boxes, FD_E_INVAL, "fd_clip_boxes: null pointer");
    FD_REQUIRE(((uintptr_t)boxes & 15) == 0, FD_E_INVAL, "fd_clip_boxes: boxes not 16-byte aligned");
    if (n_boxes <= 0) return FD_OK;
    hipLaunchKernelGGL(clip_kernel, dim3((unsigned)((n_boxes + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (float4*)boxes, (long)n_boxes, (float)(img_w - 1), (float)(img_h - 1));
    FD_CHECK_LAUNCH("fd_clip_boxes");
    return FD_OK;
}

// --------------------------------------------------------------------------------------------
// detection records for the multi-GPU all-gather: ONE fixed-size fp32 message per rank
//   rec[b][0]     = (count_b, 0, 0, 0, 0, 0)
//   rec[b][1 + r] = (x1, y1, x2, y2, score, class)      r < K   (class ids and counts are < 2^24: exact in fp32)
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_det_kernel(const float* __restrict__ scores, const long long* __restrict__ classes,
                                                        const float4* __restrict__ boxes, const int* __restrict__ counts, int K,
                                                        float* __restrict__ rec, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long b = i / (K + 1);
    const int r = (int)(i - b * (K + 1));
    float* o = rec + i * 6;
    if (r == 0) {
        o[0] = (float)counts[b]; o[1] = o[2] = o[3] = o[4] = o[5] = 0.f;
    } else {
        const long src = b * K + (r - 1);
        const float4 bx = boxes[src];
        o[0] = bx.x; o[1] = bx.y; o[2] = bx.z; o[3] = bx.w; o[4] = scores[src]; o[5] = (float)classes[src];
    }
}

__global__ __launch_bounds__(256) void unpack_det_kernel(const float* __restrict__ rec, int K, float* __restrict__ scores,
                                                          long long* __restrict__ classes, float4* __restrict__ boxes,
                                                          int* __restrict__ counts, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long b = i / (K + 1);
    const int r = (int)(i - b * (K + 1));
    const float* o = rec + i * 6;
    if (r == 0) {
        counts[b] = (int)o[0];
    } else {
        const long dst = b * K + (r - 1);
        boxes[dst] = make_float4(o[0], o[1], o[2], o[3]);
        scores[dst] = o[4];
        classes[dst] = (long long)o[5];
    }
}

extern "C" int32_t fd_pack_detections(const float* scores, const int64_t* classes, const float* boxes, const int32_t* counts,
                                      int32_t B, int32_t K, float* records, fd_stream_t stream) {
    FD_REQUIRE(scores && classes && boxes && counts && records && B >= 1 && K >= 1, FD_E_INVAL, "fd_pack_detections: bad argument");
    FD_REQUIRE(((uintptr_t)boxes & 15) == 0, FD_E_INVAL, "fd_pack_detections: boxes not 16-byte aligned");
    const long total = (long)B * (K + 1);
    hipLaunchKernelGGL(pack_det_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, scores,
                       (const long long*)classes, (const float4*)boxes, counts, K, records, total);
    FD_CHECK_LAUNCH("fd_pack_detections");
    return FD_OK;
}

extern "C" int32_t fd_unpack_detections(const float* records, int32_t B, int32_t K, float* scores, int64_t* classes, float* boxes,
                                        int32_t* counts, fd_stream_t stream) {
    FD_REQUIRE(scores && classes && boxes && counts && records && B >= 1 && K >= 1, FD_E_INVAL, "fd_unpack_detections: bad argument");
    FD_REQUIRE(((uintptr_t)boxes & 15) == 0, FD_E_INVAL, "fd_unpack_detections: boxes not 16-byte aligned");
    const long total = (long)B * (K + 1);
    hipLaunchKernelGGL(unpack_det_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, records, K, scores,
                       (long long*)classes, (float4*)boxes, counts, total);
    FD_CHECK_LAUNCH("fd_unpack_detections");
    return FD_OK;
}
