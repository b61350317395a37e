// fd_conv_f16.hip -- dense convolution with f16 operands and fp32 accumulation on K-tiles of 64 channels (FD_TILE_F16K64): the AMP training step's conv kernel
// (train.py:33,175-181: the reference trains under torch.autocast(float16) + GradScaler; forward and data gradient of every dense conv).
//
// The single-plane f16 instantiations of the fp32 kernel (fd_conv.hip, H1) keep that kernel's K-tile of 32 channels: per tile and wave 2 K-steps of
// v_mfma_f32_32x32x16_f16 -- a quarter of the matrix time the fp32 MFMAs give the same loader, barrier and LDS traffic to hide behind: 0.16 of the f16 peak, feed-bound.
// Here a K-tile is 64 channels = 128 bytes per row in f16, i.e. exactly the row the fp32 kernel's loader and LDS layout are built for (8 lanes x 16 bytes, chunks
// XOR-swizzled by the row): twice the matrix work per tile, and with activation maps STORED as f16 (fd_conv_params.io_f16) one 16-byte fetch per lane carries eight
// channels straight to LDS.  fp32 maps are accepted too (two fetches per lane, one rounding on the way to LDS): every AMP conv can take this kernel.
//   * weights: fd_pack_conv_weight_f32 mode | 16 -> f16 [N][K / 64][taps][64] (K-tile order (64-channel chunk, tap), as the fp32 packing);
//   * 1x1 and k x k, any stride / dilation / padding, pyramids, explicit output size + scatter (the parity classes of a strided data gradient);
//   * epilogue: scale / shift, residual add or ReLU mask (f16 or fp32 map), ReLU / SiLU / none, f16 or fp32 output; 4-channel aligned views.  No split-K, gate, gn_stats.
#include "fd_conv_common.h"
#include <type_traits>
#ifndef FD_F16_DBG
#define FD_F16_DBG 0     /* timing builds only (tools/_ab): 1 no MFMAs, 16 no residual fetch, 32 no output stores, 2 fetch only the first two K-tiles, 4 park nothing in the loop, 8 no fragment reads */
#endif

// SIMPLE: the epilogue of most AMP layers -- f16 output in 16-byte accesses, no activation or ReLU on every channel, no output scatter, residual / mask (if any) an f16 map
// in 16-byte accesses -- as its own instantiation: the general epilogue's code (per-channel activation start, SiLU, fp32 maps, 8-byte accesses, scatter) costs
// instruction fetch and scalar branch work (2.2 x the fetch requests of the lean form; the cache does not miss) (64 > 256 + residual: 128 us with the whole activation switch inlined, 88 us without, on the same bytes).
template <int WGM, int WGN, int TM, int TN, bool SIMPLE, bool X16>
__global__ __launch_bounds__(WGM * WGN * 64, 2) void conv_f16k64_kernel(ConvArgs a) {
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    constexpr int NT = WGM * WGN * 64;
    constexpr int RPP = NT / 8;                 // rows per loader pass (8 lanes fetch one 128-byte row of the tile)
    constexpr int AP = BM / RPP, BP = BN / RPP;
    constexpr int STG = (BM + BN) * 32;         // floats (= 64 halves per row) per buffer
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Ls = reinterpret_cast<float*>(smem); // [2][BM + BN rows][32 floats]: A rows first, then B rows; 16-byte chunks swizzled by lds_off()

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int l31 = lane & 31, lh = lane >> 5;

    // XCD-aware tile order (as conv_igemm_kernel)
    const int nblk = a.mtiles * a.ntiles;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int mt = bid / a.ntiles, nt = bid - mt * a.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;

    const int lrow = tid >> 3, chunk = tid & 7;          // this lane: row lrow (+ RPP per pass), channels 8 chunk .. 8 chunk + 7 of the K-tile
    constexpr unsigned OOB = 0xC0000000u;
    constexpr int esh = X16 ? 1 : 2;                     // bytes per input element, as a shift (X16: the input map is stored as f16)
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, (short)0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, (short)0, (int)a.w_bytes, 0x00020000);
    unsigned a_off[AP];
    int a_wcs[AP], a_hi0[AP], a_wi0[AP], a_H[AP], a_W[AP];
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int m = m0 + lrow + RPP * i;
        if (a.is_gemm) {     // 1x1, stride 1, unpadded: the input row is the output row
            a_off[i] = ((unsigned)m * (unsigned)a.x_cs + (unsigned)(a.x_co + chunk * 8)) << esh;
            a_wcs[i] = 0; a_hi0[i] = 0; a_wi0[i] = 0; a_H[i] = (m < a.M) ? 1 : 0; a_W[i] = 1;
            continue;
        }
        int s = 0;
#pragma unroll
        for (int t = 1; t < FD_MAX_SEG; ++t)
            if (t < a.nseg && m >= a.m_out[t]) s = t;
        const int Ho = a.Ho[s], Wo = a.Wo[s], H = a.H[s], W = a.W[s];
        const int local = m - a.m_out[s];
        const int hw = Ho * Wo;
        const int n = local / hw;
        const int rem = local - n * hw;
        const int ho = rem / Wo, wo = rem - ho * Wo;
        a_hi0[i] = ho * a.stride - a.pad;
        a_wi0[i] = wo * a.stride - a.pad;
        a_H[i] = (m < a.M) ? H : 0;
        a_W[i] = W;
        a_wcs[i] = (W * a.x_cs) << esh;
        a_off[i] = ((unsigned)(a.m_in[s] + n * H * W + a_hi0[i] * W + a_wi0[i]) * (unsigned)a.x_cs + (unsigned)(a.x_co + chunk * 8)) << esh;
    }
    unsigned b_off[BP];
#pragma unroll
    for (int j = 0; j < BP; ++j) {
        const int n = n0 + lrow + RPP * j;
        b_off[j] = (n < a.Cout) ? ((unsigned)n * (unsigned)a.KT * 64u + (unsigned)(chunk * 8)) * 2u : OOB;
    }

    // TWO register sets: the fetches of K-tile kt + 2 are issued while K-tile kt is multiplied (a K-tile's MFMAs are 0.25 us of work, a fetch takes 1 - 2 us: one tile
    // ahead, every K-tile cost a whole memory latency -- 1.6 us per tile and workgroup on the 3x3 layers)
    float4 ra[2][AP], ra2[2][X16 ? 1 : AP], rb[2][BP];
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;
    int ld_cc = 0, ld_r = 0, ld_q = 0;      // K-tile = (64-channel chunk, filter row, filter column), advancing as counters
    struct LdCtx { int dr, dq; unsigned dbytes, kb; bool c_ok; };
    auto load_begin = [&](int kt, bool live) {           // the K-tile the counters point at; live = false: every fetch of it is out of range (no memory traffic)
        LdCtx c;
        c.dr = ld_r * a.dil; c.dq = ld_q * a.dil;
        c.dbytes = (unsigned)(c.dq * a.x_cs + ld_cc * 64) << esh;
        c.c_ok = live & (ld_cc * 64 + chunk * 8 < a.Cin);
        c.kb = live ? (unsigned)kt * 128u : OOB;
        if ((FD_F16_DBG & 2) && kt >= 2) { c.c_ok = false; c.kb = OOB; }
        if (++ld_q == a.KW) { ld_q = 0; if (++ld_r * a.KW == a.ntaps) { ld_r = 0; ++ld_cc; } }
        return c;
    };
    auto load_one = [&](const LdCtx& c, auto set, int idx) {      // row idx of the AP + BP (A rows first) of that K-tile into register set S
        constexpr int S = decltype(set)::value;
        if (idx < AP) {
            const int hi = a_hi0[idx] + c.dr, wi = a_wi0[idx] + c.dq;
            const bool ok = ((unsigned)hi < (unsigned)a_H[idx]) & ((unsigned)wi < (unsigned)a_W[idx]) & c.c_ok;      // (&, not &&: no branches between the MFMAs)
            const unsigned off = ok ? a_off[idx] + (unsigned)__mul24(c.dr, a_wcs[idx]) + c.dbytes : OOB;
            ra[S][idx] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)off, 0, 0));
            if constexpr (!X16) ra2[S][idx] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)off, 16, 0));      // fp32 map: channels 4 .. 7 of the lane's eight
        } else {
            const unsigned bo = b_off[idx - AP];
            rb[S][idx - AP] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (int)(((bo == OOB) | (c.kb == OOB)) ? OOB : bo + c.kb), 0, 0));
        }
    };
    auto load_tile = [&](int kt, auto set) {
        const LdCtx c = load_begin(kt, true);
#pragma unroll
        for (int idx = 0; idx < AP + BP; ++idx) load_one(c, set, idx);
    };
    auto store_tile = [&](int buf, auto set) {
        constexpr int S = decltype(set)::value;
        float* Ab = Ls + buf * STG;
        float* Bb = Ab + BM * 32;
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            float4 v = ra[S][i];
            if constexpr (!X16) {       // round the eight fp32 values once, to nearest even
                const f32x4 lo = {ra[S][i].x, ra[S][i].y, ra[S][i].z, ra[S][i].w}, hi = {ra2[S][i].x, ra2[S][i].y, ra2[S][i].z, ra2[S][i].w};
                const h4 l4 = __builtin_convertvector(lo, h4), h4_ = __builtin_convertvector(hi, h4);
                const h8 p = {l4[0], l4[1], l4[2], l4[3], h4_[0], h4_[1], h4_[2], h4_[3]};
                v = __builtin_bit_cast(float4, p);
            }
            *reinterpret_cast<float4*>(Ab + lds_off(lrow + RPP * i, chunk)) = v;
        }
#pragma unroll
        for (int j = 0; j < BP; ++j) *reinterpret_cast<float4*>(Bb + lds_off(lrow + RPP * j, chunk)) = rb[S][j];
    };

    f32x16 acc[TM][TN];

    // one parked row: element idx of the AP + BP (A rows first) of register set S into buffer `buf`
    auto store_one = [&](int buf, auto set, int idx) {
        constexpr int S = decltype(set)::value;
        if (FD_F16_DBG & 4) return;
        float* Ab = Ls + buf * STG;
        float* Bb = Ab + BM * 32;
        if (idx < AP) {
            float4 v = ra[S][idx];
            if constexpr (!X16) {
                const f32x4 lo = {ra[S][idx].x, ra[S][idx].y, ra[S][idx].z, ra[S][idx].w}, hi = {ra2[S][idx].x, ra2[S][idx].y, ra2[S][idx].z, ra2[S][idx].w};
                const h4 l4 = __builtin_convertvector(lo, h4), h4_ = __builtin_convertvector(hi, h4);
                const h8 p = {l4[0], l4[1], l4[2], l4[3], h4_[0], h4_[1], h4_[2], h4_[3]};
                v = __builtin_bit_cast(float4, p);
            }
            *reinterpret_cast<float4*>(Ab + lds_off(lrow + RPP * idx, chunk)) = v;
        } else {
            *reinterpret_cast<float4*>(Bb + lds_off(lrow + RPP * (idx - AP), chunk)) = rb[S][idx - AP];
        }
    };
    // multiply buffer `buf`; PARK: meanwhile park register set S (K-tile kt + 1) in the other buffer, a quarter of its rows behind each K-step's MFMAs -- parked after
    // the last MFMA, the 13-cycle ds_write_b128s and the fragment reads of the next step ran with the matrix pipe idle (MFMA-only 83 us, + reads 112, + fetch / park 159).
    // Fragments of K-step ks + 1 are read before the MFMAs of ks.
    auto mfma_tile = [&](int buf, auto set, auto park, auto lset, int kt_load, bool live) {
        constexpr bool PARK = decltype(park)::value;      // (PARK steps also FETCH K-tile kt_load into register set lset, a row behind every other MFMA)
        if (FD_F16_DBG & 1) return;
        LdCtx lc{};
        if constexpr (PARK) lc = load_begin(kt_load, live);
        constexpr int NS = 4 * TM * TN, NI = 2 * (AP + BP);        // MFMA slots; rows to park + rows to fetch, taken alternately
        const float* Ab = Ls + buf * STG + (wm * TM * 32) * 32;
        const float* Bb = Ls + buf * STG + BM * 32 + (wn * TN * 32) * 32;
        h8 fa[2][TM], fb[2][TN];
        auto frags = [&](int ks, int q) {       // four K-steps of 16: lane half lh carries k = 16 ks + 8 lh .. + 7 (one 16-byte chunk)
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[q][i] = (FD_F16_DBG & 8) ? h8{(_Float16)l31, 1, 1, 1, 1, 1, 1, 1} : *reinterpret_cast<const h8*>(Ab + lds_off(i * 32 + l31, 2 * ks + lh));
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[q][j] = (FD_F16_DBG & 8) ? h8{(_Float16)lh, 1, 1, 1, 1, 1, 1, 1} : *reinterpret_cast<const h8*>(Bb + lds_off(j * 32 + l31, 2 * ks + lh));
        };
        frags(0, 0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks < 3) frags(ks + 1, (ks + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);        // (the scheduler sinks these reads below the MFMAs otherwise: the matrix pipe then waits for them every K-step)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ks & 1][i], fb[ks & 1][j], acc[i][j], 0, 0, 0);
                    if constexpr (PARK) {       // this slot's share of the rows: item t even = park row t / 2, odd = fetch row t / 2
                        const int slot = ks * TM * TN + i * TN + j;
#pragma unroll
                        for (int t = slot * NI / NS; t < (slot + 1) * NI / NS; ++t) {
                            if (t & 1) load_one(lc, lset, t >> 1);
                            else store_one(buf ^ 1, set, t >> 1);
                        }
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    typedef std::integral_constant<bool, true> PK;
    typedef std::integral_constant<bool, false> NPK;

#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    load_tile(0, S0{});
    if (a.KT > 1) load_tile(1, S1{});
    store_tile(0, S0{});
    __syncthreads();
    // step kt (parity P): multiply buffer P; behind its MFMAs fetch K-tile kt + 2 into register set P (emptied into LDS by step kt - 1) and park K-tile kt + 1 (set 1 - P)
    // in buffer 1 - P.  (The fetches' address arithmetic ahead of the first MFMA left the matrix pipe idle for ~400 cycles per K-tile.)
    int kt = 0;
    for (; kt + 2 <= a.KT - 1; kt += 2) {          // (two steps per trip, nothing conditional inside: one basic block, so the waits on the fetches stay counted)
        mfma_tile(0, S1{}, PK{}, S0{}, kt + 2, kt + 2 < a.KT);
        __syncthreads();
        mfma_tile(1, S0{}, PK{}, S1{}, kt + 3, kt + 3 < a.KT);
        __syncthreads();
    }
    if (kt < a.KT - 1) {
        mfma_tile(0, S1{}, PK{}, S0{}, kt + 2, false);
        __syncthreads();
    }

    // ---- epilogue: the wave's 32-pixel x (TN * 32)-channel strips through a per-wave LDS stage; a lane then owns EIGHT consecutive channels of one pixel (16 bytes
    // of an f16 map: 4 TN lanes cover a row's 64 TN bytes per instruction).  The f16 residual / mask values of the whole wave tile are fetched BEFORE the last
    // K-tile's MFMAs: a K = 64 layer is nothing but epilogue, and fetched inside the store loop they were 16 dependent HBM round trips per wave (1.8 TB/s).
    constexpr int SW = TN * 32 + 4;                          // stage row pitch in floats
    constexpr int LPR = TN * 4, RPS = 64 / LPR, NP = 32 / RPS;   // lanes per row, rows per pass, passes per 32-row strip
    float* stage = reinterpret_cast<float*>(smem) + wave * (32 * SW);
    const _Float16* res16 = reinterpret_cast<const _Float16*>(a.res);
    _Float16* y16 = reinterpret_cast<_Float16*>(a.y);
    const int c8 = (lane % LPR) * 8, prow = lane / LPR;
    const int nn = n0 + wn * TN * 32 + c8;
    const bool ok0 = nn < a.Cout, ok1 = nn + 4 < a.Cout;     // (Cout % 4 == 0: the lane's channels are valid in fours)
    const bool pre = a.res && a.res16;                       // (uniform)
    const bool relu_all = a.act == FD_ACT_RELU && a.act_c0 <= n0;      // (uniform) ReLU on every channel of this tile
    h8 rr[TM][NP];
    if (pre) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int m = m0 + (wm * TM + i) * 32 + prow + RPS * p;
                h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (m < a.M && ok0 && !(FD_F16_DBG & 16)) {
                    const _Float16* q = res16 + (size_t)(SIMPLE ? m : out_row(a, m)) * a.res_cs + a.res_co + nn;
                    if (SIMPLE || a.wide8) v = *reinterpret_cast<const h8*>(q);
                    else {
                        const h4 lo = *reinterpret_cast<const h4*>(q);
                        h4 hi = {0, 0, 0, 0};
                        if (ok1) hi = *reinterpret_cast<const h4*>(q + 4);
                        v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    }
                }
                rr[i][p] = v;
            }
    }
    mfma_tile((a.KT - 1) & 1, S0{}, NPK{}, S0{}, 0, false);
    __syncthreads();

    float sc[TN], sf[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + (wn * TN + j) * 32 + l31;
        const bool n_ok = n < a.Cout;
        sc[j] = (a.scale && n_ok) ? a.scale[n] : 1.0f;
        sf[j] = (a.shift && n_ok) ? a.shift[n] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int mb = m0 + (wm * TM + i) * 32;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) stage[((e & 3) + 8 * (e >> 2) + 4 * lh) * SW + j * 32 + l31] = acc[i][j][e] * sc[j] + sf[j];
        wave_lds_sync();
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int row = prow + RPS * p;
            const int m = mb + row;
            if (m < a.M && ok0) {
                const float4 v0 = *reinterpret_cast<const float4*>(stage + row * SW + c8), v1 = *reinterpret_cast<const float4*>(stage + row * SW + c8 + 4);
                float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                const size_t mo = (size_t)(SIMPLE ? m : out_row(a, m));
                if (a.res) {
                    float r[8];
                    if (SIMPLE || pre) {
#pragma unroll
                        for (int c = 0; c < 8; ++c) r[c] = (float)rr[i][p][c];
                    } else {         // fp32 residual map
                        const float* q = a.res + mo * a.res_cs + a.res_co + nn;
                        const float4 r0 = *reinterpret_cast<const float4*>(q);
                        float4 r1 = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (ok1) r1 = *reinterpret_cast<const float4*>(q + 4);
                        r[0] = r0.x; r[1] = r0.y; r[2] = r0.z; r[3] = r0.w; r[4] = r1.x; r[5] = r1.y; r[6] = r1.z; r[7] = r1.w;
                    }
#pragma unroll
                    for (int c = 0; c < 8; ++c) v[c] = a.res_mask ? (r[c] > 0.f ? v[c] : 0.f) : v[c] + r[c];
                }
                if (relu_all) {          // (uniform) the common case as eight v_max: no per-channel compare, no activation switch
#pragma unroll
                    for (int c = 0; c < 8; ++c) v[c] = fd_act(v[c], FD_ACT_RELU, 0.f);
                } else if (SIMPLE) {
                } else if (a.act == FD_ACT_SILU) {      // (the launcher admits ReLU, SiLU, none; selects, not branches: with fd_act's whole switch behind a branch per
#pragma unroll                                  // channel the epilogue was 11 000 instructions of compares and branches: DESIGN 4.1n)
                    for (int c = 0; c < 8; ++c) {
                        const float t = fd_act(v[c], FD_ACT_SILU, 0.f);
                        v[c] = nn + c >= a.act_c0 ? t : v[c];
                    }
                } else if (a.act == FD_ACT_RELU) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const float t = fd_act(v[c], FD_ACT_RELU, 0.f);
                        v[c] = nn + c >= a.act_c0 ? t : v[c];
                    }
                }
                const size_t yo = mo * a.y_cs + a.y_co + nn;
                if ((FD_F16_DBG & 32) && v[0] != 123.25f) continue;
                if (SIMPLE || a.y16) {
                    const f32x4 lo = {v[0], v[1], v[2], v[3]}, hi = {v[4], v[5], v[6], v[7]};
                    const h4 l4 = __builtin_convertvector(lo, h4), u4 = __builtin_convertvector(hi, h4);
                    if (SIMPLE || a.wide8) *reinterpret_cast<h8*>(y16 + yo) = __builtin_shufflevector(l4, u4, 0, 1, 2, 3, 4, 5, 6, 7);
                    else {
                        *reinterpret_cast<h4*>(y16 + yo) = l4;
                        if (ok1) *reinterpret_cast<h4*>(y16 + yo + 4) = u4;
                    }
                } else {
                    *reinterpret_cast<float4*>(a.y + yo) = make_float4(v[0], v[1], v[2], v[3]);
                    if (ok1) *reinterpret_cast<float4*>(a.y + yo + 4) = make_float4(v[4], v[5], v[6], v[7]);
                }
            }
        }
        wave_lds_sync();
    }
}

template <int WGM, int WGN, int TM, int TN, bool SIMPLE, bool X16>
static int launch_f16k64_e(const ConvArgs& a, hipStream_t stream) {
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32, NT = WGM * WGN * 64;
    constexpr int lds = 2 * (BM + BN) * 128;                  // >= the epilogue's 32 x (32 TN + 4) floats per wave
    static_assert(lds >= NT / 64 * 32 * (TN * 32 + 4) * 4, "epilogue stage");
    ConvArgs b = a;
    b.mtiles = (a.M + BM - 1) / BM;
    b.ntiles = (a.Cout + BN - 1) / BN;
    auto kern = conv_f16k64_kernel<WGM, WGN, TM, TN, SIMPLE, X16>;
    static std::atomic<unsigned> attr_mask{0};
    fd_set_max_lds_once(attr_mask, reinterpret_cast<const void*>(kern), lds);
    hipLaunchKernelGGL(kern, dim3(b.mtiles * b.ntiles), dim3(NT), lds, stream, b);
    FD_CHECK_LAUNCH("fd_conv2d_nhwc_f32 (FD_TILE_F16K64)");
    return FD_OK;
}

template <int WGM, int WGN, int TM, int TN>
static int launch_f16k64(const ConvArgs& a, hipStream_t stream) {
    const bool simple = (a.act == FD_ACT_NONE || (a.act == FD_ACT_RELU && a.act_c0 <= 0)) && !a.sc_on && a.y16 && a.wide8 && (!a.res || a.res16);
    if (a.x16) return simple ? launch_f16k64_e<WGM, WGN, TM, TN, true, true>(a, stream) : launch_f16k64_e<WGM, WGN, TM, TN, false, true>(a, stream);
    return simple ? launch_f16k64_e<WGM, WGN, TM, TN, true, false>(a, stream) : launch_f16k64_e<WGM, WGN, TM, TN, false, false>(a, stream);
}

// `a`: the argument block fd_conv2d_nhwc_f32 has filled (geometry, views, epilogue, io_f16 flags); p->w = the mode | 16 packing
int fd_launch_conv_f16k64(const fd_conv_params* p, ConvArgs& a, hipStream_t stream) {
    FD_REQUIRE(p->precision == FD_PREC_F16 && p->mode != FD_CONV_STEM && p->Cin % 64 == 0 && a.vec_epi && p->ksplit <= 1 && !p->gate && !p->gn_stats && !p->x2 &&
                   (p->act == FD_ACT_NONE || p->act == FD_ACT_RELU || p->act == FD_ACT_SILU),
               FD_E_UNSUPPORTED, "fd_conv2d: FD_TILE_F16K64 needs FD_PREC_F16, Cin %% 64 == 0, 4-channel aligned output / residual views, ReLU / SiLU / no "
                                 "activation, no split-K / gate / gn_stats / x2");
    FD_REQUIRE(p->x_cs % 8 == 0 && p->x_co % 8 == 0, FD_E_UNSUPPORTED, "fd_conv2d: FD_TILE_F16K64 fetches eight channels per lane: x_cs and x_co must be multiples of 8");
    a.ntaps = p->KH * p->KW;
    a.KT = a.ntaps * (p->Cin / 64);
    const long wb = (long)p->Cout * a.KT * 128;
    FD_REQUIRE(wb < 0xC0000000L, FD_E_UNSUPPORTED, "fd_conv2d: weight buffer exceeds 3 GiB");
    a.w_bytes = (unsigned)wb;
    // 16-byte f16 accesses in the epilogue (eight channels per lane): the f16 views must be 8-channel aligned
    a.wide8 = p->Cout % 8 == 0 && (!a.y16 || (p->y_cs % 8 == 0 && p->y_co % 8 == 0)) && (!(a.res && a.res16) || (p->res_cs % 8 == 0 && p->res_co % 8 == 0 && ((uintptr_t)p->res & 15) == 0));
    auto blocks = [&](int bm, int bn) { return (long)((a.M + bm - 1) / bm) * ((a.Cout + bn - 1) / bn); };
    // the largest tile that still gives >= 2 workgroups per CU; narrow layers take the 64-wide tiles
    if (a.Cout > 64 && blocks(128, 128) >= 512) return launch_f16k64<2, 2, 2, 2>(a, stream);
    if (blocks(128, 64) >= 512) return launch_f16k64<2, 2, 2, 1>(a, stream);
    return launch_f16k64<2, 2, 1, 1>(a, stream);
}
