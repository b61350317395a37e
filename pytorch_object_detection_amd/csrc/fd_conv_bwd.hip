// fd_conv_bwd.hip — convolution weight gradient on v_mfma_f32_32x32x2_f32 (exact fp32), gfx950.
//
//   dW[co][r][q][ci] = sum_m dY[m][co] * X[pix(m, r, q)][ci]            (m = output pixel of the forward conv)
//
// i.e. a GEMM whose reduction runs over PIXELS: C[co][ci] (per filter tap) = dY^T * X_shifted.  Both operands are
// NHWC rows, so a K-tile is simply 32 consecutive pixel rows of dY (128 output channels wide) and the 32 shifted
// pixel rows of X (BN input channels wide); they are staged as [pixel][channel] in LDS and the MFMA operands are
// single floats: lane (i = l&31, k = l>>5) reads LDS[pixel 2p + k][channel i] — 32 consecutive floats per half
// wave, conflict-free.  The pixel range is split over `nsplit` workgroups per (co, tap, ci) tile; each writes its
// partial tile to a slab and a second launch adds the slabs in order (deterministic).  The data gradient needs no
// kernel of its own: for stride-1 layers it is the forward kernel run on dY with the flipped / transposed weights.
// Replaces the weight-gradient half of torch's convolution_backward for the reference's train step (train.py:175-181).
#include <algorithm>

#include "fd_conv_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define FD_WGRAD_MAX_RANGES 128

struct WgradArgs {
    const float* x; const float* dy; float* out;
    int x_cs, x_co, dy_cs, dy_co;
    int Cin, Cout, KW, stride, pad, dil, ntaps;
    int M, Ktot;
    int nseg;
    int H[FD_MAX_SEG], W[FD_MAX_SEG], Ho[FD_MAX_SEG], Wo[FD_MAX_SEG];
    int m_in[FD_MAX_SEG], m_out[FD_MAX_SEG + 1];
    unsigned mg_hw[FD_MAX_SEG], mg_w[FD_MAX_SEG];   // division by Ho*Wo / Wo as mulhi + shift (0 = divisor 1)
    int sh_hw[FD_MAX_SEG], sh_w[FD_MAX_SEG];
    int co_tiles, ci_tiles;      // tiles of 128 output channels, tiles of BN input channels (per tap)
    int rows_per_split;          // 1x1 layers: uniform pixel ranges of this many rows (multiple of 32)
    int r_begin[FD_WGRAD_MAX_RANGES], r_end[FD_WGRAD_MAX_RANGES];   // other layers: pixel range of workgroup row y ...
    unsigned char r_seg[FD_WGRAD_MAX_RANGES];                         // ... and the one pyramid level it lies in
    long slab;                   // elements per split slab (= Cout * Ktot)
    unsigned x_bytes, dy_bytes;
    int is_gemm;
    int x16, dy16;               // conv_wgrad_f16_kernel: the operand maps hold f16 elements (AMP activations / gradients stored as f16: fd_conv_wgrad_params.io_f16)
};

// floor(n / d) for 0 <= n < 2^31 with (m, sh) from magic_div(): m = ceil(2^(31+l) / d), l = ceil(log2 d), sh = l - 1
__device__ __forceinline__ int fast_div(int n, unsigned m, int sh) {
    return m ? (int)(__umulhi((unsigned)n, m) >> sh) : n;
}

// BM = 128: 2 x 2 waves, wave tile 64 co x BN/2 ci.  BM = 32 (narrow predictors, Cout <= 32): 1 x 4 waves, wave tile
// 32 co x BN/4 ci (BN = 128 only) -- a 128-wide tile would spend 3/4 of its MFMAs on padding.
template <int BN, int BM>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
    constexpr int WAVES_N = BM == 128 ? 2 : 4;
    constexpr int TM = BM == 128 ? 2 : 1;         // 32-wide co sub-tiles per wave
    constexpr int TN = BN / (32 * WAVES_N);       // 32-wide ci sub-tiles per wave
    static_assert(TN >= 1, "BM = 32 needs BN = 128");
    constexpr int ATPR = BM / 4;                  // threads per A row (float4 each)
    constexpr int ARPP = 256 / ATPR;              // A rows per pass
    constexpr int AP = ARPP >= 32 ? 1 : 32 / ARPP;
    constexpr int BTPR = BN / 4;                  // threads per B row (float4 each)
    constexpr int BRPP = 256 / BTPR;              // B rows per pass
    constexpr int BP = 32 / BRPP;                 // passes for the 32 pixel rows
    __shared__ __attribute__((aligned(16))) float As[32 * BM];
    __shared__ __attribute__((aligned(16))) float Bs[32 * BN];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int l31 = lane & 31, lh = lane >> 5;

    int t = blockIdx.x;
    const int cit = t % a.ci_tiles; t /= a.ci_tiles;
    const int tap = t % a.ntaps;
    const int cot = t / a.ntaps;
    const int co0 = cot * BM, ci0 = cit * BN;
    const int fr = tap / a.KW, fq = tap - fr * a.KW;
    // pixel range of this workgroup: never crosses a pyramid level, so the level's geometry is wave-uniform
    const int m_begin = a.is_gemm ? blockIdx.y * a.rows_per_split : a.r_begin[blockIdx.y];
    const int m_end = a.is_gemm ? min(a.M, m_begin + a.rows_per_split) : a.r_end[blockIdx.y];

    constexpr unsigned OOB = 0xC0000000u;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, (short)0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, (short)0, (int)a.dy_bytes, 0x00020000);

    // A rows (dY): ATPR threads per pixel row of BM channels
    const int arow = tid / ATPR, ac4 = tid % ATPR;
    const bool a_col_ok = co0 + ac4 * 4 < a.Cout && arow < 32;
    // B rows (X): BTPR threads per pixel row
    const int brow = tid / BTPR, bc4 = tid % BTPR;
    const bool b_col_ok = ci0 + bc4 * 4 < a.Cin;

    const int gs = a.is_gemm ? 0 : a.r_seg[blockIdx.y];
    const int g_Wo = a.Wo[gs], g_hw = a.Ho[gs] * a.Wo[gs], g_H = a.H[gs], g_W = a.W[gs], g_mout = a.m_out[gs], g_min = a.m_in[gs];
    const unsigned g_mgh = a.mg_hw[gs], g_mgw = a.mg_w[gs];
    const int g_shh = a.sh_hw[gs], g_shw = a.sh_w[gs];

    float4 ra[AP], rb[BP];
    auto load_tile = [&](int m0) {
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int m = m0 + arow + ARPP * i;
            const unsigned off = ((unsigned)m * (unsigned)a.dy_cs + (unsigned)(a.dy_co + co0 + ac4 * 4)) * 4u;
            ra[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(yrsrc, (int)((m < m_end && a_col_ok) ? off : OOB), 0, 0));
        }
#pragma unroll
        for (int i = 0; i < BP; ++i) {
            const int m = m0 + brow + BRPP * i;
            unsigned off = OOB;
            if (m < m_end && b_col_ok) {
                if (a.is_gemm) {
                    off = ((unsigned)m * (unsigned)a.x_cs + (unsigned)(a.x_co + ci0 + bc4 * 4)) * 4u;
                } else {
                    const int Wo = g_Wo, hw = g_hw, H = g_H, W = g_W, local = m - g_mout, m_in = g_min;
                    const unsigned mgh = g_mgh, mgw = g_mgw;
                    const int shh = g_shh, shw = g_shw;
                    const int n = fast_div(local, mgh, shh);
                    const int rem = local - n * hw;
                    const int ho = fast_div(rem, mgw, shw), wo = rem - ho * Wo;
                    const int hi = ho * a.stride - a.pad + fr * a.dil, wi = wo * a.stride - a.pad + fq * a.dil;
                    if ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W)
                        off = ((unsigned)(m_in + (n * H + hi) * W + wi) * (unsigned)a.x_cs + (unsigned)(a.x_co + ci0 + bc4 * 4)) * 4u;
                }
            }
            rb[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)off, 0, 0));
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < AP; ++i)
            if (ARPP <= 32 || arow < 32) *reinterpret_cast<float4*>(As + (arow + ARPP * i) * BM + ac4 * 4) = ra[i];
#pragma unroll
        for (int i = 0; i < BP; ++i) *reinterpret_cast<float4*>(Bs + (brow + BRPP * i) * BN + bc4 * 4) = rb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    if (m_begin < m_end) {
        load_tile(m_begin);
        for (int m0 = m_begin; m0 < m_end; m0 += 32) {
            __syncthreads();                 // previous tile fully consumed
            store_tile();
            __syncthreads();
            if (m0 + 32 < m_end) load_tile(m0 + 32);
            const float* Ab = As + wm * (TM * 32) + l31;
            const float* Bb = Bs + wn * (TN * 32) + l31;
            __builtin_amdgcn_s_setprio(1);
            float fa[2][TM], fb[2][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[0][i] = Ab[lh * BM + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[0][j] = Bb[lh * BN + j * 32];
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const int cur = p & 1, nxt = cur ^ 1;
                if (p + 1 < 16) {                 // operands of the next k-step are in flight while this one multiplies
                    const int k = 2 * (p + 1) + lh;
#pragma unroll
                    for (int i = 0; i < TM; ++i) fa[nxt][i] = Ab[k * BM + i * 32];
#pragma unroll
                    for (int j = 0; j < TN; ++j) fb[nxt][j] = Bb[k * BN + j * 32];
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i], fb[cur][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);        // LDS reads of step p+1 ...
                __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);  // ... ahead of the MFMAs of step p
            }
            __builtin_amdgcn_s_setprio(0);
        }
    }

    // C[row = co][col = ci]: reg e of lane l is row (e&3) + 8*(e>>2) + 4*(l>>5), col l&31 -> 128-byte runs along ci
    float* out = a.out + (size_t)blockIdx.y * a.slab;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int ci = ci0 + wn * (TN * 32) + j * 32 + l31;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = co0 + wm * (TM * 32) + i * 32 + 4 * lh + (e & 3) + 8 * (e >> 2);
                if (co < a.Cout && ci < a.Cin) out[(size_t)co * a.Ktot + tap * a.Cin + ci] = acc[i][j][e];
            }
        }
}

// ---- the same weight gradient with f16 operands (FD_PREC_F16: torch.autocast's arithmetic, train.py:175-181): dY and X are rounded to f16
// on their way to LDS and multiplied on v_mfma_f32_32x32x16_f16 with fp32 accumulation.  The reduction index of this GEMM is the PIXEL,
// and both operands arrive pixel-major ([pixel][channel] rows, coalesced along channels), while the MFMA wants 8 consecutive k (pixels)
// of one channel per lane: the hardware transposing LDS read of gfx950 (ds_read_b64_tr_b16: a 16-lane group reads 4 rows x 16 columns
// of 16-bit elements and receives them column-major) turns the staged [pixel][channel] image into operands with no shuffles.
// LDS image: plain 256-byte rows (128 halves per pixel), 16-byte chunks XORed with ((row & 3) << 2 | (row >> 2) & 3) -- conflict-free for
// the 8-byte staging writes and for the transposed reads (cdna guide T10, layout (b)).  128 x 128 tiles only (narrower layers are masked).
typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));

__device__ __forceinline__ int wg16_off(int row, int chunk) { return 256 * row + 16 * (chunk ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

// Round 5: K-tiles of 64 pixels in two LDS buffers (one barrier per 16 MFMAs; the first form had 32-pixel tiles, one buffer, two barriers per 8 MFMAs), 16-byte
// fetches (eight channels per lane) and, as in fd_conv_f16.hip, the parking stores of tile t + 1 and the fetches of tile t + 2 one behind each MFMA of tile t.
template <bool X16, bool DY16, bool GEMM>
__global__ __launch_bounds__(256, 2) void conv_wgrad_f16_kernel(WgradArgs a) {
    constexpr int TM = 2, TN = 2, WAVES_N = 2, BM = 128, BN = 128;
    constexpr int BUF = 2 * 64 * 256;                                 // bytes per buffer: [64 pixels][128 co] f16, then [64 pixels][128 ci] f16
    extern __shared__ __attribute__((aligned(16))) char wsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int l31 = lane & 31, lh = lane >> 5;

    int t = blockIdx.x;
    const int cit = t % a.ci_tiles; t /= a.ci_tiles;
    const int tap = t % a.ntaps;
    const int cot = t / a.ntaps;
    const int co0 = cot * BM, ci0 = cit * BN;
    const int fr = tap / a.KW, fq = tap - fr * a.KW;
    const int m_begin = GEMM ? blockIdx.y * a.rows_per_split : a.r_begin[blockIdx.y];
    const int m_end = GEMM ? min(a.M, m_begin + a.rows_per_split) : a.r_end[blockIdx.y];

    constexpr unsigned OOB = 0xC0000000u;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, (short)0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, (short)0, (int)a.dy_bytes, 0x00020000);

    // staging: 16 threads per pixel row (8 channels each: 16 bytes of an f16 map, 2 x 16 of an fp32 one), 16 rows per pass, 4 passes for the 64 pixels of a K-tile.
    // Channels past Cout / Cin inside a lane's eight are fetched as they lie (the descriptor bounds the buffer): they only reach rows / columns of the tile that
    // are never stored.
    const int srow = tid >> 4, sc8 = tid & 15;
    const bool a_col_ok = co0 + sc8 * 8 < a.Cout, b_col_ok = ci0 + sc8 * 8 < a.Cin;
    const int gs = GEMM ? 0 : a.r_seg[blockIdx.y];
    const int g_Wo = a.Wo[gs], g_hw = a.Ho[gs] * a.Wo[gs], g_H = a.H[gs], g_W = a.W[gs], g_mout = a.m_out[gs], g_min = a.m_in[gs];
    const unsigned g_mgh = a.mg_hw[gs], g_mgw = a.mg_w[gs];
    const int g_shh = a.sh_hw[gs], g_shw = a.sh_w[gs];
    const unsigned a_c = (unsigned)(a.dy_co + co0 + sc8 * 8), b_c = (unsigned)(a.x_co + ci0 + sc8 * 8);

    float4 ra[4], ra2[DY16 ? 1 : 4], rb[4], rb2[X16 ? 1 : 4];
    auto load_a = [&](int m0, int i) {
        const int m = m0 + srow + 16 * i;
        const unsigned off = ((unsigned)m * (unsigned)a.dy_cs + a_c) << (DY16 ? 1 : 2);
        const unsigned o = ((m < m_end) & a_col_ok) ? off : OOB;
        ra[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(yrsrc, (int)o, 0, 0));
        if constexpr (!DY16) ra2[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(yrsrc, (int)o, 16, 0));
    };
    auto fdiv = [](int n, unsigned mg, int sh) {      // fast_div() without its branch (magic 0 = divisor 1): nothing may split the MFMA block
        const int q = (int)(__umulhi((unsigned)n, mg) >> sh);
        return __builtin_amdgcn_readfirstlane((int)(mg != 0)) ? q : n;
    };
    auto load_b = [&](int m0, int i) {
        const int m = m0 + srow + 16 * i;
        unsigned row = (unsigned)m;
        bool ok = (m < m_end) & b_col_ok;
        if constexpr (!GEMM) {
            const int local = m - g_mout;
            const int n = fdiv(local, g_mgh, g_shh);
            const int rem = local - n * g_hw;
            const int ho = fdiv(rem, g_mgw, g_shw), wo = rem - ho * g_Wo;
            const int hi = ho * a.stride - a.pad + fr * a.dil, wi = wo * a.stride - a.pad + fq * a.dil;
            ok = ok & ((unsigned)hi < (unsigned)g_H) & ((unsigned)wi < (unsigned)g_W);
            row = (unsigned)(g_min + (n * g_H + hi) * g_W + wi);
        }
        const unsigned o = ok ? (row * (unsigned)a.x_cs + b_c) << (X16 ? 1 : 2) : OOB;
        rb[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)o, 0, 0));
        if constexpr (!X16) rb2[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)o, 16, 0));
    };
    auto h8_of = [&](const float4& lo4, const float4& hi4, bool is16) -> float4 {      // the lane's eight channels as f16 (fp32 maps: rounded to nearest even)
        if (is16) return lo4;
        const f32x4 lo = {lo4.x, lo4.y, lo4.z, lo4.w}, hi = {hi4.x, hi4.y, hi4.z, hi4.w};
        const h4 l4 = __builtin_convertvector(lo, h4), u4 = __builtin_convertvector(hi, h4);
        const h8 p = __builtin_shufflevector(l4, u4, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(float4, p);
    };
    auto park_a = [&](char* buf, int i) { *reinterpret_cast<float4*>(buf + wg16_off(srow + 16 * i, sc8)) = h8_of(ra[i], ra2[DY16 ? 0 : i], DY16); };
    auto park_b = [&](char* buf, int i) { *reinterpret_cast<float4*>(buf + 64 * 256 + wg16_off(srow + 16 * i, sc8)) = h8_of(rb[i], rb2[X16 ? 0 : i], X16); };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // transposed-read addresses: lane 4q + p of the 16-lane group g supplies row q, 8-byte piece p of its block; group g holds channels
    // 16 (g & 1) .. + 15 of the sub-tile and the k-half g >> 1 (= lane >> 5, as the MFMA operand layout wants)
    const int grp = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
    auto tr_read = [&](const char* base, int sub_chunk0, int ks) -> h8 {
        const int chunk = sub_chunk0 + 2 * (grp & 1) + (tp >> 1);
        const int r0 = 16 * ks + 8 * (grp >> 1) + tq;
        const fp16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
            (__attribute__((address_space(3))) fp16x4_t*)(base + wg16_off(r0, chunk) + 8 * (tp & 1)));
        const fp16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
            (__attribute__((address_space(3))) fp16x4_t*)(base + wg16_off(r0 + 4, chunk) + 8 * (tp & 1)));
        const h4 l4 = __builtin_bit_cast(h4, lo), h4_ = __builtin_bit_cast(h4, hi);
        return h8{l4[0], l4[1], l4[2], l4[3], h4_[0], h4_[1], h4_[2], h4_[3]};
    };

    if (m_begin < m_end) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { load_a(m_begin, i); load_b(m_begin, i); }
#pragma unroll
        for (int i = 0; i < 4; ++i) { park_a(wsm, i); park_b(wsm, i); }
#pragma unroll
        for (int i = 0; i < 4; ++i) { load_a(m_begin + 64, i); load_b(m_begin + 64, i); }
        __syncthreads();
        int cur = 0;
        for (int m0 = m_begin; m0 < m_end; m0 += 64) {
            // multiply the tile at m0 (buffer cur); behind its MFMAs park the registers (tile m0 + 64) in the other buffer and refill them with the tile at m0 + 128
            const char* Ab = wsm + cur * BUF;
            const char* Bb = Ab + 64 * 256;
            char* Pw = wsm + (cur ^ 1) * BUF;
            h8 fa[2][TM], fb[2][TN];
            auto frags = [&](int ks, int q) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[q][i] = tr_read(Ab, (wm * TM + i) * 4, ks);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[q][j] = tr_read(Bb, (wn * TN + j) * 4, ks);
            };
            frags(0, 0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks < 3) frags(ks + 1, (ks + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ks & 1][i], fb[ks & 1][j], acc[i][j], 0, 0, 0);
                        const int item = i * TN + j;          // row ks of the four: park A, refill A, park B, refill B
                        if (item == 0) park_a(Pw, ks);
                        else if (item == 1) load_a(m0 + 128, ks);
                        else if (item == 2) park_b(Pw, ks);
                        else load_b(m0 + 128, ks);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_setprio(0);
            __syncthreads();
            cur ^= 1;
        }
    }

    float* out = a.out + (size_t)blockIdx.y * a.slab;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int ci = ci0 + wn * (TN * 32) + j * 32 + l31;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = co0 + wm * (TM * 32) + i * 32 + 4 * lh + (e & 3) + 8 * (e >> 2);
                if (co < a.Cout && ci < a.Cin) out[(size_t)co * a.Ktot + tap * a.Cin + ci] = acc[i][j][e];
            }
        }
}

template <bool X16, bool DY16, bool GEMM>
static void launch_wgrad_f16_g(const WgradArgs& a, dim3 grid, hipStream_t stream) {
    constexpr int lds = 2 * 2 * 64 * 256;
    auto kern = conv_wgrad_f16_kernel<X16, DY16, GEMM>;
    static std::atomic<unsigned> attr_mask{0};
    fd_set_max_lds_once(attr_mask, reinterpret_cast<const void*>(kern), lds);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, a);
}

template <bool X16, bool DY16>
static void launch_wgrad_f16(const WgradArgs& a, dim3 grid, hipStream_t stream) {
    if (a.is_gemm) launch_wgrad_f16_g<X16, DY16, true>(a, grid, stream);
    else launch_wgrad_f16_g<X16, DY16, false>(a, grid, stream);
}

// ordered sum of the split slabs (+ optional per-output-channel scale) written as OHWI (layout 0) or OIHW (layout 1)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, long n4,
                                                            int nsplit, long slab, const float* __restrict__ scale, int layout,
                                                            int Cin, int ntaps) {
    const int Ktot = Cin * ntaps;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 v = reinterpret_cast<const float4*>(ws)[i];
        int s = 1;
        for (; s + 8 <= nsplit; s += 8) {             // eight slabs in flight, added in slab order (a plain loop waited for every slab in turn)
            float4 u[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) u[q] = reinterpret_cast<const float4*>(ws + (s + q) * slab)[i];
#pragma unroll
            for (int q = 0; q < 8; ++q) { v.x += u[q].x; v.y += u[q].y; v.z += u[q].z; v.w += u[q].w; }
        }
        for (; s < nsplit; ++s) {
            const float4 u = reinterpret_cast<const float4*>(ws + s * slab)[i];
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        const long e = 4 * i;                         // element index in [co][tap][ci]; Cin % 4 == 0 keeps a quad inside one tap
        const int co = (int)(e / Ktot);
        if (scale) { const float sc = scale[co]; v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc; }
        if (layout == 0) {
            reinterpret_cast<float4*>(dw)[i] = v;
        } else {
            const int r = (int)(e - (long)co * Ktot);
            const int tap = r / Cin, ci = r - tap * Cin;
            float* o = dw + ((long)co * Cin + ci) * ntaps + tap;
            o[0] = v.x; o[ntaps] = v.y; o[2 * ntaps] = v.z; o[3 * ntaps] = v.w;
        }
    }
}

// element i of the fp32 packing [N][K/32][taps][32] -> its (hi, lo) f16 pair in the FD_PREC_F16X3 / FD_PREC_F16 packing [..][2][32]
__device__ __forceinline__ void pack_store_f16(float* out, long i, float v) {
    _Float16* oh = reinterpret_cast<_Float16*>(out) + (i >> 5) * 64 + (i & 31);
    const _Float16 hi = (_Float16)v;
    oh[0] = hi;
    oh[32] = (_Float16)((v - (float)hi) * 2048.0f);
}

// framework OIHW weights -> the conv kernel's [N][K/32][KH][KW][32] layout in one pass.
// mode 0: forward weights (N = Cout, K = Cin).  mode 1: weights of the stride-1 data-gradient conv (N = Cin, K = Cout):
// w'[ci][co][r][q] = w[co][ci][KH-1-r][KW-1-q] * scale[co].
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ w, const float* __restrict__ scale,
                                                           float* __restrict__ out, int Cout, int Cin, int KH, int KW, int mode,
                                                           long total) {
    const int taps = KH * KW;
    const int N = (mode & 1) ? Cin : Cout, K = (mode & 1) ? Cout : Cin;
    const int ksh = (mode & 16) ? 6 : 5;          // mode | 16: K-tiles of 64 channels, f16 only (fd_conv_f16.hip)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cl = (int)(i & ((1 << ksh) - 1));
        long t = i >> ksh;
        const int tap = (int)(t % taps); t /= taps;
        const int chunk = (int)(t % (K >> ksh));
        const int n = (int)(t / (K >> ksh));
        const int k = (chunk << ksh) + cl;
        float v;
        if ((mode & 1) == 0) {
            v = w[((long)n * Cin + k) * taps + tap];
        } else {
            v = w[((long)k * Cin + n) * taps + (taps - 1 - tap)];
            if (scale) v *= scale[k];
        }
        if (mode & 16) reinterpret_cast<_Float16*>(out)[i] = (_Float16)v;
        else if (mode & 4) pack_store_f16(out, i, v); else out[i] = v;
    }
    (void)N;
}

extern "C" int32_t fd_pack_conv_weight_f32(const float* w, const float* scale, float* out, int32_t Cout, int32_t Cin,
                                           int32_t KH, int32_t KW, int32_t mode, fd_stream_t stream) {
    FD_REQUIRE(w && out && Cout >= 1 && Cin >= 1 && KH >= 1 && KW >= 1 && (mode & ~21) == 0 && (mode & 20) != 20, FD_E_INVAL,
               "fd_pack_conv_weight: bad arguments");
    FD_REQUIRE(((mode & 1) == 0 ? Cin : Cout) % ((mode & 16) ? 64 : 32) == 0, FD_E_UNSUPPORTED,
               "fd_pack_conv_weight: the reduction width (%d) must be a multiple of %d", (mode & 1) == 0 ? Cin : Cout, (mode & 16) ? 64 : 32);
    const long total = (long)Cout * Cin * KH * KW;
    long g = (total + 255) / 256;
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w, scale, out, Cout, Cin, KH, KW,
                       mode, total);
    FD_CHECK_LAUNCH("fd_pack_conv_weight_f32");
    return FD_OK;
}

// Many weight tensors in one launch (the train step re-packs ~150 of them after every optimizer update): a job table in
// device memory, one workgroup range per job (blockIdx.y = job, grid-stride over its elements).
__global__ __launch_bounds__(256) void pack_weight_batch_kernel(const fd_pack_job* __restrict__ jobs) {
    const fd_pack_job j = jobs[blockIdx.y];
    if (j.mode == 2 || j.mode == 3) {       // Winograd packing (fd_conv_wino.hip): mode 2 = forward, 3 = data-gradient weights; one (n, k) filter per thread
        const int N = j.mode == 2 ? j.Cout : j.Cin, K = j.mode == 2 ? j.Cin : j.Cout;
        const long total = (long)((N + 31) & ~31) * K;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
            const int n = (int)(i / K), k = (int)(i - (long)n * K);
            fd_wino_pack_one(j.w, j.scale, j.out, N, K, j.mode - 2, n, k);
        }
        return;
    }
    if (j.mode == 8 || j.mode == 9) {       // F(4x4, 3x3) packing (fd_conv_wino4.hip): mode 8 = forward, 9 = data-gradient weights
        const int N = j.mode == 8 ? j.Cout : j.Cin, K = j.mode == 8 ? j.Cin : j.Cout;
        const long total = (long)((N + 31) & ~31) * K;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
            const int n = (int)(i / K), k = (int)(i - (long)n * K);
            fd_wino4_pack_one(j.w, j.scale, j.out, N, K, j.mode - 8, n, k);
        }
        return;
    }
    const int taps = j.KH * j.KW;
    const int K = (j.mode & 1) ? j.Cout : j.Cin;
    const long total = (long)j.Cout * j.Cin * taps;
    const float* __restrict__ w = j.w;
    const float* __restrict__ scale = j.scale;
    const int ksh = (j.mode & 16) ? 6 : 5;        // mode | 16: the f16 K-tile-64 packing of fd_conv_f16.hip
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cl = (int)(i & ((1 << ksh) - 1));
        long t = i >> ksh;
        const int tap = (int)(t % taps); t /= taps;
        const int chunk = (int)(t % (K >> ksh));
        const int n = (int)(t / (K >> ksh));
        const int k = (chunk << ksh) + cl;
        float v;
        if ((j.mode & 1) == 0) {
            v = w[((long)n * j.Cin + k) * taps + tap];
        } else {
            v = w[((long)k * j.Cin + n) * taps + (taps - 1 - tap)];
            if (scale) v *= scale[k];
        }
        if (j.mode & 16) reinterpret_cast<_Float16*>(j.out)[i] = (_Float16)v;
        else if (j.mode & 4) pack_store_f16(j.out, i, v); else j.out[i] = v;
    }
}

extern "C" int32_t fd_pack_conv_weights_batch_f32(const fd_pack_job* jobs_dev, int32_t n_jobs, int64_t max_elems,
                                                  fd_stream_t stream) {
    FD_REQUIRE(jobs_dev && n_jobs >= 1 && n_jobs <= 65535 && max_elems >= 1, FD_E_INVAL, "fd_pack_conv_weights_batch: bad arguments");
    long gx = (max_elems + 256L * 8 - 1) / (256L * 8);       // ~8 elements per thread for the largest job
    if (gx > 1024) gx = 1024;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(pack_weight_batch_kernel, dim3((unsigned)gx, (unsigned)n_jobs), dim3(256), 0, (hipStream_t)stream, jobs_dev);
    FD_CHECK_LAUNCH("fd_pack_conv_weights_batch_f32");
    return FD_OK;
}

static void magic_div(int d, unsigned& m, int& sh) {
    if (d <= 1) { m = 0; sh = 0; return; }
    int l = 0;
    while ((1L << l) < d) ++l;
    m = (unsigned)(((1ULL << (31 + l)) + (unsigned long long)d - 1) / (unsigned long long)d);
    sh = l - 1;
}

static int wgrad_splits(long M, int tiles, bool gemm) {
    // Three workgroups are resident per CU (VGPR-bound).  Pick the split count whose total workgroup count fills whole
    // rounds of 256 CUs best (1026 workgroups would run as 1024 + 2), preferring fewer splits.  Measured on MI355X
    // (tools/time_wgrad.py): 1x1 layers want <= 768 workgroups with >= 10 K-tiles of 32 pixels each (their per-tile
    // work is small next to the slab write + ordered reduce), 3x3 layers are best near 756-1017 workgroups.
    const long min_rows = gemm ? 320 : 256;
    const long max_wg = gemm ? 768 : 1024;
    long maxs = M / min_rows;
    if (maxs < 1) maxs = 1;
    if (maxs > 256) maxs = 256;
    int best = 1;
    double best_score = -1.0;
    for (long s = 1; s <= maxs; ++s) {
        const long wg = (long)tiles * s;
        if (wg > max_wg && s > 1) break;
        const long rounds = (wg + 255) / 256;
        const double fill = (double)wg / (double)(rounds * 256);          // CU balance
        const double par = wg >= 512 ? 0.97 : (wg >= 256 ? 0.90 : 0.90 * wg / 256.0);   // latency hiding
        const double score = fill * par;
        if (score > best_score + 1e-9) { best_score = score; best = (int)s; }
    }
    return best;
}

// FD_PREC_F16 (round 5's kernel: 64-pixel K-tiles at ~0.3 us each): the slab write + ordered reduce of a split cost as much as ~20 K-tiles, so ranges of
// ~2048 pixels (32 K-tiles) per workgroup, shortened to no less than 512 only while fewer than 128 workgroups would run (tools/time_wgrad.py F16=1 sweep:
// 128>512 1x1 at 64k pixels 58 -> 37 us, 256>128 1x1 at 262k pixels 106 -> 74, 128>128 3x3 75 -> 63; the 3x3 256-wide layers unchanged).
static int wgrad_splits_f16(long M, int tiles) {
    long s = M / 2048;
    if ((long)tiles * s > 1024) s = 1024 / tiles;          // (four rounds of 256 workgroups are enough: the 3x3 256-wide layers, as the fp32 rule has them)
    if (s < 1) s = 1;
    while ((long)tiles * s < 128 && M / (s * 2) >= 512) s *= 2;
    if (s > 256) s = 256;
    return (int)s;
}

extern "C" int64_t fd_conv_wgrad_workspace_bytes(int64_t out_rows, int32_t Cin, int32_t Cout, int32_t KH, int32_t KW) {
    if (out_rows < 1 || Cin < 1 || Cout < 1 || KH < 1 || KW < 1) return -1;
    const int bn = (Cin % 128 == 0) ? 128 : 64;
    const int bm = (Cout <= 32 && bn == 128) ? 32 : 128;
    const int tiles = ((Cout + bm - 1) / bm) * KH * KW * ((Cin + bn - 1) / bn);
    const int tiles16 = ((Cout + 127) / 128) * KH * KW * ((Cin + 127) / 128);      // FD_PREC_F16: 128 x 128 tiles whatever the widths
    // upper bound over both split rules (the launcher knows stride / padding, this query does not) and both precisions
    const int ns = std::max(std::max(std::max(wgrad_splits(out_rows, tiles, false), wgrad_splits(out_rows, tiles, true)),
                                     std::max(wgrad_splits(out_rows, tiles16, false), wgrad_splits(out_rows, tiles16, true))), wgrad_splits_f16(out_rows, tiles16));
    // + FD_MAX_SEG: level-aligned ranges give every pyramid level at least one slab of its own
    return (int64_t)(ns + FD_MAX_SEG) * Cout * KH * KW * Cin * 4;
}

extern "C" int32_t fd_conv2d_bwd_weight_f32(const fd_conv_wgrad_params* p, fd_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    FD_REQUIRE(p && p->x && p->dy && p->dw && p->workspace, FD_E_INVAL, "fd_conv2d_bwd_weight: null pointer");
    FD_REQUIRE(fd_segs_ok(&p->in), FD_E_INVAL, "fd_conv2d_bwd_weight: bad segment table");
    FD_REQUIRE(p->Cin >= 4 && p->Cin % 4 == 0 && p->Cout >= 4 && p->Cout % 4 == 0, FD_E_UNSUPPORTED,
               "fd_conv2d_bwd_weight: Cin=%d / Cout=%d must be multiples of 4", p->Cin, p->Cout);
    FD_REQUIRE(p->x_cs % 4 == 0 && p->x_co % 4 == 0 && p->dy_cs % 4 == 0 && p->dy_co % 4 == 0, FD_E_INVAL,
               "fd_conv2d_bwd_weight: channel views must be 4-aligned");
    FD_REQUIRE((((uintptr_t)p->x | (uintptr_t)p->dy | (uintptr_t)p->dw | (uintptr_t)p->workspace) & 15) == 0, FD_E_INVAL,
               "fd_conv2d_bwd_weight: pointers not 16-byte aligned");
    FD_REQUIRE(p->KH >= 1 && p->KW >= 1 && p->stride >= 1 && p->dil >= 1 && p->pad >= 0, FD_E_INVAL, "fd_conv2d_bwd_weight: bad geometry");
    WgradArgs a;
    a.x = p->x; a.dy = p->dy;
    a.x_cs = p->x_cs; a.x_co = p->x_co; a.dy_cs = p->dy_cs; a.dy_co = p->dy_co;
    a.Cin = p->Cin; a.Cout = p->Cout; a.KW = p->KW; a.stride = p->stride; a.pad = p->pad; a.dil = p->dil;
    a.ntaps = p->KH * p->KW;
    a.Ktot = a.ntaps * p->Cin;
    a.nseg = p->in.nseg;
    long mo = 0;
    for (int s = 0; s < FD_MAX_SEG; ++s) {
        if (s < p->in.nseg) {
            a.H[s] = p->in.H[s]; a.W[s] = p->in.W[s];
            a.Ho[s] = (p->in.H[s] + 2 * p->pad - p->dil * (p->KH - 1) - 1) / p->stride + 1;
            a.Wo[s] = (p->in.W[s] + 2 * p->pad - p->dil * (p->KW - 1) - 1) / p->stride + 1;
            FD_REQUIRE(a.Ho[s] >= 1 && a.Wo[s] >= 1, FD_E_INVAL, "fd_conv2d_bwd_weight: empty output");
            a.m_in[s] = p->in.m_start[s];
            a.m_out[s] = (int)mo;
            mo += (long)p->in.batch * a.Ho[s] * a.Wo[s];
        } else { a.H[s] = a.W[s] = a.Ho[s] = a.Wo[s] = 1; a.m_in[s] = 0; a.m_out[s] = (int)mo; }
        magic_div(a.Ho[s] * a.Wo[s], a.mg_hw[s], a.sh_hw[s]);
        magic_div(a.Wo[s], a.mg_w[s], a.sh_w[s]);
    }
    a.m_out[FD_MAX_SEG] = (int)mo;
    FD_REQUIRE(mo > 0 && mo < (1L << 31), FD_E_INVAL, "fd_conv2d_bwd_weight: row count out of range");
    a.M = (int)mo;
    FD_REQUIRE((p->io_f16 & ~3) == 0 && (!p->io_f16 || p->precision == FD_PREC_F16), FD_E_UNSUPPORTED,
               "fd_conv2d_bwd_weight: io_f16 (f16 operand maps) needs FD_PREC_F16");
    a.x16 = p->io_f16 & 1; a.dy16 = (p->io_f16 >> 1) & 1;
    const long xb = (long)p->in.m_start[p->in.nseg] * p->x_cs * (a.x16 ? 2 : 4), yb = mo * p->dy_cs * (a.dy16 ? 2 : 4);
    FD_REQUIRE(xb < 0xC0000000L && yb < 0xC0000000L, FD_E_UNSUPPORTED, "fd_conv2d_bwd_weight: buffer exceeds 3 GiB");
    a.x_bytes = (unsigned)xb; a.dy_bytes = (unsigned)yb;
    a.is_gemm = (p->KH == 1 && p->KW == 1 && p->stride == 1 && p->pad == 0) ? 1 : 0;
    // FD_PREC_F16: f16 operands on 128 x 128 tiles
    FD_REQUIRE(p->precision == FD_PREC_F32 || p->precision == FD_PREC_F16, FD_E_INVAL, "fd_conv2d_bwd_weight: precision must be FD_PREC_F32 or FD_PREC_F16");
    const bool h16 = p->precision == FD_PREC_F16;       // (narrow predictors too, since round 5: their 128-row tiles fetch only the rows that exist; 189 -> ~95 us on the head's)
    const int bn = (h16 || p->Cin % 128 == 0) ? 128 : 64;
    const int bm = (p->Cout <= 32 && bn == 128) ? 32 : 128;
    a.co_tiles = (p->Cout + bm - 1) / bm;
    a.ci_tiles = (p->Cin + bn - 1) / bn;
    const int tiles = a.co_tiles * a.ntaps * a.ci_tiles;
    int nsplit = p->nsplit > 0 ? p->nsplit : h16 ? wgrad_splits_f16(mo, tiles) : wgrad_splits(mo, tiles, a.is_gemm != 0);
    FD_REQUIRE(nsplit <= 65535, FD_E_INVAL, "fd_conv2d_bwd_weight: nsplit too large");
    a.rows_per_split = (int)(((mo + nsplit - 1) / nsplit + 31) / 32 * 32);
    unsigned ny = (unsigned)((mo + a.rows_per_split - 1) / a.rows_per_split);
    if (!a.is_gemm) {
        // level-aligned pixel ranges: every level gets its share of the splits (at least one), cut into equal parts
        if (nsplit > FD_WGRAD_MAX_RANGES) nsplit = FD_WGRAD_MAX_RANGES;
        int per[FD_MAX_SEG], tot = 0;
        for (int s = 0; s < p->in.nseg; ++s) {
            const long rows = (long)a.m_out[s + 1] - a.m_out[s];
            per[s] = (int)((rows * nsplit + mo / 2) / mo);
            if (per[s] < 1) per[s] = 1;
            if (per[s] > rows / 32 + 1) per[s] = (int)(rows / 32 + 1);
            tot += per[s];
        }
        while (tot > FD_WGRAD_MAX_RANGES) {            // only reachable with nsplit forced near the cap: trim the largest
            int big = 0;
            for (int s = 1; s < p->in.nseg; ++s) if (per[s] > per[big]) big = s;
            --per[big]; --tot;
        }
        ny = 0;
        for (int s = 0; s < p->in.nseg; ++s) {
            const long rows = (long)a.m_out[s + 1] - a.m_out[s];
            const long step = ((rows + per[s] - 1) / per[s] + 31) / 32 * 32;
            for (long b = 0; b < rows; b += step) {
                a.r_begin[ny] = a.m_out[s] + (int)b;
                a.r_end[ny] = a.m_out[s] + (int)std::min(rows, b + step);
                a.r_seg[ny] = (unsigned char)s;
                ++ny;
            }
        }
        nsplit = (int)ny;
    }
    a.slab = (long)p->Cout * a.Ktot;
    FD_REQUIRE(p->workspace_bytes >= (int64_t)nsplit * a.slab * 4, FD_E_INVAL,
               "fd_conv2d_bwd_weight: workspace too small (need fd_conv_wgrad_workspace_bytes())");
    FD_REQUIRE(p->layout == 0 || p->layout == 1, FD_E_INVAL, "fd_conv2d_bwd_weight: layout must be 0 (OHWI) or 1 (OIHW)");
    const bool need_reduce = nsplit > 1 || p->layout != 0 || p->scale != nullptr;
    a.out = need_reduce ? (float*)p->workspace : p->dw;
    const dim3 grid(tiles, ny);
    if (h16) {
        if (a.x16 && a.dy16) launch_wgrad_f16<true, true>(a, grid, stream);
        else if (a.x16) launch_wgrad_f16<true, false>(a, grid, stream);
        else if (a.dy16) launch_wgrad_f16<false, true>(a, grid, stream);
        else launch_wgrad_f16<false, false>(a, grid, stream);
    }
    else if (bm == 32) hipLaunchKernelGGL((conv_wgrad_kernel<128, 32>), grid, dim3(256), 0, stream, a);
    else if (bn == 128) hipLaunchKernelGGL((conv_wgrad_kernel<128, 128>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((conv_wgrad_kernel<64, 128>), grid, dim3(256), 0, stream, a);
    FD_CHECK_LAUNCH("fd_conv2d_bwd_weight");
    if (need_reduce) {
        const long n4 = a.slab / 4;
        long g = (n4 + 255) / 256;
        if (g > 8192) g = 8192;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)g), dim3(256), 0, stream, (const float*)p->workspace, p->dw, n4,
                           (int)grid.y, a.slab, p->scale, p->layout, p->Cin, a.ntaps);
        FD_CHECK_LAUNCH("fd_conv2d_bwd_weight (reduce)");
    }
    return FD_OK;
}
