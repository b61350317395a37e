// fd_conv_wino.hip — 3x3 stride-1 'same' convolution (dilation 1 or 2) as Winograd F(2x2, 3x3) on the fp32 MFMA of gfx950.
//
// The direct implicit-GEMM kernel (fd_conv.hip) already runs the head tower at 92 % of what v_mfma_f32_32x32x2_f32 delivers at the
// clock the chip holds (130 of ~141 TFLOP/s at 2.16 GHz): in exact fp32 the only lever left is executing fewer multiplies.
// F(2x2, 3x3) computes a 2x2 output tile from a 4x4 input patch with 16 multiplies per (cin, cout) instead of 36:
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A          U = G g G^T is packed once per layer (fd_wino_pack_weights_f32),
// so the conv becomes 16 independent GEMMs  M_f[tile][cout] = sum_c V_f[tile][c] * U_f[cout][c]  (f = frequency 0..15)
// with 2.25x fewer MFMAs.  Everything stays fp32 (the transforms only add, subtract and halve): the result differs from the fma
// chain of the direct kernel by rounding, ~1e-6 relative, inside the 1e-4 parity bar, and is the same on every run.
//
// One workgroup (4 waves; 8 waves x 128 channels for wide layers, see NCH below) = 32 tiles (128 output pixels) x 64 output channels x
// all 16 frequencies:
//   * tiles are enumerated over (level, image, dilation parity class, tile row, tile column): a dilated conv is a plain one on
//     each of the dil^2 sub-lattices (h % dil, w % dil), so dilation only changes the address arithmetic;
//   * per 8-channel chunk every thread loads ONE row of one tile's 4x4 patch (4 x 16 B raw buffer loads, zero outside the
//     image), transforms along the row in registers, exchanges with the other three rows of its quad by DPP (quad_perm) for
//     the column pass and writes V[f][tile][8 c] to LDS (16 KB per chunk, double buffered): the input transform never touches
//     HBM and its 4x patch overlap is served by L1/L2;
//   * wave (fh, ch) owns frequencies 8 fh .. 8 fh + 7 and output channels 32 ch .. 32 ch + 31: 8 accumulators of 32 x 32
//     (128 VGPRs).  The U operand goes global -> registers directly in MFMA layout (packed so that one frequency block of
//     32 cout x 8 c is 1 KB contiguous, L2 resident: blocks of one XCD walk the M tiles of ONE cout tile before the next);
//   * epilogue: each wave reduces its two patch rows of frequencies to partial 2x2 outputs, the pair (fh = 0, 1) meets in LDS,
//     then y = act(v * scale + shift (+ / mask) res) with 16-byte stores, exactly the direct kernel's epilogue contract.
#include "fd_conv_common.h"

struct WinoArgs {
    const float* x; const float* u; const float* scale; const float* shift; const float* res; float* y;
    int x_cs, x_co, res_cs, res_co, y_cs, y_co;
    int Cin, Cout, dil, act, act_c0, res_mask;
    int NC;                       // 8-channel chunks
    int nseg;
    int H[FD_MAX_SEG], W[FD_MAX_SEG], TH[FD_MAX_SEG], TW[FD_MAX_SEG];   // TH x TW tiles per parity class
    int m0[FD_MAX_SEG];           // first row of the level (input rows == output rows)
    int t0[FD_MAX_SEG + 1];       // first tile of the level
    float seg_param[FD_MAX_SEG];
    int T;                        // tiles in all
    int mtiles, ntiles, mt_per;   // M tiles (32 tiles each), N tiles (64 cout), M tiles per XCD
    int nc_per;                   // 8-channel chunks per split-K slice (blockIdx.y = slice; NC when split-K is off)
    long slice_stride;            // elements between consecutive split-K slabs of the workspace y points to
    unsigned x_bytes, u_bytes;
    float* gn_stats; int gn_G, gn_cg;   // row-group statistics of the stored output (fd_conv_common.h: fd_gn_rowstats)
};

#define WINO_TB 32     // tiles per workgroup
#define WINO_NB 64     // output channels per workgroup (NCH = 2 waves-pairs; 128 with NCH = 4)
#define WINO_KC 8      // channels per chunk

struct TilePos { int s, n, h0, w0; bool ok; };

__device__ __forceinline__ TilePos wino_decode(const WinoArgs& a, int t) {
    TilePos p;
    p.ok = t < a.T;
    if (!p.ok) t = 0;
    int s = 0;
#pragma unroll
    for (int i = 1; i < FD_MAX_SEG; ++i)
        if (i < a.nseg && t >= a.t0[i]) s = i;
    const int local = t - a.t0[s];
    const int ct = a.TH[s] * a.TW[s];
    const int tpi = a.dil * a.dil * ct;
    const int n = local / tpi, r = local - n * tpi;
    const int cls = r / ct, r2 = r - cls * ct;
    const int ti = r2 / a.TW[s], tj = r2 - ti * a.TW[s];
    const int ca = cls / a.dil, cb = cls - ca * a.dil;
    p.s = s; p.n = n;
    p.h0 = ca + 2 * a.dil * ti;
    p.w0 = cb + 2 * a.dil * tj;
    return p;
}

__device__ __forceinline__ float dpp_quad_2211(float v) {   // lane i of a quad reads lane {2, 2, 1, 1}[i]
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x5A, 0xF, 0xF, true));
}

// NCH = output-channel blocks of 32 per workgroup: 2 (4 waves, 64 channels, two workgroups per CU) or 4 (8 waves, 128 channels, one
// workgroup per CU: the V tile -- patch loads, transform, LDS writes -- is shared by twice as many MFMAs and the input is streamed half
// as many times; the two groups of four waves stage alternate chunks).
template <int TAG, int NCH>
__global__ __launch_bounds__(NCH * 128, NCH == 2 ? 2 : 1) void conv3x3_wino_kernel(WinoArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Vs = reinterpret_cast<float*>(smem);          // [2 stages][16 f][32 tiles][8 c]; the epilogue reuses all 64 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fh = wave & 1, ch = wave >> 1;
    const int l31 = lane & 31, lh = lane >> 5;

    // XCD-aware order: XCD x owns M tiles [x * mt_per, (x + 1) * mt_per) and walks them cout tile by cout tile, so the 64 workgroups
    // resident on an XCD share one cout tile's U (<= 1 MB at Cin = 256) in its L2 while the input streams through
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int mt_lo = xcd * a.mt_per;
    const int cnt = min(a.mtiles - mt_lo, a.mt_per);
    if (cnt <= 0 || idx >= cnt * a.ntiles) return;
    const int nt = idx / cnt, mt = mt_lo + (idx - nt * cnt);
    const int tile0 = mt * WINO_TB, n0 = nt * (32 * NCH);
    // who stages V: with 4 waves everyone, every chunk; with 8 waves the two wave groups take turns -- group g (waves 4g .. 4g + 3) loads,
    // transforms and writes the chunks of parity g, so each wave carries half the loader work and only one patch slot is live per wave
    const int grp = wave >> 2;                           // (wave-uniform)
#define WINO_LD_ON(PAR) (NCH == 2 || grp == (PAR))
    // split-K: slice blockIdx.y owns chunks [c0, c1) and writes raw partial outputs to its slab of the workspace (the host points y at it)
    const int c0 = blockIdx.y * a.nc_per, c1 = min(a.NC, c0 + a.nc_per);

    constexpr unsigned OOB = 0xC0000000u;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, (short)0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ursrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.u, (short)0, (int)a.u_bytes, 0x00020000);

    // ---- loader role: thread = (tile lt, channel quad q, patch row pi) ----
    const int pi = tid & 3, q = (tid >> 2) & 1, lt = (tid >> 3) & 31;
    unsigned a_off[4];
    {
        const TilePos p = wino_decode(a, tile0 + lt);
        const int H = a.H[p.s], W = a.W[p.s];
        const int hh = p.h0 + (pi - 1) * a.dil;
        const bool row_ok = p.ok && (unsigned)hh < (unsigned)H;
        const int rowbase = a.m0[p.s] + (p.n * H + hh) * W;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ww = p.w0 + (j - 1) * a.dil;
            a_off[j] = (row_ok && (unsigned)ww < (unsigned)W)
                           ? ((unsigned)(rowbase + ww) * (unsigned)a.x_cs + (unsigned)(a.x_co + q * 4)) * 4u : OOB;
        }
    }
    // column pass of B^T d B across the quad: V[pi][j] = t[pi][j] + so * t[{2, 2, 1, 1}[pi]][j]; patch row 3 is kept with the
    // opposite sign (t3 - t1 = -V3: the weight packing negates U on its four frequencies, the products are unchanged)
    const float so = (pi == 1) ? 1.f : -1.f;
    // V[f = 4 pi + j][tile][8 c]: tile rows XORed with pi (conflict-free ds_write_b128 over a quad), 16-B halves with tile bit 4
    // (conflict-free ds_read_b128 of the MFMA feed)
    const int v_wr = (pi * 4 * WINO_TB + (lt ^ pi)) * WINO_KC + 4 * (q ^ ((lt >> 4) & 1));

    // ---- MFMA role ----
    // U packed [cout / 32][chunk][16 f][32 cout][8 c]: this wave's 8 frequency blocks of a chunk are 8 KB contiguous
    const int nb = (n0 >> 5) + ch;
    const bool nb_ok = nb * 32 < ((a.Cout + 31) & ~31);
    const unsigned u_off0 = nb_ok ? ((unsigned)(nb * a.NC) * 16u + 8u * fh) * 1024u + (unsigned)(l31 * 32 + lh * 16) : OOB;
    const int v_rd = (8 * fh * WINO_TB) * WINO_KC + 4 * (lh ^ ((l31 >> 4) & 1));   // + (fi * 32 + (l31 ^ row_xor(f))) * 8

    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    // Main loop, one 8-channel chunk per iteration and workgroup barrier.  With everything but the MFMAs, the V reads and the barrier
    // compiled out the loop runs at the matrix pipe's full rate (head tower: 1.01 ms = 141 TFLOP/s at the clock the chip holds);
    // measured costs on top: patch loads 11 %, transform + LDS writes 8 %, U loads 6 %, epilogue 4 %.  Tried and dropped: two chunks
    // per barrier (slower), 16-channel stages with two patch rows per thread (64 B per pixel and load instruction, half the barriers:
    // equal), L1-bypassing / non-temporal U loads (equal / 18 % slower), one workgroup per CU (13 % slower).  So:
    //   * the patch rows are fetched TWO chunks ahead (pr[2][4]; they miss L2 -- the input streams from the Infinity Cache / HBM --
    //     and one iteration of MFMAs does not cover that latency under load);
    //   * the transform of chunk cc + 1 is spread over the MFMA groups of iteration cc (row pass in group 2, one output column
    //     = 4 v_fmac with a DPP operand + one ds_write_b128 in each of groups 3..6), so its VALU issue fits the MFMAs' shadow;
    //   * the U block of (cc + 1, fi) is fetched into the registers the MFMAs of fi have just consumed; V fragments are read two
    //     frequencies ahead.
    // The issue order is pinned with sched_barriers (left alone the compiler hoists the transform -- and with it a wait on loads it
    // has just issued -- to the top of the iteration, or sinks all U prefetches to its end).  No data-dependent branches: the last
    // iterations re-fetch the last chunk and write a stage nobody reads.
    float4 pr[2][4], bq[8];
    f32x4 tr[4];
    auto load_u = [&](int cc, int fi) {
        bq[fi] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ursrc, (int)(u_off0 + ((unsigned)cc * 16u + fi) * 1024u), 0, 0));
    };
    constexpr int PLANE = WINO_TB * WINO_KC, STAGE = 16 * PLANE;
    // plane f = 8 fh + fi was written with tile rows XORed by its patch row (f >> 2) = 2 fh + (fi >> 2)
    auto read_v = [&](const float* Vb, int fi) {
        return *reinterpret_cast<const float4*>(Vb + (fi * WINO_TB + (l31 ^ (2 * fh + (fi >> 2)))) * WINO_KC);
    };
#define WINO_LOAD_PATCH(SLOT, CC)                                                                                                  \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                                              \
        pr[SLOT][j_] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)(a_off[j_] + (unsigned)(CC) * 32u), 0, 0))
    // row pass of B^T d B in registers.  (The patch registers pass through an empty volatile asm: the transform is a pure function of
    // them, and without this pin instruction selection starts it -- and the wait on the loads -- at the top of the iteration.)
#define WINO_ROW_PASS(SLOT)                                                                                                        \
    do {                                                                                                                           \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                                          \
            asm volatile("" : "+v"(pr[SLOT][j_].x), "+v"(pr[SLOT][j_].y), "+v"(pr[SLOT][j_].z), "+v"(pr[SLOT][j_].w));           \
        const f32x4 d0_ = {pr[SLOT][0].x, pr[SLOT][0].y, pr[SLOT][0].z, pr[SLOT][0].w};                                           \
        const f32x4 d1_ = {pr[SLOT][1].x, pr[SLOT][1].y, pr[SLOT][1].z, pr[SLOT][1].w};                                           \
        const f32x4 d2_ = {pr[SLOT][2].x, pr[SLOT][2].y, pr[SLOT][2].z, pr[SLOT][2].w};                                           \
        const f32x4 d3_ = {pr[SLOT][3].x, pr[SLOT][3].y, pr[SLOT][3].z, pr[SLOT][3].w};                                           \
        tr[0] = d0_ - d2_; tr[1] = d1_ + d2_; tr[2] = d2_ - d1_; tr[3] = d1_ - d3_;                                               \
    } while (0)
    // column pass across the quad + LDS write of output column j: v = ss * t + so * t[lane {2, 2, 1, 1} of the quad] (v_fmac with DPP)
    // gfx9 hazard: a VALU write of a VGPR followed by a DPP read of it needs 2 wait states, and the hazard recogniser does not see into
    // inline asm.  The four DPP operations of one output column are ONE asm statement that opens with `s_nop 1`: whatever the compiler
    // does to tr[] before the statement (the row pass itself, or a register copy it decides to insert), the wait states sit between that
    // write and the first DPP read, independent of how the surrounding code is scheduled.
    auto col_store = [&](int stage, int j) {
        float r0 = tr[j][0], r1 = tr[j][1], r2 = tr[j][2], r3 = tr[j][3];
        asm volatile("s_nop 1\n\t"
                     "v_fmac_f32_dpp %0, %4, %8 quad_perm:[2,2,1,1] row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f32_dpp %1, %5, %8 quad_perm:[2,2,1,1] row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f32_dpp %2, %6, %8 quad_perm:[2,2,1,1] row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f32_dpp %3, %7, %8 quad_perm:[2,2,1,1] row_mask:0xf bank_mask:0xf"
                     : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3)
                     : "v"(tr[j][0]), "v"(tr[j][1]), "v"(tr[j][2]), "v"(tr[j][3]), "v"(so));
        const f32x4 v = {r0, r1, r2, r3};
        *reinterpret_cast<f32x4*>(Vs + stage * STAGE + v_wr + j * PLANE) = v;
    };
    // one iteration: MFMAs of chunk cc (LDS stage cc & 1), transform of chunk cc + 1 (fetched an iteration ago into pr[SLOT ^ 1]) into the
    // other stage, fetch of chunk cc + 2 into pr[SLOT] (SLOT = cc & 1, a compile-time constant: the loop is unrolled by two)
#define WINO_ITER(SLOT, CC)                                                                                                        \
    do {                                                                                                                           \
        const int cc_ = (CC);                                                                                                      \
        const int cn1_ = min(cc_ + 1, c1 - 1), cn2_ = min(cc_ + 2, c1 - 1);                                                        \
        const float* Vb = Vs + (SLOT) * STAGE + v_rd;                                                                              \
        float4 fa[8];                                                                                                              \
        if (WINO_LD_ON(SLOT)) { WINO_LOAD_PATCH(SLOT, cn2_); }                                                                     \
        fa[0] = read_v(Vb, 0);                                                                                                     \
        fa[1] = read_v(Vb, 1);                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
        __builtin_amdgcn_s_setprio(1);                                                                                             \
        _Pragma("unroll") for (int fi = 0; fi < 8; ++fi) {                                                                        \
            const float4 fb = bq[fi];                                                                                              \
            acc[fi] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[fi].x, fb.x, acc[fi], 0, 0, 0);                                      \
            acc[fi] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[fi].y, fb.y, acc[fi], 0, 0, 0);                                      \
            acc[fi] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[fi].z, fb.z, acc[fi], 0, 0, 0);                                      \
            acc[fi] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[fi].w, fb.w, acc[fi], 0, 0, 0);                                      \
            load_u(cn1_, fi);                                                                                                      \
            if (fi + 2 < 8) fa[fi + 2] = read_v(Vb, fi + 2);                                                                       \
            if (WINO_LD_ON((SLOT) ^ 1)) {                                                                                          \
                if (fi == 2) WINO_ROW_PASS((SLOT) ^ 1);                                                                            \
                if (fi >= 3 && fi <= 6) col_store((SLOT) ^ 1, fi - 3);                                                             \
            }                                                                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                                                     \
        }                                                                                                                          \
        __builtin_amdgcn_s_setprio(0);                                                                                             \
        __syncthreads();                                                                                                           \
    } while (0)

    if (WINO_LD_ON(0)) { WINO_LOAD_PATCH(0, c0); }
    if (WINO_LD_ON(1)) { WINO_LOAD_PATCH(1, min(c0 + 1, c1 - 1)); }
#pragma unroll
    for (int fi = 0; fi < 8; ++fi) load_u(c0, fi);
    if (WINO_LD_ON(0)) {
        WINO_ROW_PASS(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) col_store(0, j);
    }
    __syncthreads();
    for (int cc = c0; cc < c1; cc += 2) {
        WINO_ITER(0, cc);
        if (cc + 1 < c1) WINO_ITER(1, cc + 1);
    }
    float* const ybase = a.y + (size_t)blockIdx.y * a.slice_stride;
#undef WINO_ITER
#undef WINO_LD_ON
#undef WINO_ROW_PASS
#undef WINO_LOAD_PATCH

#include "fd_conv_wino_epilogue.inc"
}

// U = G g G^T per (cout, cin), G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1], computed in double and rounded once; packed
// [ceil(Cout / 32)][Cin / 8][16 f][32 cout][8 c] (zero rows past Cout; frequencies 12..15 negated, see the kernels).  mode 1 = the weights of the data-gradient conv
// (N = Cin, K = Cout): g'[ci][co][r][q] = g[co][ci][2 - r][2 - q] * (scale ? scale[co] : 1).
__global__ __launch_bounds__(256) void wino_pack_kernel(const float* __restrict__ w, const float* __restrict__ scale, float* __restrict__ out,
                                                        int N, int K, int mode) {
    const int Np = (N + 31) & ~31;
    const long total = (long)Np * K;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int n = (int)(i / K), k = (int)(i - (long)n * K);
        fd_wino_pack_one(w, scale, out, N, K, mode, n, k);
    }
}

extern "C" int64_t fd_wino_weight_bytes(int32_t Cout, int32_t Cin) {
    if (Cout < 1 || Cin < 8 || Cin % 8) return -1;
    return (int64_t)((Cout + 31) & ~31) * Cin * 16 * 4;
}

extern "C" int32_t fd_wino_pack_weights_f32(const float* w, const float* scale, float* out, int32_t Cout, int32_t Cin, int32_t mode,
                                            fd_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    FD_REQUIRE(w && out && Cout >= 1 && Cin >= 1 && (mode == 0 || mode == 1), FD_E_INVAL, "fd_wino_pack_weights: bad arguments");
    const int N = mode == 0 ? Cout : Cin, K = mode == 0 ? Cin : Cout;
    FD_REQUIRE(K % 8 == 0, FD_E_UNSUPPORTED, "fd_wino_pack_weights: the reduction width (%d) must be a multiple of 8", K);
    const long total = (long)((N + 31) & ~31) * K;
    long g = (total + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(wino_pack_kernel, dim3((unsigned)g), dim3(256), 0, stream, w, scale, out, N, K, mode);
    FD_CHECK_LAUNCH("fd_wino_pack_weights_f32");
    return FD_OK;
}

template <int TAG, int NCH>
static int launch_wino(const WinoArgs& a, hipStream_t stream) {
    constexpr int lds = NCH * 32 * 1024;      // the epilogue's pair stages: 32 KB per 32-channel block (the main loop needs 32 KB)
    auto kern = conv3x3_wino_kernel<TAG, NCH>;
    static std::atomic<unsigned> attr_mask{0};
    fd_set_max_lds_once(attr_mask, reinterpret_cast<const void*>(kern), lds);
    hipLaunchKernelGGL(kern, dim3(8 * a.mt_per * a.ntiles, (a.NC + a.nc_per - 1) / a.nc_per), dim3(NCH * 128), lds, stream, a);
    FD_CHECK_LAUNCH("fd_conv2d_nhwc_f32 (winograd)");
    return FD_OK;
}

// Called by fd_conv2d_nhwc_f32 for tile == FD_TILE_WINOGRAD (p->w = the fd_wino_pack_weights_f32 packing).
int fd_launch_conv_wino(const fd_conv_params* p, hipStream_t stream) {
    FD_REQUIRE(p->mode == FD_CONV_GENERIC && p->KH == 3 && p->KW == 3 && p->stride == 1 && p->pad == p->dil &&
                   (p->dil == 1 || p->dil == 2) && p->out_H <= 0 && p->sc_H <= 0 && p->precision == FD_PREC_F32,
               FD_E_UNSUPPORTED, "fd_conv2d: FD_TILE_WINOGRAD needs an fp32 3x3 stride-1 'same' conv with dilation 1 or 2 (no scatter)");
    FD_REQUIRE(p->Cin % 8 == 0 && p->Cout % 4 == 0, FD_E_UNSUPPORTED, "fd_conv2d: FD_TILE_WINOGRAD needs Cin %% 8 == 0 and Cout %% 4 == 0 (Cin=%d Cout=%d)",
               p->Cin, p->Cout);
    FD_REQUIRE(p->y_cs % 4 == 0 && p->y_co % 4 == 0 && ((uintptr_t)p->y & 15) == 0 &&
                   (!p->res || (p->res_cs % 4 == 0 && p->res_co % 4 == 0 && ((uintptr_t)p->res & 15) == 0)) &&
                   (!p->scale || ((uintptr_t)p->scale & 15) == 0) && (!p->shift || ((uintptr_t)p->shift & 15) == 0),
               FD_E_UNSUPPORTED, "fd_conv2d: FD_TILE_WINOGRAD needs 16-byte addressable output / residual / scale / shift views");
    WinoArgs a;
    a.x = p->x; a.u = p->w; a.scale = p->scale; a.shift = p->shift; a.res = p->res; a.y = p->y;
    a.x_cs = p->x_cs; a.x_co = p->x_co; a.res_cs = p->res_cs; a.res_co = p->res_co; a.y_cs = p->y_cs; a.y_co = p->y_co;
    a.Cin = p->Cin; a.Cout = p->Cout; a.dil = p->dil; a.act = p->act; a.act_c0 = p->act_c0;
    a.res_mask = (p->res && p->res_mode == 1) ? 1 : 0;
    a.gn_stats = nullptr; a.gn_G = 1; a.gn_cg = 4;
    if (p->gn_stats) {
        FD_REQUIRE(p->gn_groups >= 1 && p->Cout % p->gn_groups == 0 && (p->Cout / p->gn_groups) % 4 == 0 && 32 % (p->Cout / p->gn_groups) == 0 &&
                       p->Cout % 32 == 0 && p->ksplit <= 1 && ((uintptr_t)p->gn_stats & 7) == 0,
                   FD_E_UNSUPPORTED, "fd_conv2d: gn_stats needs Cout %% 32 == 0, 4 | Cout / groups | 32, no split-K");
        a.gn_stats = p->gn_stats; a.gn_G = p->gn_groups; a.gn_cg = p->Cout / p->gn_groups;
    }
    a.NC = p->Cin / 8;
    a.nseg = p->in.nseg;
    long t = 0;
    for (int s = 0; s < FD_MAX_SEG; ++s) {
        a.t0[s] = (int)t;
        if (s < p->in.nseg) {
            a.H[s] = p->in.H[s]; a.W[s] = p->in.W[s];
            a.TH[s] = ((p->in.H[s] + p->dil - 1) / p->dil + 1) / 2;
            a.TW[s] = ((p->in.W[s] + p->dil - 1) / p->dil + 1) / 2;
            a.m0[s] = p->in.m_start[s];
            t += (long)p->in.batch * p->dil * p->dil * a.TH[s] * a.TW[s];
        } else {
            a.H[s] = a.W[s] = a.TH[s] = a.TW[s] = 1; a.m0[s] = 0;
        }
        a.seg_param[s] = p->seg_param[s];
    }
    a.t0[FD_MAX_SEG] = (int)t;
    FD_REQUIRE(t > 0 && t < (1L << 30), FD_E_INVAL, "fd_conv2d: tile count out of range");
    a.T = (int)t;
    const long rows = p->in.m_start[p->in.nseg];
    FD_REQUIRE(rows * p->x_cs < (1L << 31) && rows * p->y_cs < (1L << 31), FD_E_UNSUPPORTED, "fd_conv2d: tensor exceeds 2^31 elements");
    const long xb = rows * p->x_cs * 4, ub = (long)((p->Cout + 31) & ~31) * p->Cin * 64;
    FD_REQUIRE(xb < 0xC0000000L - 65536 && ub < 0xC0000000L, FD_E_UNSUPPORTED, "fd_conv2d: input / weight buffer exceeds 3 GiB");
    a.x_bytes = (unsigned)xb; a.u_bytes = (unsigned)ub;
    a.mtiles = (a.T + WINO_TB - 1) / WINO_TB;
    // 128-channel workgroups (8 waves) for wide layers with enough work to go round: measured on MI355X 5-7 % faster on the towers and the
    // trunk's 256- / 512-wide conv2, slower at Cout = 128 and on maps with < 200 workgroups (HisBlock1 conv4: 100)
    const int nch = (p->Cout % 128 == 0 && p->Cout >= 256 && (long)a.mtiles * (p->Cout / 128) >= 192) ? 4 : 2;
    a.ntiles = (p->Cout + 32 * nch - 1) / (32 * nch);
    a.mt_per = (a.mtiles + 7) / 8;
    a.nc_per = a.NC; a.slice_stride = 0;
    const int ksplit = p->ksplit > 1 ? p->ksplit : 1;
    if (ksplit > 1) {
        // split-K: the chunk loop is divided over `ksplit` workgroups per tile, raw partial outputs (the output transform is linear) go to the
        // workspace and the direct kernel's combine launch adds the slabs in slice order and applies the epilogue (deterministic)
        FD_REQUIRE(ksplit <= 64 && a.NC >= 2 * ksplit, FD_E_INVAL, "fd_conv2d: ksplit=%d needs 1 < ksplit <= min(64, Cin / 16 = %d)", ksplit, a.NC / 2);
        const int ldw = (p->Cout + 3) & ~3;
        const long slab = rows * ldw;
        FD_REQUIRE(p->workspace && ((uintptr_t)p->workspace & 15) == 0 && p->workspace_bytes >= (int64_t)ksplit * slab * 4, FD_E_INVAL,
                   "fd_conv2d: split-K needs a 16-byte aligned workspace of fd_conv_workspace_bytes() bytes");
        a.nc_per = (a.NC + ksplit - 1) / ksplit;
        a.y = (float*)p->workspace; a.y_cs = ldw; a.y_co = 0; a.slice_stride = slab;
        a.scale = a.shift = a.res = nullptr; a.act = FD_ACT_NONE; a.res_mask = 0;
        const int rc = (nch == 4) ? launch_wino<0, 4>(a, stream) : launch_wino<0, 2>(a, stream);
        if (rc != FD_OK) return rc;
        ConvArgs o = {};
        o.scale = p->scale; o.shift = p->shift; o.res = p->res; o.y = p->y;
        o.res_cs = p->res_cs; o.res_co = p->res_co; o.y_cs = p->y_cs; o.y_co = p->y_co;
        o.Cout = p->Cout; o.act = p->act; o.act_c0 = p->act_c0; o.M = (int)rows; o.nseg = p->in.nseg;
        o.res_mask = (p->res && p->res_mode == 1) ? 1 : 0;
        for (int sg = 0; sg <= FD_MAX_SEG; ++sg) o.m_out[sg] = p->in.m_start[sg < p->in.nseg ? sg : p->in.nseg];
        for (int sg = 0; sg < FD_MAX_SEG; ++sg) { o.seg_param[sg] = p->seg_param[sg]; o.Ho[sg] = o.Wo[sg] = 1; }
        o.sc_on = 0;
        return fd_launch_splitk_reduce(o, (const float*)p->workspace, (a.NC + a.nc_per - 1) / a.nc_per, ldw, slab, stream);
    }
    if (nch == 4) return p->tag == 1 ? launch_wino<1, 4>(a, stream) : launch_wino<0, 4>(a, stream);
    return p->tag == 1 ? launch_wino<1, 2>(a, stream) : launch_wino<0, 2>(a, stream);
}
