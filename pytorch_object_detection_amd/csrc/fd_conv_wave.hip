// fd_conv_wave.hip -- the GEMM-addressed layers (1x1, stride 1, no padding: bottleneck conv1 / conv3 / downsample, FPN laterals, head
// pointwise convs) as WAVE-AUTONOMOUS tiles on v_mfma_f32_32x32x2_f32 (FD_TILE_WAVE64).
//
// Why another kernel: on these layers the K loop is short (K = 64 .. 2048) and the epilogue moves as many bytes as the loader, and in
// the workgroup-tiled kernel (fd_conv.hip) their times ADD: the four waves of a workgroup meet at one or two barriers per
// K-tile, so they enter the epilogue together, and with equal work per workgroup the co-resident workgroups of a CU -- of the whole chip
// -- run their K loops and their epilogues in lockstep: the matrix pipe idles while everybody stores, HBM idles while everybody multiplies.
// gfx950 has no named barriers to run two half-workgroups out of phase, so this kernel removes the barrier instead:
//   * ONE wave = one workgroup = one 64 x 64 output tile (2 x 2 sub-tiles of 32 x 32, 64 accumulator registers): nothing is shared
//     between waves, nothing synchronises them, and the 3 waves a SIMD hosts drift apart within the first tile -- while one drains its
//     accumulators through LDS to HBM the others keep the matrix pipe busy.  (fp32 MFMAs are 64 cycles each: one wave with 4
//     independent accumulators already saturates a SIMD's pipe, so wave-private tiles cost no matrix throughput.)
//   * A (activations) goes global -> registers -> wave-private LDS in full 128-byte rows (8 lanes per row, XOR-swizzled as in
//     fd_conv.hip), one 8 KiB stage per wave: LDS executes a wave's accesses in order, so the next stage is written right behind the
//     last fragment read of the previous one with no second buffer and no barrier.
//   * B (weights) never touches LDS: fd_pack_conv_weight_wave_f32 stores them in MFMA FRAGMENT order, so a lane fetches its operand
//     with one coalesced 16-byte load per (sub-tile, k-step) straight from L2 (a layer's weights are <= 8 MB and shared by every tile);
//     the fragment of K-tile t+1 lands in the registers the MFMAs of K-tile t have just consumed.
//   * epilogue per 32 x 32 sub-tile: scale / shift, residual add or ReLU mask, ReLU / SiLU, 16-byte stores through the wave's LDS
//     transposition (the contract of fd_conv_epilogue.inc for these layers), the next sub-tile's residual in flight meanwhile.
// Results are bit-identical to the workgroup-tiled kernel's (same k order inside a K-tile, same fma chain per output).
#include "fd_conv_common.h"

#define WV_BM 64
#define WV_BN 64

template <int UNUSED>
__global__ __launch_bounds__(64, 3) void conv1x1_wave_kernel(ConvArgs a, const float* __restrict__ wfrag) {
    constexpr int TM = 2, TN = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);                 // [64 rows][32 k], 8 KiB; reused as the epilogue's 4 KiB stage

    const int lane = threadIdx.x, l31 = lane & 31, lh = lane >> 5;

    // XCD-aware tile order (as fd_conv.hip): each XCD's L2 sees a contiguous range of M-tiles x all N-tiles
    const int nblk = a.mtiles * a.ntiles;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int mt = bid / a.ntiles, nt = bid - mt * a.ntiles;
    const int m0 = mt * WV_BM, n0 = nt * WV_BN;

    constexpr unsigned OOB = 0xC0000000u;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, (short)0, (int)a.x_bytes, 0x00020000);
    const int lrow = lane >> 3, chunk = lane & 7;
    unsigned a_off[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = m0 + lrow + 8 * i;
        a_off[i] = (m < a.M) ? ((unsigned)m * (unsigned)a.x_cs + (unsigned)(a.x_co + chunk * 4)) * 4u : OOB;   // (OOB + any K offset stays out of range: reads 0)
    }
    // B fragments: [n-tile][K-tile][sub-tile j][k-step s][lane] float4
    const float4* __restrict__ wf = reinterpret_cast<const float4*>(wfrag) + (size_t)nt * a.KT * 512 + lane;

    float4 ra[8], fb[TN][4];
    auto load_a = [&](int kt) {
        const unsigned kb = (unsigned)kt * 128u;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            ra[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)(a_off[i] + kb), 0, 0));
    };
    auto store_a = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<float4*>(As + lds_off(lrow + 8 * i, chunk)) = ra[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    float* const ybase = a.y;
    const int KT = a.KT;
    load_a(0);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int s = 0; s < 4; ++s) fb[j][s] = wf[(j * 4 + s) * 64];

    for (int kt = 0; kt < KT - 1; ++kt) {
        store_a();                                  // (waits for the loads of this K-tile, issued one tile of MFMAs ago)
        __builtin_amdgcn_sched_barrier(0);          // (the next tile's loads reuse the registers the LDS writes have just read)
        load_a(kt + 1);
        __builtin_amdgcn_sched_barrier(0);
        wave_lds_sync();
        const float4* __restrict__ wn_ = wf + (size_t)(kt + 1) * 512;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float4 fa[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const float4*>(As + lds_off(i * 32 + l31, 2 * s + lh));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j][s].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j][s].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j][s].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j][s].w, acc[i][j], 0, 0, 0);
                }
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j][s] = wn_[(j * 4 + s) * 64];      // next K-tile's fragment into the registers just consumed
            __builtin_amdgcn_sched_barrier(0);      // (left alone the scheduler hoists all eight fragment loads to the top of the K-tile: 32 more live registers)
        }
        __builtin_amdgcn_s_setprio(0);
    }
    // last K-tile: no operand loads left
    store_a();
    wave_lds_sync();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        float4 fa[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const float4*>(As + lds_off(i * 32 + l31, 2 * s + lh));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j][s].x, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j][s].y, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j][s].z, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j][s].w, acc[i][j], 0, 0, 0);
            }
    }
    __builtin_amdgcn_s_setprio(0);
    wave_lds_sync();                               // every fragment read of the A stage is done: the epilogue reuses it

    // ---- epilogue, one 32 x 32 sub-tile at a time: y = act(acc * scale + shift (+ | mask) res).  acc register e of lane l is
    // C[row (e&3) + 8 (e>>2) + 4 (l>>5)][col l&31]; each sub-tile is transposed through the wave's LDS stage so that a lane owns 4
    // consecutive channels of one pixel (16-byte residual loads / output stores, 8 full 128-byte lines per instruction).  The residual
    // of sub-tile t+1 is requested before sub-tile t is stored: 16 live registers per set instead of the whole tile's 64 (which would
    // spill at three waves per SIMD).
    float* stage = As;
    const int c4 = (lane & 7) * 4, prow = lane >> 3;
    const int act_u = a.act;
    float4 rr[2][4];
    auto load_res = [&](int t, int slot) {
        const int i = t & 1, j = t >> 1;
        const int nb = n0 + 32 * j, mb = m0 + 32 * i;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int m = mb + prow + 8 * p;
            rr[slot][p] = (a.res && m < a.M && nb < a.Cout_epi)
                              ? *reinterpret_cast<const float4*>(a.res + (size_t)m * a.res_cs + a.res_co + nb + c4)
                              : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    load_res(0, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) {           // t = 2 j + i: the two row blocks of one channel block share its scale / shift
        const int i = t & 1, j = t >> 1;
        const int nb = n0 + 32 * j, mb = m0 + 32 * i;
        if (nb >= a.Cout_epi) break;        // (uniform) Cout % 64 == 32: the second channel block does not exist
        const int n = nb + l31;
        const float sc = a.scale ? a.scale[n] : 1.0f, sf = a.shift ? a.shift[n] : 0.0f;
        const bool act_on = act_u != FD_ACT_NONE && a.act_c0 <= nb;     // (uniform; act_c0 % 32 == 0 checked by the host)
#pragma unroll
        for (int e = 0; e < 16; ++e) stage[((e & 3) + 8 * (e >> 2) + 4 * lh) * 32 + l31] = acc[i][j][e] * sc + sf;
        if (t + 1 < 4) load_res(t + 1, (t + 1) & 1);
        wave_lds_sync();
        float* yp = ybase + (size_t)(mb + prow) * a.y_cs + a.y_co + nb + c4;
        const size_t ystep = (size_t)8 * a.y_cs;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float4 v = *reinterpret_cast<const float4*>(stage + (prow + 8 * p) * 32 + c4);
            if (a.res) {
                const float4 r = rr[t & 1][p];
                if (a.res_mask) {
                    v.x = r.x > 0.f ? v.x : 0.f; v.y = r.y > 0.f ? v.y : 0.f; v.z = r.z > 0.f ? v.z : 0.f; v.w = r.w > 0.f ? v.w : 0.f;
                } else {
                    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                }
            }
            if (act_on) {
                if (act_u == FD_ACT_RELU) {
                    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                } else {
                    v.x = fd_act(v.x, FD_ACT_SILU, 0.f); v.y = fd_act(v.y, FD_ACT_SILU, 0.f);
                    v.z = fd_act(v.z, FD_ACT_SILU, 0.f); v.w = fd_act(v.w, FD_ACT_SILU, 0.f);
                }
            }
            const bool st_ok = mb + prow + 8 * p < a.M;
            if (st_ok) *reinterpret_cast<float4*>(yp + p * ystep) = v;
            if (a.gn_stats) fd_gn_rowstats(a.gn_stats, a.gn_G, a.gn_cg, v, (size_t)(mb + prow + 8 * p), nb + c4, lane, st_ok);
        }
        wave_lds_sync();
    }
}

int fd_launch_conv_wave(const ConvArgs& a, const float* wfrag, hipStream_t stream) {
    ConvArgs b = a;
    b.mtiles = (a.M + WV_BM - 1) / WV_BM;
    b.ntiles = (a.Cout + WV_BN - 1) / WV_BN;
    hipLaunchKernelGGL(conv1x1_wave_kernel<0>, dim3((unsigned)(b.mtiles * b.ntiles)), dim3(64), 8192, stream, b, wfrag);
    FD_CHECK_LAUNCH("fd_conv2d_nhwc_f32 (wave tiles)");
    return FD_OK;
}

// [Cout][Cin] (1x1 weights, Cin % 32 == 0) -> MFMA fragment order [ceil(Cout/64)][Cin/32][2 sub-tiles][4 k-steps][64 lanes][4]:
// lane (l31, lh) of sub-tile j, k-step s of K-tile kt holds w[64 nt + 32 j + l31][32 kt + 8 s + 4 lh + 0..3]; rows past Cout are zero.
__global__ __launch_bounds__(256) void pack_wave_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin, long total4) {
    const int KT = Cin >> 5;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const int lane = (int)(i & 63);
        long t = i >> 6;
        const int s = (int)(t & 3); t >>= 2;
        const int j = (int)(t & 1); t >>= 1;
        const int kt = (int)(t % KT);
        const int nt = (int)(t / KT);
        const int n = nt * 64 + j * 32 + (lane & 31), k = kt * 32 + 8 * s + 4 * (lane >> 5);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < Cout) v = *reinterpret_cast<const float4*>(w + (long)n * Cin + k);
        reinterpret_cast<float4*>(out)[i] = v;
    }
}

extern "C" int64_t fd_conv_weight_wave_bytes(int32_t Cout, int32_t Cin) {
    if (Cout < 1 || Cin < 32 || Cin % 32) return -1;
    return (int64_t)((Cout + 63) / 64) * 64 * Cin * 4;
}

extern "C" int32_t fd_pack_conv_weight_wave_f32(const float* w, float* out, int32_t Cout, int32_t Cin, fd_stream_t stream) {
    FD_REQUIRE(w && out && Cout >= 1 && Cin >= 32 && Cin % 32 == 0, FD_E_INVAL, "fd_pack_conv_weight_wave: need Cin %% 32 == 0 (Cout=%d Cin=%d)", Cout, Cin);
    FD_REQUIRE((((uintptr_t)w | (uintptr_t)out) & 15) == 0, FD_E_INVAL, "fd_pack_conv_weight_wave: pointers not 16-byte aligned");
    const long total4 = (long)((Cout + 63) / 64) * 64 * Cin / 4;
    long g = (total4 + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(pack_wave_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w, out, Cout, Cin, total4);
    FD_CHECK_LAUNCH("fd_pack_conv_weight_wave_f32");
    return FD_OK;
}
