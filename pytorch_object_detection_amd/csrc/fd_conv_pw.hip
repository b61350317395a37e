// fd_conv_pw.hip -- persistent variant of the implicit-GEMM kernel for the GEMM-addressed layers (1x1, stride 1, no padding:
// every bottleneck conv1 / conv3 / downsample, FPN laterals, MBConv expand / project, the head's pointwise convs).
//
// Why: with K = 64 .. 512 a workgroup of conv_igemm_kernel lives for 2 - 16 K-tiles, and what it cannot hide is its own start
// (kernel arguments, address set-up, the first operand tile's HBM latency) and its end (residual tile, stores): measured on
// 128 -> 512 @ 80x80 + residual, time(K) = K-loop at the matrix pipe's peak rate + a constant 77 us, i.e. the pipe idles while
// tiles turn over, whatever the number of co-resident workgroups (fd_conv.hip keeps 3 - 4 per CU).
// Here a workgroup stays on its CU and walks tiles t, t + G, t + 2G, ...; the software pipeline is rotated across the tile
// boundary: the next tile's first K-tile and this tile's residual are requested BEFORE the last K-tile's MFMAs, so both
// latencies are covered by matrix work and by the epilogue's stores, and the only per-tile serial cost left is the epilogue's
// LDS transposition.  Same LDS layout, weight packing ([Cout][Cin/32][32]), MFMA loop and epilogue as fd_conv.hip.
#include "fd_conv_common.h"

template <int WGM, int WGN, int TM, int TN, bool SB>
__global__ __launch_bounds__(WGM * WGN * 64, (TM * TN == 4) ? 2 : 3)
void conv1x1_persist_kernel(ConvArgs a) {
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    constexpr int NT = WGM * WGN * 64;
    constexpr int RPP = NT / 8;
    constexpr int AP = BM / RPP, BP = BN / RPP;
    constexpr bool SPLIT = false;                // (the shared epilogue's split-f16 branch is compiled out)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NBUF = SB ? 1 : 2;
    float* As = reinterpret_cast<float*>(smem);  // [NBUF][BM*32]
    float* Bs = As + NBUF * BM * 32;             // [NBUF][BN*32]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int l31 = lane & 31, lh = lane >> 5;
    const int lrow = tid >> 3, chunk = tid & 7;
    constexpr unsigned OOB = 0xC0000000u;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, (short)0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, (short)0, (int)a.w_bytes, 0x00020000);

    const int nblk = a.mtiles * a.ntiles;
    // tile t -> (m0, n0): the grid is a multiple of 8, so workgroup g and all its tiles g + i * G sit on XCD g & 7; each XCD walks
    // a contiguous range of tiles (n-tiles of one m-tile adjacent: the A rows are re-read from that XCD's L2)
    auto coords = [&](int t, int& m0_, int& n0_) {
        const int q = nblk >> 3, r = nblk & 7, xcd = t & 7, idx = t >> 3;
        const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        const int mt = bid / a.ntiles, nt = bid - mt * a.ntiles;
        m0_ = mt * BM; n0_ = nt * BN;
    };
    unsigned a_off[AP], b_off[BP];       // loader state of the tile being FETCHED (one ahead of the tile being computed at a tile boundary)
    auto setup = [&](int m0_, int n0_) {
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int m = m0_ + lrow + RPP * i;
            a_off[i] = (m < a.M) ? ((unsigned)m * (unsigned)a.x_cs + (unsigned)(a.x_co + chunk * 4)) * 4u : OOB;
        }
#pragma unroll
        for (int j = 0; j < BP; ++j) {
            const int n = n0_ + lrow + RPP * j;
            b_off[j] = (n < a.Cout) ? ((unsigned)n * (unsigned)a.Kpacked + (unsigned)(chunk * 4)) * 4u : OOB;
        }
    };
    float4 ra[AP], rb[BP];
    auto load_tile = [&](int kt) {
        const bool c_ok = kt * 32 + chunk * 4 < a.Cin;     // Cin % 32 != 0: the last chunk's missing channels read as zero
        const unsigned kb = (unsigned)kt * 128u;
#pragma unroll
        for (int i = 0; i < AP; ++i)
            ra[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)((c_ok && a_off[i] != OOB) ? a_off[i] + kb : OOB), 0, 0));
#pragma unroll
        for (int j = 0; j < BP; ++j)
            rb[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (int)(b_off[j] != OOB ? b_off[j] + kb : OOB), 0, 0));
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < AP; ++i) *reinterpret_cast<float4*>(As + buf * BM * 32 + lds_off(lrow + RPP * i, chunk)) = ra[i];
#pragma unroll
        for (int j = 0; j < BP; ++j) *reinterpret_cast<float4*>(Bs + buf * BN * 32 + lds_off(lrow + RPP * j, chunk)) = rb[j];
    };
    f32x16 acc[TM][TN];
    f32x16 cor[1][1];
    auto mfma_tile = [&](int buf) {
        const float* Ab = As + buf * BM * 32 + (wm * TM * 32) * 32;
        const float* Bb = Bs + buf * BN * 32 + (wn * TN * 32) * 32;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const float4*>(Ab + lds_off(i * 32 + l31, 2 * s + lh));
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
    };

    float* const ybase = a.y;
    const int G = gridDim.x;
    int t = blockIdx.x;                  // the launcher keeps G <= nblk: every workgroup owns at least one tile
    int m0, n0;
    coords(t, m0, n0);
    setup(m0, n0);
    load_tile(0);
    for (;;) {      // every wave of the workgroup takes the same path through this loop (t, G, nblk are uniform): all barriers are reached
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        store_tile(0);
        __syncthreads();
        for (int kt = 0; kt < a.KT - 1; ++kt) {
            const int buf = SB ? 0 : (kt & 1);
            load_tile(kt + 1);
            mfma_tile(buf);
            if (SB) __syncthreads();
            store_tile(SB ? 0 : (buf ^ 1));
            __syncthreads();
        }
        // last K-tile of this tile: under its MFMAs fly this tile's residual and the NEXT tile's first K-tile
#include "fd_conv_res_prefetch.inc"
        const int tn = t + G;
        const bool more = tn < nblk;
        int m0n = 0, n0n = 0;
        if (more) {
            coords(tn, m0n, n0n);
            setup(m0n, n0n);
            load_tile(0);
        }
        mfma_tile(SB ? 0 : ((a.KT - 1) & 1));
        __syncthreads();                 // the A / B buffers are free: the epilogue's per-wave stage overlays them
#define FD_EPI_RES_PREFETCHED
    constexpr bool GNS = false;
#include "fd_conv_epilogue.inc"
#undef FD_EPI_RES_PREFETCHED
        if (!more) break;
        __syncthreads();                 // every wave is done with its stage before the next tile's operands land on it
        t = tn; m0 = m0n; n0 = n0n;
    }
}

template <int WGM, int WGN, int TM, int TN, bool SB>
static int launch_pw(const ConvArgs& a, hipStream_t stream) {
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    constexpr int lds_ab = (SB ? 1 : 2) * (BM + BN) * 32 * 4;
    constexpr int NT = WGM * WGN * 64;
    constexpr int lds_epi = (NT / 64) * 4096;
    constexpr int lds = lds_ab > lds_epi ? lds_ab : lds_epi;
    ConvArgs b = a;
    b.mtiles = (a.M + BM - 1) / BM;
    b.ntiles = (a.Cout + BN - 1) / BN;
    auto kern = conv1x1_persist_kernel<WGM, WGN, TM, TN, SB>;
    static std::atomic<unsigned> attr_mask{0};
    fd_set_max_lds_once(attr_mask, reinterpret_cast<const void*>(kern), lds);
    // resident workgroups of this instantiation on this device: CUs x occupancy (queried once per device)
    static std::atomic<int> slots[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    int g = slots[dev].load(std::memory_order_relaxed);
    if (g == 0) {
        int occ = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, NT, lds) != hipSuccess || occ < 1) occ = 1;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        g = ((occ * cus) + 7) & ~7;
        slots[dev].store(g, std::memory_order_relaxed);
    }
    const int nblk = b.mtiles * b.ntiles;
    if (g > nblk) g = nblk;
    hipLaunchKernelGGL(kern, dim3(g), dim3(NT), lds, stream, b);
    FD_CHECK_LAUNCH("fd_conv2d_nhwc_f32 (persistent 1x1)");
    return FD_OK;
}

// tile ids as in fd_conv2d_nhwc_f32; returns FD_E_UNSUPPORTED for tiles without a persistent instantiation
int fd_launch_conv_pw(const ConvArgs& a, int tile, hipStream_t stream) {
    switch (tile) {
        case FD_TILE_128x128: return launch_pw<2, 2, 2, 2, false>(a, stream);
        case FD_TILE_128x128_SB: return launch_pw<2, 2, 2, 2, true>(a, stream);
        case FD_TILE_128x64: return launch_pw<2, 2, 2, 1, false>(a, stream);
        case FD_TILE_128x64_SB: return launch_pw<2, 2, 2, 1, true>(a, stream);
        case FD_TILE_64x128: return launch_pw<2, 2, 1, 2, false>(a, stream);
        case FD_TILE_64x128_SB: return launch_pw<2, 2, 1, 2, true>(a, stream);
        case FD_TILE_64x64: return launch_pw<2, 2, 1, 1, false>(a, stream);
        default: return FD_E_UNSUPPORTED;
    }
}
