// fd_stem.hip — the ResNet stem: 7x7 stride-2 pad-3 convolution 3 -> 64 + frozen BatchNorm + ReLU (torchvision resnet50.conv1 / bn1 /
// relu behind the reference's model/backbone/resnet50.py:68-80) on the [N][H][W][4] image layout, as its own fp32-MFMA kernel.
//
// Through the generic implicit-GEMM kernel (FD_CONV_STEM) the stem's K = 7 * 7 * 3 = 147 is padded to 7 filter rows x 8 pixels x 4
// channels = 224 (the zero 4th channel and the zero 8th pixel cost 34 % of the MFMAs) and every output pixel's window is fetched from
// global memory again: 62 TFLOP/s, MFMA-bound on the padding.  Here a workgroup owns 8 x 32 output pixels x all 64 channels:
//   * its 21 x 69-pixel input patch is staged ONCE in LDS with the zero channel dropped ([21][69 * 3 + 1] floats, 17 KB) and the whole
//     filter bank ([7 rows][22 k][64 cout], 39 KB: 21 = 7 pixels x 3 channels per filter row + one zero k so that k pairs never
//     straddle a row) beside it;
//   * K = 7 x 22 = 154 (5 % padding): per filter row 11 MFMA steps of K = 2, the A operand read as single floats at
//     patch[2 * row + r][2 * col * 3 + j] (lane = output column: stride 6 floats, 2-way bank conflict at worst), the B operand 32
//     consecutive output channels per half wave (conflict-free);
//   * wave w = output rows 2w, 2w + 1 of the tile x 64 channels: 4 accumulators of 32 x 32; the operands of step s + 1 are read from LDS
//     before the MFMAs of step s are issued (pinned with sched_barriers);
//   * a workgroup walks 4 consecutive tiles (filter bank staged once), the next tile's patch travels global -> registers under the
//     current tile's MFMAs;
//   * epilogue: BN scale / shift + ReLU, each 32 x 32 block transposed through a per-wave LDS stage into 16-byte NHWC stores.
// Measured (MI355X, 16 x 640 x 640): 0.50 -> 0.35 ms (62 -> 87 TFLOP/s in the stem's 30.8 algorithmic GFLOP).
//
// POOL = true (round 4, fd_stem7x7_pool_nhwc4): the 3x3 stride-2 pad-1 MAX-POOL that follows the stem (torchvision resnet50.maxpool) in the epilogue.
// As two launches the 64-channel 320 x 320 map (420 MB per 16 images) is written by the stem and read back by the pool; here a workgroup walks its
// 4 tiles DOWN the image, keeps a tile's outputs in LDS (32 channels at a time, in the patch / transposition area), and emits pooled pixels:
//   * a window whose 3 x 3 conv outputs all lie in this workgroup's tiles -- the row above a tile comes from a carry of the previous tile's last
//     row kept in LDS -- is stored once (16-byte stores);
//   * a window that straddles a workgroup boundary (the first row of the strip, the left column, and the contributions of the strip's last row /
//     the tile's last column to the neighbours' windows) is combined with the neighbours' parts by integer atomicMax on the float bits: the values are
//     ReLU outputs (>= +0), for which the integer order IS the float order and 0 is the identity -- a small launch zero-fills exactly those pixels first;
//     max is order-independent: bitwise reproducible.  12 % of the pooled pixels take that path (every workgroup-strip boundary row, every 16th column).
#include "fd_conv_common.h"

#define ST_TH 8
#define ST_TW 32
#define ST_PR (2 * ST_TH + 5)        // 21 patch rows
#define ST_PC (2 * ST_TW + 5)        // 69 patch pixels per row
#define ST_PITCH 208                 // floats per patch row: 69 * 3 + 1 (the zero-weight k of the last pixel reads index 207)
#define ST_KR 22                     // k per filter row: 7 pixels x 3 channels + 1 zero
#define ST_CO 64
#define ST_TPW 4                     // consecutive output tiles per workgroup (the 39 KB filter bank is staged once)

struct StemArgs {
    const float4* x; const float* xp;      // xp (NCHW instantiations): the reference's own [N][3][H][W] planes
    const float* w; const float* scale; const float* shift; float* y;
    int y_cs, y_co, N, H, W, Ho, Wo, act, tiles_h, tiles_w;
};

template <bool POOL, bool NCHW>
__global__ __launch_bounds__(256, 2) void stem7x7_kernel(StemArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Ws = reinterpret_cast<float*>(smem);                   // [7 * 22][64]
    float* Ps = Ws + 7 * ST_KR * ST_CO;                           // [21][208]
    float* St = Ps + ST_PR * ST_PITCH;                            // per-wave transpose stages (4 x 4 KB)
    float* Cb = Ps;                                               // POOL: [8 rows][32 cols][32 channels] of this tile's outputs (patch + stages: 8 464 >= 8 192 floats)
    float* Cy = St + 4 * 1024;                                    // POOL: carry = the previous tile's last output row, [32 cols][64 channels]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    // ---- the filter bank once per workgroup; ST_TPW consecutive output tiles per workgroup amortise it ----
    for (int i = tid; i < 7 * ST_KR * ST_CO / 4; i += 256)
        reinterpret_cast<float4*>(Ws)[i] = reinterpret_cast<const float4*>(a.w)[i];
    if (tid < ST_PR) Ps[tid * ST_PITCH + ST_PITCH - 1] = 0.f;

    constexpr int NPX = (ST_PR * ST_PC + 255) / 256;          // patch pixels per thread (6)
    float4 pv[NPX];
    const int ntile = a.N * a.tiles_h * a.tiles_w;
    auto tile_pos = [&](int t, int& n, int& ho0, int& wo0) {
        if constexpr (POOL) {   // consecutive tiles walk DOWN a 32-column strip (a workgroup's ST_TPW tiles share the carry row); strips of ST_TPW tiles never cross an image
            const int th = t % a.tiles_h; t /= a.tiles_h;
            const int tw = t % a.tiles_w;
            n = t / a.tiles_w; ho0 = th * ST_TH; wo0 = tw * ST_TW;
        } else {
            const int tw = t % a.tiles_w; t /= a.tiles_w;
            const int th = t % a.tiles_h;
            n = t / a.tiles_h; ho0 = th * ST_TH; wo0 = tw * ST_TW;
        }
    };
    auto load_patch = [&](int t) {            // global -> registers (zero outside the image)
        int n, ho0, wo0;
        tile_pos(t, n, ho0, wo0);
        const size_t HW = (size_t)a.H * a.W;
        const float4* xin = a.x + (size_t)n * HW;
        const float* xpl = a.xp + (size_t)n * 3 * HW;
#pragma unroll
        for (int u = 0; u < NPX; ++u) {
            const int i = tid + 256 * u;
            const int pr = i / ST_PC, pc = i - pr * ST_PC;
            const int hi = 2 * ho0 - 3 + pr, wi = 2 * wo0 - 3 + pc;
            pv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < ST_PR * ST_PC && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W) {
                if constexpr (NCHW) {       // three plane reads, each coalesced along the patch row (consecutive threads = consecutive pixels)
                    const float* q = xpl + (size_t)hi * a.W + wi;
                    pv[u].x = q[0]; pv[u].y = q[HW]; pv[u].z = q[2 * HW];
                } else {
                    pv[u] = xin[(size_t)hi * a.W + wi];
                }
            }
        }
    };
    auto store_patch = [&]() {                // registers -> LDS, channel 3 dropped
#pragma unroll
        for (int u = 0; u < NPX; ++u) {
            const int i = tid + 256 * u;
            if (i < ST_PR * ST_PC) {
                const int pr = i / ST_PC, pc = i - pr * ST_PC;
                float* d = Ps + pr * ST_PITCH + pc * 3;
                d[0] = pv[u].x; d[1] = pv[u].y; d[2] = pv[u].z;
            }
        }
    };

    const int t0 = blockIdx.x * ST_TPW, t1 = min(ntile, t0 + ST_TPW);
    load_patch(t0);
    store_patch();
    __syncthreads();
    // lane = output column l31 of output rows 2 * wave (+ 1); lane half lh carries k + 1
    const float* A0 = Ps + (2 * (2 * wave)) * ST_PITCH + (2 * l31) * 3 + lh;
    const float* A1 = A0 + 2 * ST_PITCH;
    const float* B0 = Ws + lh * ST_CO + l31;
    float* stage = St + wave * 1024;
    for (int t = t0; t < t1; ++t) {
        int n, ho0, wo0;
        tile_pos(t, n, ho0, wo0);
        if (t + 1 < t1) load_patch(t + 1);     // the next tile's patch travels under this tile's MFMAs
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        // 77 steps of K = 2 (filter row r = s / 11, k pair jj = s % 11); the operands of step s + 1 are read from LDS before the MFMAs of step s
        // are issued (left alone the compiler reads them right before use and waits out the LDS latency every four MFMAs)
        constexpr int NSTEP = 7 * (ST_KR / 2);
        float a0 = A0[0], a1 = A1[0], b0 = B0[0], b1 = B0[32];
#pragma unroll
        for (int s_ = 0; s_ < NSTEP; ++s_) {
            float a0n = 0.f, a1n = 0.f, b0n = 0.f, b1n = 0.f;
            if (s_ + 1 < NSTEP) {
                const int r = (s_ + 1) / (ST_KR / 2), jj = (s_ + 1) % (ST_KR / 2);
                a0n = A0[r * ST_PITCH + 2 * jj]; a1n = A1[r * ST_PITCH + 2 * jj];
                b0n = B0[(r * ST_KR + 2 * jj) * ST_CO]; b1n = B0[(r * ST_KR + 2 * jj) * ST_CO + 32];
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            a0 = a0n; a1 = a1n; b0 = b0n; b1 = b1n;
        }
        __syncthreads();                    // everyone is done reading the patch
        if constexpr (POOL) {
            const bool first = (t == t0), last = (t + 1 == t1);
            const int Hp = (a.Ho - 1) / 2 + 1, Wp = (a.Wo - 1) / 2 + 1;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int nn0 = j * 32;
                const float sc = a.scale ? a.scale[nn0 + l31] : 1.0f;
                const float sf = a.shift ? a.shift[nn0 + l31] : 0.0f;
                // this half's outputs -> Cb[row][col][32 ch]; positions outside the image hold 0 (the identity of max over ReLU outputs)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = 2 * wave + i;
                    const bool rok = ho0 + row < a.Ho;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int col = (e & 3) + 8 * (e >> 2) + 4 * lh;
                        const float v = acc[i][j][e] * sc + sf;
                        Cb[(row * 32 + col) * 32 + l31] = (rok && wo0 + col < a.Wo && v > 0.f) ? v : 0.f;
                    }
                }
                __syncthreads();
                // pooled pixels: centre rows cr = 0, 2, 4, 6 (+ 8: the strip's last row seen from the window below), centre columns cc = 0 .. 30 (+ 32)
                for (int it = tid; it < 5 * 17 * 8; it += 256) {
                    const int c4 = it & 7, pc = (it >> 3) % 17, pr = (it >> 3) / 17;
                    const int cr = 2 * pr, cc = 2 * pc;
                    if (pr == 4 && !last) continue;                       // (the next tile of this workgroup takes row 7 from the carry)
                    const int ph = (ho0 + cr) >> 1, pw = (wo0 + cc) >> 1;
                    if (ph >= Hp || pw >= Wp) continue;
                    float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int dr = -1; dr <= 1; ++dr) {
                        const int r = cr + dr;
                        if (r > 7) continue;
                        if (r < 0 && first) continue;                      // (the row above the strip belongs to another workgroup)
#pragma unroll
                        for (int dc = -1; dc <= 1; ++dc) {
                            const int c = cc + dc;
                            if (c < 0 || c > 31) continue;
                            const float4 v = r < 0 ? *reinterpret_cast<const float4*>(Cy + c * 64 + nn0 + c4 * 4)
                                                   : *reinterpret_cast<const float4*>(Cb + (r * 32 + c) * 32 + c4 * 4);
                            m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
                        }
                    }
                    const bool whole = (pr < 4) && (cr > 0 || !first || ho0 == 0) && (pc < 16) && (cc > 0 || wo0 == 0);
                    float* yp = a.y + ((size_t)(n * Hp + ph) * Wp + pw) * a.y_cs + a.y_co + nn0 + c4 * 4;
                    if (whole) {
                        *reinterpret_cast<float4*>(yp) = m;
                    } else {      // a window shared with a neighbouring workgroup: integer max on the bits of non-negative floats
                        unsigned* up = reinterpret_cast<unsigned*>(yp);
                        atomicMax(up + 0, __float_as_uint(m.x)); atomicMax(up + 1, __float_as_uint(m.y));
                        atomicMax(up + 2, __float_as_uint(m.z)); atomicMax(up + 3, __float_as_uint(m.w));
                    }
                }
                // carry: this tile's last row, for the first window row of the next tile (this half's 32 channels) -- picked up into a register beside the pooling
                // reads and written behind the barrier that ends them (the old carry is still being read until then; one barrier less per half)
                const float4 cyv = *reinterpret_cast<const float4*>(Cb + (7 * 32 + (tid >> 3)) * 32 + (tid & 7) * 4);
                __syncthreads();
                if (!last) *reinterpret_cast<float4*>(Cy + (tid >> 3) * 64 + nn0 + (tid & 7) * 4) = cyv;
            }
            __syncthreads();                // (the carry row is complete before the next tile reads it; the tile buffer is free for the next patch)
            if (tid < ST_PR) Ps[tid * ST_PITCH + ST_PITCH - 1] = 0.f;      // (the tile buffer overwrote the patch rows' zero pad)
            if (t + 1 < t1) store_patch();
        } else {
        if (t + 1 < t1) store_patch();
        // ---- epilogue: acc reg e of lane l is C[pixel (e & 3) + 8 (e >> 2) + 4 lh][cout l31]; 32 x 32 block -> LDS -> float4 rows ----
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int nn0 = j * 32;
            const float sc = a.scale ? a.scale[nn0 + l31] : 1.0f;
            const float sf = a.shift ? a.shift[nn0 + l31] : 0.0f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ho = ho0 + 2 * wave + i;
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    stage[((e & 3) + 8 * (e >> 2) + 4 * lh) * 32 + l31] = fd_act1(acc[i][j][e] * sc + sf, a.act, 0.f);
                wave_lds_sync();
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int row = (lane >> 3) + 8 * p, c4 = (lane & 7) * 4;
                    const int wo = wo0 + row;
                    if (ho < a.Ho && wo < a.Wo) {
                        const float4 v = *reinterpret_cast<const float4*>(stage + row * 32 + c4);
                        *reinterpret_cast<float4*>(a.y + ((size_t)(n * a.Ho + ho) * a.Wo + wo) * a.y_cs + a.y_co + nn0 + c4) = v;
                    }
                }
                wave_lds_sync();
            }
        }
        }
        __syncthreads();                    // the next tile's patch is complete
    }
}

// POOL: the pooled pixels that more than one workgroup contributes to -- the first pooled row of every strip of ST_TPW tiles and the first pooled column of every tile
// column (except the image's own first row / column) -- are combined by atomicMax and therefore start from 0, written here.  The launch enumerates THOSE pixels only
// (per image nrows * Wp + Hp * ncols of them, 11 % of the map; round 4's version walked every pooled pixel to find them: 55 us for 12 MB of stores)
__global__ __launch_bounds__(256) void stem_pool_zero_kernel(float* __restrict__ y, int y_cs, int y_co, int N, int Hp, int Wp) {
    constexpr int RS = ST_TH * ST_TPW / 2, CS = ST_TW / 2;        // pooled rows per workgroup strip, pooled columns per tile column
    const int nrows = (Hp - 1) / RS, ncols = (Wp - 1) / CS;       // boundary rows ph = RS, 2 RS, ..; boundary columns pw = CS, 2 CS, ..
    const int per_img = nrows * Wp + Hp * ncols;
    const long total = (long)N * per_img * 16;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c4 = (int)(i & 15);
        const long q = i >> 4;
        const int n = (int)(q / per_img);
        int r = (int)(q - (long)n * per_img), ph, pw;
        if (r < nrows * Wp) { ph = RS * (1 + r / Wp); pw = r % Wp; }
        else { r -= nrows * Wp; ph = r / ncols; pw = CS * (1 + r % ncols); }
        *reinterpret_cast<float4*>(y + (((long)n * Hp + ph) * Wp + pw) * y_cs + y_co + c4 * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

/* w packed [7 filter rows][22][64 cout]: k = 3 * q + c for filter column q and input channel c, k = 21 zero (ops.pack_stem7_weight). */
static int stem_launch(bool pool, bool nchw, const float* x4, const float* w, const float* scale, const float* shift, float* y, int32_t y_cs, int32_t y_co, int32_t N,
                       int32_t H, int32_t W, int32_t act, hipStream_t stream) {
    FD_REQUIRE(x4 && w && y && N >= 1 && H >= 2 && W >= 2, FD_E_INVAL, "fd_stem7x7: bad arguments");
    FD_REQUIRE(((((nchw ? (uintptr_t)0 : (uintptr_t)x4)) | (uintptr_t)w | (uintptr_t)y) & 15) == 0 && ((uintptr_t)x4 & 3) == 0 && y_cs % 4 == 0 && y_co % 4 == 0 &&
                   y_cs >= y_co + ST_CO, FD_E_INVAL, "fd_stem7x7: pointers must be 16-byte aligned (the NCHW planes: 4-byte) and the 64-channel output view 4-aligned");
    FD_REQUIRE(act == FD_ACT_NONE || act == FD_ACT_RELU || act == FD_ACT_SILU, FD_E_INVAL, "fd_stem7x7: activation none / ReLU / SiLU");
    StemArgs a;
    a.x = reinterpret_cast<const float4*>(x4); a.xp = x4; a.w = w; a.scale = scale; a.shift = shift; a.y = y;
    a.y_cs = y_cs; a.y_co = y_co; a.N = N; a.H = H; a.W = W; a.act = act;
    a.Ho = (H + 6 - 7) / 2 + 1; a.Wo = (W + 6 - 7) / 2 + 1;
    a.tiles_h = (a.Ho + ST_TH - 1) / ST_TH; a.tiles_w = (a.Wo + ST_TW - 1) / ST_TW;
    const long out_rows = pool ? (long)N * ((a.Ho - 1) / 2 + 1) * ((a.Wo - 1) / 2 + 1) : (long)N * a.Ho * a.Wo;
    FD_REQUIRE(out_rows * y_cs < (1L << 31), FD_E_UNSUPPORTED, "fd_stem7x7: tensor exceeds 2^31 elements");
    if (pool) {
        // a workgroup's ST_TPW tiles must lie in ONE column strip of ONE image (the carry row): the tile rows are padded to a multiple of ST_TPW per strip
        // (tiles past the image compute nothing that is stored)
        a.tiles_h = (a.tiles_h + ST_TPW - 1) / ST_TPW * ST_TPW;
        const long blocks = (long)N * a.tiles_w * (a.tiles_h / ST_TPW);
        FD_REQUIRE(blocks < (1L << 31), FD_E_UNSUPPORTED, "fd_stem7x7: too many tiles");
        constexpr int lds = (7 * ST_KR * ST_CO + ST_PR * ST_PITCH + 4 * 1024 + 32 * 64) * 4;   // 81 KB: filter bank + patch + stages (= the tile buffer) + carry row
        static std::atomic<unsigned> attr_mask{0}, attr_mask_p{0};
        if (nchw) fd_set_max_lds_once(attr_mask_p, reinterpret_cast<const void*>(stem7x7_kernel<true, true>), lds);
        else fd_set_max_lds_once(attr_mask, reinterpret_cast<const void*>(stem7x7_kernel<true, false>), lds);
        {
            const int Hp = (a.Ho - 1) / 2 + 1, Wp = (a.Wo - 1) / 2 + 1;
            const long items = (long)N * (((Hp - 1) / (ST_TH * ST_TPW / 2)) * Wp + Hp * ((Wp - 1) / (ST_TW / 2))) * 16;
            long g = (items + 255) / 256;
            if (g > 8192) g = 8192;
            if (g > 0) hipLaunchKernelGGL(stem_pool_zero_kernel, dim3((unsigned)g), dim3(256), 0, stream, y, y_cs, y_co, N, Hp, Wp);
            FD_CHECK_LAUNCH("fd_stem7x7_pool_nhwc4 (zero)");
        }
        if (nchw) hipLaunchKernelGGL((stem7x7_kernel<true, true>), dim3((unsigned)blocks), dim3(256), lds, stream, a);
        else hipLaunchKernelGGL((stem7x7_kernel<true, false>), dim3((unsigned)blocks), dim3(256), lds, stream, a);
        FD_CHECK_LAUNCH("fd_stem7x7 (+ max-pool)");
        return FD_OK;
    }
    const long blocks = ((long)N * a.tiles_h * a.tiles_w + ST_TPW - 1) / ST_TPW;
    FD_REQUIRE(blocks < (1L << 31), FD_E_UNSUPPORTED, "fd_stem7x7: too many tiles");
    constexpr int lds = (ST_PR * ST_PITCH + 7 * ST_KR * ST_CO + 4 * 1024) * 4;       // 73 KB: patch + filter bank + transpose stages
    static std::atomic<unsigned> attr_mask{0}, attr_mask_p{0};
    if (nchw) {
        fd_set_max_lds_once(attr_mask_p, reinterpret_cast<const void*>(stem7x7_kernel<false, true>), lds);
        hipLaunchKernelGGL((stem7x7_kernel<false, true>), dim3((unsigned)blocks), dim3(256), lds, stream, a);
    } else {
        fd_set_max_lds_once(attr_mask, reinterpret_cast<const void*>(stem7x7_kernel<false, false>), lds);
        hipLaunchKernelGGL((stem7x7_kernel<false, false>), dim3((unsigned)blocks), dim3(256), lds, stream, a);
    }
    FD_CHECK_LAUNCH("fd_stem7x7");
    return FD_OK;
}

extern "C" int32_t fd_stem7x7_nhwc4(const float* x4, const float* w, const float* scale, const float* shift, float* y, int32_t y_cs,
                                    int32_t y_co, int32_t N, int32_t H, int32_t W, int32_t act, fd_stream_t stream_) {
    return stem_launch(false, false, x4, w, scale, shift, y, y_cs, y_co, N, H, W, act, (hipStream_t)stream_);
}

extern "C" int32_t fd_stem7x7_pool_nhwc4(const float* x4, const float* w, const float* scale, const float* shift, float* y_pooled, int32_t y_cs,
                                         int32_t y_co, int32_t N, int32_t H, int32_t W, fd_stream_t stream_) {
    return stem_launch(true, false, x4, w, scale, shift, y_pooled, y_cs, y_co, N, H, W, FD_ACT_RELU, (hipStream_t)stream_);
}

// the same two kernels reading the reference's own input layout -- fp32 [N][3][H][W] planes (dataset/voc.py:141-173) -- in their patch loaders: the
// fd_nchw3_to_nhwc4 pass (a read of the batch and a write of its 4-channel copy) disappears.  pool != 0: + ReLU + max-pool as fd_stem7x7_pool_nhwc4 (act ignored).
extern "C" int32_t fd_stem7x7_nchw3(const float* x, const float* w, const float* scale, const float* shift, float* y, int32_t y_cs, int32_t y_co,
                                    int32_t N, int32_t H, int32_t W, int32_t act, int32_t pool, fd_stream_t stream_) {
    FD_REQUIRE((long)N * 3 * H * W < (1L << 31), FD_E_UNSUPPORTED, "fd_stem7x7_nchw3: input exceeds 2^31 elements");
    return stem_launch(pool != 0, true, x, w, scale, shift, y, y_cs, y_co, N, H, W, pool ? FD_ACT_RELU : act, (hipStream_t)stream_);
}
