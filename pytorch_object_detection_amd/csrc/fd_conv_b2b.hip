// fd_conv_b2b.hip -- two GEMM-addressed layers back to back in ONE launch: a ResNet bottleneck's conv3 (+ BN + residual + ReLU) and the
// NEXT block's conv1 (+ BN + ReLU), torchvision Bottleneck.forward behind model/backbone/resnet50.py:68-80.
//
//     y = act1(x  . W1^T * scale1 + shift1 + res)        [M][N1]   (conv3: K1 -> N1 = 4 * planes; written to HBM: it is the next residual)
//     z = act2(y  . W2^T * scale2 + shift2)              [M][N2]   (next conv1: N1 -> N2)
//
// As two launches the N1-wide map y (256 channels at 160 x 160 in layer1: 420 MB per 16 images) is written by the first and read back by
// the second; here the rows of y a wave has just produced are multiplied by W2 while they are still in its LDS, so that read disappears.
// Wave-autonomous like fd_conv_wave.hip (one wave = one workgroup = 32 TM rows, no barrier anywhere, weights in MFMA fragment order
// straight from L2):
//   for every 64-channel tile n1t of N1:
//     phase 1  acc1[TM][2] = x[rows][K1] . W1[n1t]^T         K1 / 32 K-tiles through the wave's LDS stage (x is re-read from L1 / L2 per tile)
//     for the two 32-channel halves j of the tile:
//       epilogue 1 of sub-tiles (i, j): scale / shift, residual, activation -> 16-byte stores of y, and the same values into the wave's
//                  second LDS stage in A-operand layout [rows][32 k]
//       phase 2  acc2[TM][TN2] += that stage . W2[:, 64 n1t + 32 j ..]^T      one K-tile of the second GEMM
//   epilogue 2: z.
// Both GEMMs add their products in the k order of the workgroup-tiled / wave kernels (K-tiles ascending, same order inside a K-tile), so y and z
// are BIT-IDENTICAL to the two-launch plan's (tests/test_layers_gpu.py::test_conv1x1_back_to_back_...).
#include "fd_conv_common.h"

struct B2BArgs {
    const float* x; const float* res; float* y; float* z;
    const float* scale1; const float* shift1; const float* scale2; const float* shift2;
    int x_cs, x_co, res_cs, res_co, y_cs, y_co, z_cs, z_co;
    int M, KT1, NT1, N2, act1, act2;
    unsigned x_bytes, y_bytes, z_bytes, res_bytes;     // extents of the views' buffers: rows past M fall outside and are dropped (stores) / read as zero (loads)
};

template <int TM, int TN2>
__global__ __launch_bounds__(64, 2) void conv1x1_b2b_kernel(B2BArgs a, const float* __restrict__ wf1_, const float* __restrict__ wf2_) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);                 // [32 TM rows][32 k]: phase-1 operand stage, and the epilogues' 4 KiB transposition stage
    float* A2 = As + 32 * TM * 32;                              // [32 TM rows][32 k]: 32 channels of y as the phase-2 operand
    constexpr int NL = 4 * TM;                                  // float4 loads per lane and K-tile of x (8 lanes per 128-byte row)
    const int lane = threadIdx.x, l31 = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.x * 32 * TM;

    constexpr unsigned OOB = 0xC0000000u;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, (short)0, (int)a.x_bytes, 0x00020000);
    const int lrow = lane >> 3, chunk = lane & 7;
    unsigned a_off[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int m = m0 + lrow + 8 * i;
        a_off[i] = (m < a.M) ? ((unsigned)m * (unsigned)a.x_cs + (unsigned)(a.x_co + chunk * 4)) * 4u : OOB;
    }
    const float4* __restrict__ wf1 = reinterpret_cast<const float4*>(wf1_) + lane;     // [n1t][KT1][2 j][4 s][64 lanes]
    const float4* __restrict__ wf2 = reinterpret_cast<const float4*>(wf2_) + lane;     // [nt2][KT2 = 2 NT1][2 j][4 s][64 lanes]
    const int KT1 = a.KT1, KT2 = 2 * a.NT1;

    float4 ra[NL];
    auto load_a = [&](int kt) {
        const unsigned kb = (unsigned)kt * 128u;
#pragma unroll
        for (int i = 0; i < NL; ++i)
            ra[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)(a_off[i] + kb), 0, 0));
    };
    auto store_a = [&]() {
#pragma unroll
        for (int i = 0; i < NL; ++i) *reinterpret_cast<float4*>(As + lds_off(lrow + 8 * i, chunk)) = ra[i];
    };

    f32x16 acc2[TM][TN2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc2[i][j][e] = 0.f;

    const int c4 = (lane & 7) * 4, prow = lane >> 3;
    float* stage = As;
    // y / z / residual through raw buffer descriptors: a lane's row offset is ONE register, the sub-tile / row-step / channel-block part goes in the
    // instruction's scalar offset, and rows past M are outside the descriptor (no exec-mask branches, no 64-bit address arithmetic in the loop)
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, (short)0, (int)a.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t zrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.z, (short)0, (int)a.z_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.res ? a.res : a.x), (short)0, (int)a.res_bytes, 0x00020000);   // res NULL: 0 bytes, every load reads 0
    const unsigned y_off = ((unsigned)(m0 + prow) * (unsigned)a.y_cs + (unsigned)(a.y_co + c4)) * 4u;
    const unsigned z_off = ((unsigned)(m0 + prow) * (unsigned)a.z_cs + (unsigned)(a.z_co + c4)) * 4u;
    const unsigned r_off = ((unsigned)(m0 + prow) * (unsigned)a.res_cs + (unsigned)(a.res_co + c4)) * 4u;

    for (int n1t = 0; n1t < a.NT1; ++n1t) {
        // ---------------- phase 1: acc1 = x . W1[n1t]^T ----------------
        f32x16 acc1[TM][2];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc1[i][j][e] = 0.f;
        // B fragments travel two k-steps ahead of their MFMAs in a two-deep register ring (16 registers; a whole K-tile ahead as in fd_conv_wave.hip
        // would be 32, and this kernel carries a second accumulator set)
        float4 fb[2][2];
        const float4* __restrict__ w1t = wf1 + (size_t)n1t * KT1 * 512;
        load_a(0);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int s = 0; s < 2; ++s) fb[j][s] = w1t[(j * 4 + s) * 64];
        for (int kt = 0; kt < KT1; ++kt) {
            store_a();
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 1 < KT1) load_a(kt + 1);
            __builtin_amdgcn_sched_barrier(0);
            wave_lds_sync();
            const float4* __restrict__ wc_ = w1t + (size_t)kt * 512;
            const float4* __restrict__ wn_ = w1t + (size_t)min(kt + 1, KT1 - 1) * 512;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float4 fa[TM];
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const float4*>(As + lds_off(i * 32 + l31, 2 * s + lh));
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j][s & 1].x, acc1[i][j], 0, 0, 0);
                        acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j][s & 1].y, acc1[i][j], 0, 0, 0);
                        acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j][s & 1].z, acc1[i][j], 0, 0, 0);
                        acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j][s & 1].w, acc1[i][j], 0, 0, 0);
                    }
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[j][s & 1] = s < 2 ? wc_[(j * 4 + s + 2) * 64] : wn_[(j * 4 + s - 2) * 64];   // the fragment two k-steps ahead
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_setprio(0);
            wave_lds_sync();                       // every fragment read of this K-tile is done before the stage is written again
        }

        // ---------------- epilogue 1 + phase 2, one 32-channel half of the tile at a time ----------------
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int nb = n1t * 64 + 32 * j;
            const float sc = a.scale1 ? a.scale1[nb + l31] : 1.0f, sf = a.shift1 ? a.shift1[nb + l31] : 0.0f;
            // the first two k-steps' fragments of this phase-2 K-tile are requested now and land under the epilogue; the other two follow in the ring
            float4 fb2[TN2][2];
            const int kt2 = 2 * n1t + j;
            auto w2at = [&](int jj, int s) { return wf2[((size_t)((jj >> 1) * KT2 + kt2) * 8 + (jj & 1) * 4 + s) * 64]; };
#pragma unroll
            for (int jj = 0; jj < TN2; ++jj)
#pragma unroll
                for (int s = 0; s < 2; ++s) fb2[jj][s] = w2at(jj, s);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                float4 rr[4];
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    rr[p] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, (int)r_off, ((32 * i + 8 * p) * a.res_cs + nb) * 4, 0));
#pragma unroll
                for (int e = 0; e < 16; ++e) stage[((e & 3) + 8 * (e >> 2) + 4 * lh) * 32 + l31] = acc1[i][j][e] * sc + sf;
                wave_lds_sync();
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    float4 v = *reinterpret_cast<const float4*>(stage + (prow + 8 * p) * 32 + c4);
                    const float4 r = rr[p];
                    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                    if (a.act1 == FD_ACT_RELU) {
                        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                    } else if (a.act1 == FD_ACT_SILU) {
                        v.x = fd_act(v.x, FD_ACT_SILU, 0.f); v.y = fd_act(v.y, FD_ACT_SILU, 0.f); v.z = fd_act(v.z, FD_ACT_SILU, 0.f); v.w = fd_act(v.w, FD_ACT_SILU, 0.f);
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(f32x4, v), yrsrc, (int)y_off, ((32 * i + 8 * p) * a.y_cs + nb) * 4, 0);
                    *reinterpret_cast<float4*>(A2 + lds_off(32 * i + prow + 8 * p, lane & 7)) = v;        // the same values as the second GEMM's operand
                }
                wave_lds_sync();
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float4 fa[TM];
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const float4*>(A2 + lds_off(i * 32 + l31, 2 * s + lh));
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int jj = 0; jj < TN2; ++jj) {
                        acc2[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb2[jj][s & 1].x, acc2[i][jj], 0, 0, 0);
                        acc2[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb2[jj][s & 1].y, acc2[i][jj], 0, 0, 0);
                        acc2[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb2[jj][s & 1].z, acc2[i][jj], 0, 0, 0);
                        acc2[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb2[jj][s & 1].w, acc2[i][jj], 0, 0, 0);
                    }
                if (s < 2) {
#pragma unroll
                    for (int jj = 0; jj < TN2; ++jj) fb2[jj][s & 1] = w2at(jj, s + 2);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_setprio(0);
            wave_lds_sync();                       // A2 is read: the next half may overwrite it
        }
    }

    // ---------------- epilogue 2: z = act2(acc2 * scale2 + shift2) ----------------
#pragma unroll
    for (int jj = 0; jj < TN2; ++jj) {
        const int nb = 32 * jj;
        if (nb >= a.N2) break;
        const float sc = a.scale2 ? a.scale2[nb + l31] : 1.0f, sf = a.shift2 ? a.shift2[nb + l31] : 0.0f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) stage[((e & 3) + 8 * (e >> 2) + 4 * lh) * 32 + l31] = acc2[i][jj][e] * sc + sf;
            wave_lds_sync();
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float4 v = *reinterpret_cast<const float4*>(stage + (prow + 8 * p) * 32 + c4);
                if (a.act2 == FD_ACT_RELU) {
                    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                } else if (a.act2 == FD_ACT_SILU) {
                    v.x = fd_act(v.x, FD_ACT_SILU, 0.f); v.y = fd_act(v.y, FD_ACT_SILU, 0.f); v.z = fd_act(v.z, FD_ACT_SILU, 0.f); v.w = fd_act(v.w, FD_ACT_SILU, 0.f);
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(f32x4, v), zrsrc, (int)z_off, ((32 * i + 8 * p) * a.z_cs + nb) * 4, 0);
            }
            wave_lds_sync();
        }
    }
}

extern "C" int32_t fd_conv1x1_b2b_f32(const fd_b2b_params* p, fd_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    FD_REQUIRE(p && p->x && p->w1_frag && p->w2_frag && p->y && p->z, FD_E_INVAL, "fd_conv1x1_b2b: null pointer");
    FD_REQUIRE(p->rows >= 1 && p->K1 >= 32 && p->K1 % 32 == 0 && p->N1 >= 64 && p->N1 % 64 == 0 && (p->N2 == 64 || p->N2 == 128), FD_E_UNSUPPORTED,
               "fd_conv1x1_b2b: needs K1 %% 32 == 0, N1 %% 64 == 0, N2 in {64, 128} (got %d -> %d -> %d)", p->K1, p->N1, p->N2);
    FD_REQUIRE(p->x_cs % 4 == 0 && p->x_co % 4 == 0 && p->x_cs >= p->x_co + p->K1 && p->y_cs % 4 == 0 && p->y_co % 4 == 0 && p->y_cs >= p->y_co + p->N1 &&
                   p->z_cs % 4 == 0 && p->z_co % 4 == 0 && p->z_cs >= p->z_co + p->N2 &&
                   (!p->res || (p->res_cs % 4 == 0 && p->res_co % 4 == 0 && p->res_cs >= p->res_co + p->N1)),
               FD_E_INVAL, "fd_conv1x1_b2b: channel views must be 4-aligned and in range");
    FD_REQUIRE((((uintptr_t)p->x | (uintptr_t)p->y | (uintptr_t)p->z | (uintptr_t)p->res | (uintptr_t)p->w1_frag | (uintptr_t)p->w2_frag) & 15) == 0, FD_E_INVAL,
               "fd_conv1x1_b2b: pointers must be 16-byte aligned");
    auto act_ok = [](int a_) { return a_ == FD_ACT_NONE || a_ == FD_ACT_RELU || a_ == FD_ACT_SILU; };
    FD_REQUIRE(act_ok(p->act1) && act_ok(p->act2), FD_E_UNSUPPORTED, "fd_conv1x1_b2b: activations none / ReLU / SiLU");
    const long xb = (long)p->rows * p->x_cs * 4;
    FD_REQUIRE(xb < 0xC0000000L - 65536 && (long)p->rows * p->y_cs * 4 < 0xC0000000L && (long)p->rows * p->z_cs * 4 < 0xC0000000L &&
                   (!p->res || (long)p->rows * p->res_cs * 4 < 0xC0000000L), FD_E_UNSUPPORTED, "fd_conv1x1_b2b: a view's buffer exceeds 3 GiB");
    B2BArgs a;
    a.x = p->x; a.res = p->res; a.y = p->y; a.z = p->z;
    a.scale1 = p->scale1; a.shift1 = p->shift1; a.scale2 = p->scale2; a.shift2 = p->shift2;
    a.x_cs = p->x_cs; a.x_co = p->x_co; a.res_cs = p->res_cs; a.res_co = p->res_co; a.y_cs = p->y_cs; a.y_co = p->y_co; a.z_cs = p->z_cs; a.z_co = p->z_co;
    a.M = (int)p->rows; a.KT1 = p->K1 / 32; a.NT1 = p->N1 / 64; a.N2 = p->N2; a.act1 = p->act1; a.act2 = p->act2;
    a.x_bytes = (unsigned)xb;
    a.y_bytes = (unsigned)((long)p->rows * p->y_cs * 4); a.z_bytes = (unsigned)((long)p->rows * p->z_cs * 4);
    a.res_bytes = p->res ? (unsigned)((long)p->rows * p->res_cs * 4) : 0u;
    if (p->N2 == 64) {      // 64-row waves (32-row waves for N2 = 64 measured slower: DESIGN 4.1h)
        const unsigned blocks = (unsigned)((p->rows + 63) / 64);
        hipLaunchKernelGGL((conv1x1_b2b_kernel<2, 2>), dim3(blocks), dim3(64), 2 * 64 * 32 * 4, stream, a, p->w1_frag, p->w2_frag);
    } else {
        const unsigned blocks = (unsigned)((p->rows + 31) / 32);
        hipLaunchKernelGGL((conv1x1_b2b_kernel<1, 4>), dim3(blocks), dim3(64), 2 * 32 * 32 * 4, stream, a, p->w1_frag, p->w2_frag);
    }
    FD_CHECK_LAUNCH("fd_conv1x1_b2b_f32");
    return FD_OK;
}
