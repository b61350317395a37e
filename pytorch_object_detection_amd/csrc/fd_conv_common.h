// fd_conv_common.h -- what the implicit-GEMM conv kernels share: argument block, LDS swizzles, the fused epilogue.
#pragma once
#include "fd_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define FD_SPLIT_SCALE 2048.0f   // lo = (x - hi) * 2^11 keeps the low half in f16's normal range

struct ConvArgs {
    const float* x; const float* w; const float* scale; const float* shift; const float* res; float* y;
    int x_cs, x_co, res_cs, res_co, y_cs, y_co;
    int Cin, Cout, KW, stride, pad, dil, act, act_c0;
    int M, KT, ntaps, Kpacked;
    int nseg;
    int H[FD_MAX_SEG], W[FD_MAX_SEG], Ho[FD_MAX_SEG], Wo[FD_MAX_SEG];
    int m_in[FD_MAX_SEG];        // first input row of the segment
    int m_out[FD_MAX_SEG + 1];   // first output row of the segment
    float seg_param[FD_MAX_SEG];
    int mtiles, ntiles;
    int vec_epi;   // output / residual views are 16-byte addressable: transposed float4 epilogue
    int is_gemm;   // 1x1 stride-1 unpadded conv: pure GEMM addressing
    int Cout_epi;  // output-channel bound of the epilogue (= Cout, or the padded row length of a split-K slab)
    int kt_per;    // K-tiles per split-K slice (blockIdx.y = slice); KT when split-K is off
    long slice_stride;  // elements between consecutive split-K slabs in the workspace
    unsigned x_bytes, w_bytes;   // extents of the input / packed-weight buffers (raw buffer descriptors: OOB reads return 0)
    int res_mask;  // 1: `res` is a ReLU mask (y = res > 0 ? v : 0) instead of an addend
    int res_up;    // RUP kernels: `res` is a HALF-resolution map added AFTER the activation: y[n, i, j] = act(v) + res[n, i / 2, j / 2] (FPN top-down path)
    // output scatter (single level): output pixel (n, i, j) is written to row (n*sc_H + sc_sy*i + sc_oy)*sc_W + sc_sx*j + sc_ox of
    // y (and reads `res` there): one parity class of the data gradient of a strided conv lands interleaved in dX
    int sc_on, sc_sy, sc_sx, sc_oy, sc_ox, sc_H, sc_W;
    const float* gate; int gate_cs, gate_hw;   // per-(image, input channel) input gate (GATE kernels): rows of gate_cs floats, one per (level, image)
    const float* gate_b; int gate_act, gate_batch;   // GATE kernels: x' = act(x * gate + gate_b) (gate_b NULL: x * gate); images per level
    // row-group statistics of the STORED output (GroupNorm fused into the producer): gn_stats[m][g] = (sum, sum of squares) of the gn_cg
    // channels of group g in output row m, fp32 float2; gn_G groups over the conv's Cout channels
    float* gn_stats; int gn_G, gn_cg;
    // DUAL kernels: K-tiles >= kt2 come from a second 1x1 source (its own row addresses, sampled with x2_stride): conv3 + downsample as one GEMM
    const float* x2; int x2_cs, x2_co, x2_stride, x2_H, x2_W, kt2; unsigned x2_bytes;
    int p_halo;    // patch kernel (fd_conv_patch.hip): input rows staged on either side of an M-tile = dil * (max level width + 1)
    // FD_PREC_F16 (H1 kernels) only -- activations stored as f16 in HBM (fd_conv_params.io_f16): x / y / res hold _Float16 elements; views stay in ELEMENTS
    int x16, y16, res16;
    int wide8;     // FD_TILE_F16K64: the f16 output / residual views allow 16-byte (eight-channel) accesses
};

// four fp32 values <-> four consecutive f16 (8 bytes), round to nearest even -- the storage form of AMP activations
__device__ __forceinline__ float4 fd_ld_h4(const void* p) {
    const h4 v = *reinterpret_cast<const h4*>(p);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
// four f16 whose bits travelled as two 32-bit registers (an 8-byte f16 fetch kept in a float4's .x / .y).  (Through HIP's float2 CLASS the bit cast duplicated the
// first register: ext-vector types only.)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ h4 fd_h4_bits(float lo, float hi) {
    const f32x2 t = {lo, hi};
    return __builtin_bit_cast(h4, t);
}
__device__ __forceinline__ void fd_st_h4(void* p, const float4& v) {
    const f32x4 f = {v.x, v.y, v.z, v.w};
    *reinterpret_cast<h4*>(p) = __builtin_convertvector(f, h4);
}

// fd_conv_patch.hip: 3x3 stride-1 'same' conv with the (tile + halo) input patch staged ONCE per 32-channel chunk in LDS
#define FD_PATCH_BM 128
#define FD_PATCH_MAXROWS 320     /* FD_PATCH_BM + 2 * halo <= this (W <= 95 at dilation 1, <= 47 at dilation 2) */
int fd_launch_conv_patch(const ConvArgs& a, int tag, int split, hipStream_t stream);
// fd_conv_wino.hip: 3x3 stride-1 'same' conv as Winograd F(2x2, 3x3) (FD_TILE_WINOGRAD; p->w is the fd_wino_pack_weights_f32 packing)
int fd_launch_conv_wino(const fd_conv_params* p, hipStream_t stream);
// fd_conv_wino4.hip: 3x3 stride-1 pad-1 conv as Winograd F(4x4, 3x3) (FD_TILE_WINOGRAD4; p->w is the fd_wino4_pack_weights_f32 packing)
int fd_launch_conv_wino4(const fd_conv_params* p, hipStream_t stream);
int fd_wino4_workgroups(const fd_conv_params* p);
int fd_wino4_workgroups_live(const fd_conv_params* p, int first, int count);
// fd_conv_narrow.hip: 3x3 stride-1 pad-1 conv with Cout <= 8 on the vector unit (FD_TILE_NARROW; p->w is the [Cin/16][3][4][3][NCO][4] packing)
int fd_launch_conv_narrow(const fd_conv_params* p, hipStream_t stream);
// fd_conv_wave.hip: GEMM-addressed layers as wave-autonomous 64 x 64 tiles (FD_TILE_WAVE64); wfrag = fd_pack_conv_weight_wave_f32 packing
int fd_launch_conv_wave(const ConvArgs& a, const float* wfrag, hipStream_t stream);
// fd_conv_f16.hip: FD_PREC_F16 convs on K-tiles of 64 channels, f16 or fp32 activation maps (FD_TILE_F16K64; p->w is the fd_pack_conv_weight_f32 mode | 16 packing)
int fd_launch_conv_f16k64(const fd_conv_params* p, ConvArgs& a, hipStream_t stream);
// fd_conv.hip: y = act(sum over the nslice slabs of ws (in slice order) * scale + shift (+ | mask) res) -- the split-K combine launch
// (`orig`: scale / shift / res / y views, Cout, act, act_c0, M, nseg, m_out, seg_param, res_mask, sc_* of the real conv)
int fd_launch_splitk_reduce(const ConvArgs& orig, const float* ws, int nslice, int ldw, long slab, hipStream_t stream);

__device__ __forceinline__ long out_row(const ConvArgs& a, int m) {
    if (!a.sc_on) return m;
    const int hw = a.Ho[0] * a.Wo[0];
    const int n = m / hw, rem = m - n * hw;
    const int i = rem / a.Wo[0], j = rem - i * a.Wo[0];
    return ((long)n * a.sc_H + (a.sc_sy * i + a.sc_oy)) * a.sc_W + (a.sc_sx * j + a.sc_ox);
}

// RUP: the residual row of output row m = the pixel (i / 2, j / 2) of the [N][Ho / 2][Wo / 2] map (nearest-neighbour x2 upsampling read in place; single level)
__device__ __forceinline__ long res_row_up(const ConvArgs& a, int m) {
    const int W = a.Wo[0], hw = a.Ho[0] * W;
    const int n = m / hw, rem = m - n * hw;
    const int i = rem / W, j = rem - i * W;
    return ((long)n * (a.Ho[0] >> 1) + (i >> 1)) * (W >> 1) + (j >> 1);
}

// LDS hand-off between lanes of ONE wave: the LDS pipe executes a wave's ds instructions in order, so later reads see
// earlier writes once lgkmcnt has drained; the fence + wave barrier keep the compiler from moving accesses across.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Row-group statistics of a lane's four stored channels v (channels nn .. nn+3 of output row m): the gn_cg / 4 lanes that hold one group's
// channels of this row are adjacent (lane & 7 = 16-byte chunk of the row) -- summed by xor-shuffles in a fixed order, written by the first.
// EVERY lane of the wave must call it (the shuffles); `ok` only guards the store.
// lane ^ 1 / lane ^ 2 exchanges inside a quad as DPP modifiers of a VALU move (no LDS-pipe ds_bpermute)
__device__ __forceinline__ float fd_quad_xor1(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)); }
__device__ __forceinline__ float fd_quad_xor2(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)); }

__device__ __forceinline__ void fd_gn_rowstats(float* gn_stats, int gn_G, int gn_cg, const float4& v, size_t m, int nn, int lane, bool ok) {
    float s1 = (v.x + v.y) + (v.z + v.w);
    float s2 = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    const int cgq = gn_cg >> 2;                   // 1, 2, 4 or 8 lanes per group (uniform)
    if (cgq > 1) { s1 += fd_quad_xor1(s1); s2 += fd_quad_xor1(s2); }
    if (cgq > 2) { s1 += fd_quad_xor2(s1); s2 += fd_quad_xor2(s2); }
    if (cgq > 4) { s1 += __shfl_xor(s1, 4); s2 += __shfl_xor(s2, 4); }
    if (ok && ((lane & 7) & (cgq - 1)) == 0)
        reinterpret_cast<float2*>(gn_stats)[m * gn_G + nn / gn_cg] = make_float2(s1, s2);
}

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 32 + ((chunk ^ ((row >> 1) & 7)) << 2); }
// split-f16 planes: rows of 32 halves (64 B); c8 = 16-byte chunk (8 halves) 0..3, XOR-swizzled by (row>>2)&3
__device__ __forceinline__ int lds_off_h(int row, int c8) { return row * 32 + ((c8 ^ ((row >> 2) & 3)) << 3); }

// F(4x4, 3x3) weight transform of one (n, k) filter (fd_conv_wino4.hip): U = G g G^T, G = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6;
// 1/24 -1/12 1/6; 0 0 1], computed in double and rounded once; packed [ceil(N / 32)][K / 8][36 f][32 n][8 k] (zero rows past N).
// mode 0: n = cout, k = cin of w [Cout][Cin][3][3].  mode 1 (data gradient): n = cin, k = cout, taps flipped, times scale[cout].
__device__ __forceinline__ void fd_wino4_pack_one(const float* __restrict__ w, const float* __restrict__ scale, float* __restrict__ out,
                                                  int N, int K, int mode, int n, int k) {
    const double G[6][3] = {{0.25, 0.0, 0.0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                            {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
    double gg[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double v = 0.0;
            if (n < N) {
                if (mode == 0) v = w[((long)n * K + k) * 9 + r * 3 + c];
                else v = (double)w[((long)k * N + n) * 9 + (2 - r) * 3 + (2 - c)] * (scale ? (double)scale[k] : 1.0);
            }
            gg[r][c] = v;
        }
    double t[6][3];
#pragma unroll
    for (int a_ = 0; a_ < 6; ++a_)
#pragma unroll
        for (int c = 0; c < 3; ++c) t[a_][c] = G[a_][0] * gg[0][c] + G[a_][1] * gg[1][c] + G[a_][2] * gg[2][c];
    const int nbk = n >> 5, nl = n & 31, cc = k >> 3, kl = k & 7;
    float* o = out + (((long)nbk * (K >> 3) + cc) * 36) * 256 + nl * 8 + kl;
#pragma unroll
    for (int a_ = 0; a_ < 6; ++a_)
#pragma unroll
        for (int b = 0; b < 6; ++b)
            o[(a_ * 6 + b) * 256] = (float)(t[a_][0] * G[b][0] + t[a_][1] * G[b][1] + t[a_][2] * G[b][2]);
}

// Winograd F(2x2, 3x3) weight transform of ONE (n, k) filter: U = G g G^T, G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1], computed in double
// and rounded once, written to the [ceil(N / 32)][K / 8][16 f][32 n][8 k] packing of fd_conv_wino.hip (frequencies 12..15 negated: the
// kernel forms patch row 3 of B^T d B with the opposite sign).  mode 0: g = w[n][k] (N = Cout, K = Cin).  mode 1 (data-gradient
// conv: N = Cin, K = Cout): g[r][q] = w[k][n][2 - r][2 - q] * (scale ? scale[k] : 1).  n >= N writes zeros (padding rows).
__device__ __forceinline__ void fd_wino_pack_one(const float* __restrict__ w, const float* __restrict__ scale, float* __restrict__ out,
                                                 int N, int K, int mode, int n, int k) {
    double g[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double v = 0.0;
            if (n < N) {
                if (mode == 0) v = w[((long)n * K + k) * 9 + r * 3 + c];
                else v = (double)w[((long)k * N + n) * 9 + (2 - r) * 3 + (2 - c)] * (scale ? (double)scale[k] : 1.0);
            }
            g[r][c] = v;
        }
    double t[4][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        t[0][c] = g[0][c];
        t[1][c] = 0.5 * (g[0][c] + g[1][c] + g[2][c]);
        t[2][c] = 0.5 * (g[0][c] - g[1][c] + g[2][c]);
        t[3][c] = g[2][c];
    }
    const int nbk = n >> 5, nl = n & 31, cc = k >> 3, kl = k & 7;
    float* o = out + (((long)nbk * (K >> 3) + cc) * 16) * 256 + nl * 8 + kl;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const double sg = (a == 3) ? -1.0 : 1.0;
        o[(a * 4 + 0) * 256] = (float)(sg * t[a][0]);
        o[(a * 4 + 1) * 256] = (float)(sg * 0.5 * (t[a][0] + t[a][1] + t[a][2]));
        o[(a * 4 + 2) * 256] = (float)(sg * 0.5 * (t[a][0] - t[a][1] + t[a][2]));
        o[(a * 4 + 3) * 256] = (float)(sg * t[a][2]);
    }
}
