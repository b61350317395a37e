// fd_conv.hip — implicit-GEMM convolution for gfx950 on v_mfma_f32_32x32x2_f32 (exact fp32).
//
//   C[m][n] = sum_k A[m][k] * B[n][k]      m = output pixel (NHWC row), n = output channel,
//                                          k = ((c/32)*KH*KW + r*KW + q)*32 + c%32  (32-channel chunk, tap, channel)
//   A is gathered on the fly from the NHWC input (zero outside the image), B is the weight tensor packed
//   [Cout][Cin/32][KH][KW][32].  Both operands are staged as [rows][32 k] tiles in LDS (16-byte chunks XOR-swizzled
//   by (row>>1)&7, conflict-free for ds_read_b128 and ds_write_b128), global->register->LDS double buffered:
//   the loads of k-tile t+1 are issued before the MFMAs of tile t and written to the other LDS buffer after.
//   Each of the 4 waves owns TM x TN sub-tiles of 32x32 (16 accumulator VGPRs each); a ds_read_b128 hands a
//   lane 4 consecutive k of one row, consumed by 4 MFMAs (lanes 0-31 carry k, lanes 32-63 carry k+4).
//   Epilogue: y = act(acc*scale[n] + shift[n] + res), columns (n) on lanes -> 128-byte coalesced NHWC stores.
//
// Replaces nn.Conv2d(+BatchNorm2d eval)+ReLU/SiLU(+add) of the reference models (see include/fcosdet.h).
#include "fd_conv_common.h"

// GATE: the A operand is multiplied by a per-(image, input channel) gate on its way to LDS (1x1 GEMM layers, single level): the
// squeeze-excitation gate of an MBConv block folded into its project conv.
// GNS: the epilogue also writes the row-group statistics of the stored output (fd_conv_params.gn_stats; one- and two-sub-tile tiles only).
// H1 (with SPLIT): single-plane f16 -- operands rounded to f16 once, ONE v_mfma_f32_32x32x16_f16 per product, fp32 accumulation: the
// arithmetic of torch.autocast(float16) convolutions (FD_PREC_F16; the reference trains under AMP, train.py:33,175-181).
// DUAL (with GEMM): K-tiles >= a.kt2 are gathered from a second 1x1 source a.x2 (K-concatenation: a bottleneck's conv3 and its downsample conv as one GEMM).
// RUP (with GEMM): `res` is a half-resolution map read at (i / 2, j / 2) and added AFTER the activation: an FPN lateral conv with the x2-upsampled coarser level
// folded into its epilogue (HISFcos.py:155-165, Fcos.py:77-91) -- the upsample-add pass over the finer map disappears.
template <int WGM, int WGN, int TM, int TN, bool STEM, bool SB, int TAG, bool SPLIT, bool GATE = false, bool GNS = false, bool H1 = false, bool GEMM = false, bool DUAL = false,
          bool RUP = false>
__global__ __launch_bounds__(WGM * WGN * 64, (TM * TN == 4) ? ((SPLIT || WGM * WGN == 8) ? 2 : (SB ? 3 : 1)) : 1)
void conv_igemm_kernel(ConvArgs a) {
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    constexpr int NT = WGM * WGN * 64;         // threads: 4 waves (256) or 8 waves (512: the 256x128 tile)
    constexpr int RPP = NT / 8;                // rows per loader pass (8 lanes fetch one 128-byte row segment)
    constexpr int AP = BM / RPP, BP = BN / RPP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NBUF = SB ? 1 : 2;             // SB: one LDS buffer (2 barriers per K-tile, half the LDS -> more blocks/CU)
    float* As = reinterpret_cast<float*>(smem);  // [NBUF][BM*32]
    float* Bs = As + NBUF * BM * 32;             // [NBUF][BN*32]
    // SPLIT (f16 x 3): the same bytes hold, per buffer, four f16 planes  A_hi | A_lo | B_hi | B_lo  of [rows][32 k]
    _Float16* Hs = reinterpret_cast<_Float16*>(smem);
    constexpr int HSTG = (BM + BN) * 64;         // halves per buffer

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;

    // XCD-aware tile order: blocks dealt round-robin over 8 XCDs -> give each XCD a contiguous tile range
    const int nblk = a.mtiles * a.ntiles;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int mt = bid / a.ntiles, nt = bid - mt * a.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;

    // ---- per-thread A rows: decode (segment, image, ho, wo) once ----
    // Operands are fetched with raw buffer loads: an out-of-range byte offset returns zeros, so padding taps, rows past
    // M and channels past Cout need no exec-mask branch and no zero fill; per K-tile a row costs one 24-bit mad, two
    // adds, two compares and a select.
    const int lrow = tid >> 3, chunk = tid & 7;
    constexpr unsigned OOB = 0xC0000000u;       // >= any buffer size (host checks < 3 GiB): reads as zero
    // H1 with a.x16: the input map holds f16 elements (AMP activations stored as f16): byte offsets are element offsets << 1, a lane fetches its four channels as
    // 8 bytes and they go to LDS as they are -- half the bytes, no conversion.  `esh` is uniform; in every other kernel it is the constant 2.
    const int esh = (H1 && a.x16) ? 1 : 2;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, (short)0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, (short)0, (int)a.w_bytes, 0x00020000);
    unsigned a_off[AP];                         // byte offset of tap (0, 0), channel chunk 0 (may wrap below zero)
    int a_wcs[AP], a_hi0[AP], a_wi0[AP], a_H[AP], a_W[AP];
    const float* g_row[GATE ? AP : 1];          // GATE: this row's (level, image) gate vector, at this thread's channel quad
    const long gate_bo = GATE && a.gate_b ? (long)(a.gate_b - a.gate) : 0;     // element distance gate -> gate_b (same row layout)
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int m = m0 + lrow + RPP * i;
        if (GATE) {
            const int mm = min(m, a.M - 1);
            int sg = 0;
#pragma unroll
            for (int t = 1; t < FD_MAX_SEG; ++t)
                if (t < a.nseg && mm >= a.m_out[t]) sg = t;
            const int img = sg * a.gate_batch + (mm - a.m_out[sg]) / (a.Ho[sg] * a.Wo[sg]);
            g_row[i] = a.gate + (size_t)img * a.gate_cs + chunk * 4;
        }
        if constexpr (GEMM) {   // compiled for 1x1 unpadded layers with Cin % 32 == 0: a row has ONE input address, the K-tile goes in the scalar offset
            if (a.is_gemm) {    // stride 1: the input row is the output row
                a_off[i] = (m < a.M) ? ((unsigned)m * (unsigned)a.x_cs + (unsigned)(a.x_co + chunk * 4)) << esh : OOB;
                a_wcs[i] = 0; a_hi0[i] = 0; a_wi0[i] = 0; a_H[i] = 1; a_W[i] = 1;
                continue;
            }                   // strided 1x1 (the ResNet downsample convs): the decode below gives the row's address once; rows past M read zeros
        }
        if (a.is_gemm) {  // 1x1, stride 1, no padding: the input row IS the output row (no divisions)
            a_off[i] = ((unsigned)m * (unsigned)a.x_cs + (unsigned)(a.x_co + chunk * 4)) << esh;
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
        a_H[i] = (m < a.M) ? H : 0;  // H = 0 makes every tap invalid for rows past M
        a_W[i] = W;
        a_wcs[i] = (W * a.x_cs) << esh;   // bytes per input image row
        a_off[i] = ((unsigned)(a.m_in[s] + n * H * W + a_hi0[i] * W + a_wi0[i]) * (unsigned)a.x_cs +
                    (unsigned)(a.x_co + chunk * 4)) << esh;
        if constexpr (GEMM) { if (m >= a.M) a_off[i] = OOB; }
    }
    // DUAL: the second source's row addresses (single level; output pixel (n, i, j) reads x2[n, s * i, s * j]) and its buffer descriptor
    unsigned a2_off[DUAL ? AP : 1];
    const __amdgpu_buffer_rsrc_t x2rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(DUAL ? a.x2 : a.x), (short)0, (int)(DUAL ? a.x2_bytes : 0u), 0x00020000);
    if constexpr (DUAL) {
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int m = m0 + lrow + RPP * i;
            const int hw = a.Ho[0] * a.Wo[0];
            const int n = m / hw, rem = m - n * hw;
            const int ho = rem / a.Wo[0], wo = rem - ho * a.Wo[0];
            const unsigned row2 = (unsigned)((n * a.x2_H + ho * a.x2_stride) * a.x2_W + wo * a.x2_stride);
            a2_off[i] = (m < a.M) ? (row2 * (unsigned)a.x2_cs + (unsigned)(a.x2_co + chunk * 4)) * 4u : OOB;
        }
    }
    unsigned b_off[BP];
#pragma unroll
    for (int j = 0; j < BP; ++j) {
        const int n = n0 + lrow + RPP * j;
        b_off[j] = (n < a.Cout && !(H1 && (chunk & 4))) ? ((unsigned)n * (unsigned)a.Kpacked + (unsigned)(chunk * 4)) * 4u : OOB;   // (H1: the lo plane is not fetched)
    }

    float4 ra[AP], rb[BP], rg[GATE ? AP : 1], rgb[GATE ? AP : 1];
    // one lane's four input channels: 16 bytes of fp32 -- or (H1, a.x16) 8 bytes of f16, kept as bits in .x / .y
    auto ld_x = [&](unsigned voff, int soff) -> float4 {
        if (H1 && a.x16) {
            // (the WHOLE result is bit-cast: indexing the builtin's vector element-wise made the compiler fetch one dword and use it twice)
            const f32x2 v = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xrsrc, (int)voff, soff, 0));
            return make_float4(v.x, v.y, 0.f, 0.f);
        }
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)voff, soff, 0));
    };
    // k-tile order = (32-channel chunk, tap): the 9 taps of one channel chunk are consecutive, so the shifted re-reads
    // of the same input pixels hit L1/L2 instead of travelling from the Infinity Cache.  load_tile() is called with
    // consecutive kt, so (chunk, filter row, filter column) advance as counters (no divisions in the K loop).
    int ld_cc = 0, ld_r = 0, ld_q = 0;
    auto load_tile = [&](int kt) {
        if constexpr (GEMM) {      // no per-row arithmetic: lane offsets fixed, K-tile kt = 128 bytes further along every row (scalar offset)
            const int kb = kt * 128;
            const int kbx = kt << (5 + esh);            // the input's K-tile step in bytes (64 for an f16 map)
            if (GATE) {
#pragma unroll
                for (int i = 0; i < AP; ++i) {
                    rg[i] = *reinterpret_cast<const float4*>(g_row[i] + kt * 32);
                    if (a.gate_b) rgb[i] = *reinterpret_cast<const float4*>(g_row[i] + gate_bo + kt * 32);
                }
            }
            if (DUAL && kt >= a.kt2) {      // (uniform) this K-tile's channels belong to the second source
                const int kb2 = (kt - a.kt2) * 128;
#pragma unroll
                for (int i = 0; i < AP; ++i) ra[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(x2rsrc, (int)a2_off[DUAL ? i : 0], kb2, 0));
            } else {
#pragma unroll
                for (int i = 0; i < AP; ++i) ra[i] = ld_x(a_off[i], kbx);
            }
#pragma unroll
            for (int j = 0; j < BP; ++j) rb[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (int)b_off[j], kb, 0));
            return;
        }
        int dr, dq;          // tap displacement in input rows / columns
        unsigned dbytes;     // uniform byte displacement: column shift + channel chunk
        bool c_ok = true;    // Cin % 32 != 0 (EfficientNet widths: 24, 40, 48, 136, 144, 232 ...): the last chunk's missing
                             // channels read as zero (their packed weights are zero too, but 0 * garbage could be NaN)
        if (STEM) { dr = kt; dq = 0; dbytes = 0; }
        else {
            dr = ld_r * a.dil;
            dq = ld_q * a.dil;
            dbytes = (unsigned)(dq * a.x_cs + ld_cc * 32) << esh;
            c_ok = ld_cc * 32 + chunk * 4 < a.Cin;
            if (GATE) {
#pragma unroll
                for (int i = 0; i < AP; ++i) {
                    rg[i] = c_ok ? *reinterpret_cast<const float4*>(g_row[i] + ld_cc * 32) : make_float4(0.f, 0.f, 0.f, 0.f);
                    if (a.gate_b) rgb[i] = c_ok ? *reinterpret_cast<const float4*>(g_row[i] + gate_bo + ld_cc * 32) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            if (++ld_q == a.KW) { ld_q = 0; if (++ld_r * a.KW == a.ntaps) { ld_r = 0; ++ld_cc; } }
        }
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int hi = a_hi0[i] + dr, wi = a_wi0[i] + dq + (STEM ? chunk : 0);
            bool ok = (unsigned)hi < (unsigned)a_H[i] && (unsigned)wi < (unsigned)a_W[i] && c_ok;
            if (STEM) ok = ok && (chunk < 7);
            const unsigned off = a_off[i] + (unsigned)__mul24(dr, a_wcs[i]) + dbytes;
            ra[i] = ld_x(ok ? off : OOB, 0);
        }
        const unsigned kb = (unsigned)kt * 128u;
#pragma unroll
        for (int j = 0; j < BP; ++j)
            rb[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (int)(b_off[j] + kb), 0, 0));
    };
    auto store_tile = [&](int buf) {
        if constexpr (SPLIT) {
            _Float16* Ahi = Hs + buf * HSTG;
            _Float16* Alo = Ahi + BM * 32;
            _Float16* Bhi = Alo + BM * 32;
            _Float16* Blo = Bhi + BN * 32;
#pragma unroll
            for (int i = 0; i < AP; ++i) {
                const int row = lrow + RPP * i;
                const f32x4 v = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
                h4 hi = __builtin_convertvector(v, h4);                             // round to nearest
                if (H1 && a.x16) hi = fd_h4_bits(ra[i].x, ra[i].y);      // (uniform) already f16: the bits as loaded
                const int off = row * 32 + ((((chunk >> 1) ^ ((row >> 2) & 3)) << 3) | ((chunk & 1) << 2));
                *reinterpret_cast<h4*>(Ahi + off) = hi;
                if constexpr (!H1) {
                    const f32x4 rem = (v - __builtin_convertvector(hi, f32x4)) * FD_SPLIT_SCALE;   // exact residual
                    const h4 lo = __builtin_convertvector(rem, h4);
                    *reinterpret_cast<h4*>(Alo + off) = lo;
                }
            }
#pragma unroll
            for (int j = 0; j < BP; ++j) {   // weights arrive pre-split: 16-B chunks 0-3 = hi, 4-7 = lo of this K-tile
                const int row = lrow + RPP * j;
                if (!H1 || !(chunk & 4)) *reinterpret_cast<float4*>(((chunk & 4) ? Blo : Bhi) + lds_off_h(row, chunk & 3)) = rb[j];
            }
        } else {
#pragma unroll
            for (int i = 0; i < AP; ++i) {
                if (GATE) {
                    if (a.gate_b) {      // the preceding GroupNorm's affine + activation, applied on the way to LDS (uniform branches)
                        float4 t = make_float4(fmaf(ra[i].x, rg[i].x, rgb[i].x), fmaf(ra[i].y, rg[i].y, rgb[i].y),
                                               fmaf(ra[i].z, rg[i].z, rgb[i].z), fmaf(ra[i].w, rg[i].w, rgb[i].w));
                        if (a.gate_act == FD_ACT_RELU) {
                            t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f);
                        } else if (a.gate_act == FD_ACT_SILU) {
                            t.x *= __builtin_amdgcn_rcpf(1.0f + __expf(-t.x)); t.y *= __builtin_amdgcn_rcpf(1.0f + __expf(-t.y));
                            t.z *= __builtin_amdgcn_rcpf(1.0f + __expf(-t.z)); t.w *= __builtin_amdgcn_rcpf(1.0f + __expf(-t.w));
                        }
                        ra[i] = t;
                    } else { ra[i].x *= rg[i].x; ra[i].y *= rg[i].y; ra[i].z *= rg[i].z; ra[i].w *= rg[i].w; }
                }
                *reinterpret_cast<float4*>(As + buf * BM * 32 + lds_off(lrow + RPP * i, chunk)) = ra[i];
            }
#pragma unroll
            for (int j = 0; j < BP; ++j)
                *reinterpret_cast<float4*>(Bs + buf * BN * 32 + lds_off(lrow + RPP * j, chunk)) = rb[j];
        }
    };

    f32x16 acc[TM][TN];
    f32x16 cor[SPLIT ? TM : 1][SPLIT ? TN : 1];   // SPLIT: hi*lo + lo*hi cross terms (scaled by 2^11)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc[i][j][e] = 0.f;
                if constexpr (SPLIT) cor[i][j][e] = 0.f;
            }

    const int l31 = lane & 31, lh = lane >> 5;
    // split-K: slice blockIdx.y owns K-tiles [kt0, kt1) and writes raw partial sums to its slab of the workspace
    const int kt0 = blockIdx.y * a.kt_per;
    const int kt1 = min(a.KT, kt0 + a.kt_per);
    if (!STEM) {
        ld_cc = kt0 / a.ntaps;
        const int tap0 = kt0 - ld_cc * a.ntaps;
        ld_r = tap0 / a.KW;
        ld_q = tap0 - ld_r * a.KW;
    }
    float* const ybase = a.y + (size_t)blockIdx.y * a.slice_stride;   // (never write to the kernarg struct itself)
    auto mfma_tile = [&](int buf) {
        if constexpr (SPLIT) {
            const _Float16* Ahi = Hs + buf * HSTG + (wm * TM * 32) * 32;
            const _Float16* Alo = Ahi + BM * 32;
            const _Float16* Bhi = Hs + buf * HSTG + 2 * BM * 32 + (wn * TN * 32) * 32;
            const _Float16* Blo = Bhi + BN * 32;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {   // two K=16 steps per 32-wide tile; lane half lh carries k = 8*lh .. 8*lh+7
                h8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    ah[i] = *reinterpret_cast<const h8*>(Ahi + lds_off_h(i * 32 + l31, 2 * ks + lh));
                    if constexpr (!H1) al[i] = *reinterpret_cast<const h8*>(Alo + lds_off_h(i * 32 + l31, 2 * ks + lh));
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    bh[j] = *reinterpret_cast<const h8*>(Bhi + lds_off_h(j * 32 + l31, 2 * ks + lh));
                    if constexpr (!H1) bl[j] = *reinterpret_cast<const h8*>(Blo + lds_off_h(j * 32 + l31, 2 * ks + lh));
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                        if constexpr (!H1) {
                            cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], cor[i][j], 0, 0, 0);
                            cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], cor[i][j], 0, 0, 0);
                        }
                    }
            }
        } else {
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
        }
    };
    load_tile(kt0);
    store_tile(0);
    __syncthreads();
    for (int kt = kt0; kt < kt1 - 1; ++kt) {
        const int buf = SB ? 0 : ((kt - kt0) & 1);
        load_tile(kt + 1);
        mfma_tile(buf);
        if (SB) __syncthreads();                   // every wave is done reading the tile
        store_tile(SB ? 0 : (buf ^ 1));
        __syncthreads();
    }
    // last K-tile (peeled: no operand loads left): the residual tile is fetched under its MFMAs
#include "fd_conv_res_prefetch.inc"
    mfma_tile(SB ? 0 : ((kt1 - 1 - kt0) & 1));
    __syncthreads();

#define FD_EPI_RES_PREFETCHED
#include "fd_conv_epilogue.inc"
#undef FD_EPI_RES_PREFETCHED
}

// split-K combine: y = act(sum_slices(ws) * scale + shift + res); one float4 of output channels per thread, slices
// added in index order (deterministic).  `a` is the ORIGINAL conv (real y / scale / shift / res / act).
__global__ __launch_bounds__(256) void splitk_reduce_kernel(ConvArgs a, const float* __restrict__ ws, int nslice, int ldw,
                                                             long slab) {
    const int q4 = ldw >> 2;
    const long total = (long)a.M * q4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long m = i / q4;
        const int nn = (int)(i - m * q4) * 4;
        float4 v = *reinterpret_cast<const float4*>(ws + m * ldw + nn);
        for (int s = 1; s < nslice; ++s) {
            const float4 u = *reinterpret_cast<const float4*>(ws + s * slab + m * ldw + nn);
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        float o[4] = {v.x, v.y, v.z, v.w};
        float prm = a.seg_param[0];      // (select chain over uniform argument reads: see fd_conv_epilogue.inc)
        if (a.act == FD_ACT_EXP) {
#pragma unroll
            for (int t = 1; t < FD_MAX_SEG; ++t)
                prm = (t < a.nseg && m >= a.m_out[t]) ? a.seg_param[t] : prm;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = nn + e;
            if (n >= a.Cout) continue;
            float r = o[e] * (a.scale ? a.scale[n] : 1.0f) + (a.shift ? a.shift[n] : 0.0f);
            const size_t mo = (size_t)out_row(a, (int)m);
            if (a.res) {
                const float rv = a.res16 ? (float)reinterpret_cast<const _Float16*>(a.res)[mo * a.res_cs + a.res_co + n] : a.res[mo * a.res_cs + a.res_co + n];
                r = a.res_mask ? (rv > 0.f ? r : 0.f) : r + rv;
            }
            const float ov = fd_act(r, n >= a.act_c0 ? a.act : FD_ACT_NONE, prm);
            if (a.y16) reinterpret_cast<_Float16*>(a.y)[mo * a.y_cs + a.y_co + n] = (_Float16)ov;
            else a.y[mo * a.y_cs + a.y_co + n] = ov;
        }
    }
}

int fd_launch_splitk_reduce(const ConvArgs& orig, const float* ws, int nslice, int ldw, long slab, hipStream_t stream) {
    long g = ((long)orig.M * (ldw >> 2) + 255) / 256;
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)g), dim3(256), 0, stream, orig, ws, nslice, ldw, slab);
    FD_CHECK_LAUNCH("fd_conv2d (split-K reduce)");
    return FD_OK;
}

template <int WGM, int WGN, int TM, int TN, bool STEM, bool SB = false, int TAG = 0, bool SPLIT = false, bool GATE = false, bool GNS = false, bool H1 = false, bool GEMM = false,
          bool DUAL = false, bool RUP = false>
static int launch_conv(const ConvArgs& a, hipStream_t stream) {
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    constexpr int lds_ab = (SB ? 1 : 2) * (BM + BN) * 32 * 4;
    constexpr int NT = WGM * WGN * 64;
    constexpr int lds_epi = (NT / 64) * 4096;              // the epilogue stages one 4 KiB sub-tile per wave
    constexpr int lds = lds_ab > lds_epi ? lds_ab : lds_epi;
    ConvArgs b = a;
    b.mtiles = (a.M + BM - 1) / BM;
    b.ntiles = (a.Cout + BN - 1) / BN;
    auto kern = conv_igemm_kernel<WGM, WGN, TM, TN, STEM, SB, TAG, SPLIT, GATE, GNS, H1, GEMM, DUAL, RUP>;
    static std::atomic<unsigned> attr_mask{0};       // per kernel instantiation, one bit per device
    fd_set_max_lds_once(attr_mask, reinterpret_cast<const void*>(kern), lds);
    hipLaunchKernelGGL(kern, dim3(b.mtiles * b.ntiles, (a.KT + a.kt_per - 1) / a.kt_per), dim3(NT), lds, stream, b);
    FD_CHECK_LAUNCH("fd_conv2d_nhwc_f32");
    return FD_OK;
}

static int dispatch_conv(const fd_conv_params* p, ConvArgs& a, bool stem, hipStream_t stream);

extern "C" int32_t fd_conv_workgroups(const fd_conv_params* p) {
    FD_REQUIRE(p && fd_segs_ok(&p->in) && p->dil >= 1 && p->Cout >= 1, FD_E_INVAL, "fd_conv_workgroups: bad arguments");
    FD_REQUIRE(p->tile == FD_TILE_WINOGRAD4, FD_E_UNSUPPORTED, "fd_conv_workgroups: only FD_TILE_WINOGRAD4 launches can be sliced");
    return fd_wino4_workgroups(p);
}

extern "C" int32_t fd_conv_workgroups_live(const fd_conv_params* p) {
    FD_REQUIRE(p && fd_segs_ok(&p->in) && p->dil >= 1 && p->Cout >= 1, FD_E_INVAL, "fd_conv_workgroups_live: bad arguments");
    FD_REQUIRE(p->tile == FD_TILE_WINOGRAD4, FD_E_UNSUPPORTED, "fd_conv_workgroups_live: only FD_TILE_WINOGRAD4 launches can be sliced");
    const int all = fd_wino4_workgroups(p);
    FD_REQUIRE(p->wg_count <= 0 || (p->wg_first >= 0 && (long)p->wg_first + p->wg_count <= all), FD_E_INVAL, "fd_conv_workgroups_live: slice outside the grid");
    return p->wg_count > 0 ? fd_wino4_workgroups_live(p, p->wg_first, p->wg_count) : fd_wino4_workgroups_live(p, 0, all);
}

extern "C" int64_t fd_conv_workspace_bytes(int64_t out_rows, int32_t Cout, int32_t ksplit) {
    if (out_rows < 1 || Cout < 1 || ksplit < 1) return -1;
    return ksplit > 1 ? (int64_t)ksplit * out_rows * ((Cout + 3) & ~3) * 4 : 0;
}

extern "C" int32_t fd_conv2d_nhwc_f32(const fd_conv_params* p, fd_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    FD_REQUIRE(p && p->x && p->w && p->y, FD_E_INVAL, "fd_conv2d: null pointer");
    FD_REQUIRE(fd_segs_ok(&p->in), FD_E_INVAL, "fd_conv2d: bad segment table");
    const bool stem = p->mode == FD_CONV_STEM;
    FD_REQUIRE(p->Cout >= 1 && p->KH >= 1 && p->KW >= 1 && p->stride >= 1 && p->dil >= 1 && p->pad >= 0, FD_E_INVAL,
               "fd_conv2d: bad geometry");
    if (stem) {
        FD_REQUIRE(p->KH == 7 && p->KW == 7 && p->stride == 2 && p->pad == 3 && p->dil == 1 && p->in.nseg == 1 &&
                       p->x_cs == 4 && p->x_co == 0,
                   FD_E_INVAL, "fd_conv2d: stem mode needs 7x7 s2 p3 on an [N][H][W][4] input");
    } else {
        FD_REQUIRE(p->Cin >= 4 && p->Cin % 4 == 0, FD_E_UNSUPPORTED, "fd_conv2d: Cin=%d must be a multiple of 4", p->Cin);
        FD_REQUIRE(p->x_cs % 4 == 0 && p->x_co % 4 == 0 && p->x_cs >= p->x_co + p->Cin, FD_E_INVAL,
                   "fd_conv2d: input channel view (cs=%d co=%d Cin=%d) must be 4-aligned and in range", p->x_cs, p->x_co, p->Cin);
    }
    FD_REQUIRE((((uintptr_t)p->x | (uintptr_t)p->w) & 15) == 0, FD_E_INVAL, "fd_conv2d: x / w not 16-byte aligned");
    FD_REQUIRE(p->y_cs >= p->y_co + p->Cout, FD_E_INVAL, "fd_conv2d: output channel view out of range");
    FD_REQUIRE(!p->res || p->res_cs >= p->res_co + p->Cout, FD_E_INVAL, "fd_conv2d: residual channel view out of range");
    if (p->in.nseg > 1)
        FD_REQUIRE(p->stride == 1 && 2 * p->pad == p->dil * (p->KH - 1) && p->KH == p->KW, FD_E_INVAL,
                   "fd_conv2d: multi-level input needs stride 1 and 'same' padding");

    FD_REQUIRE(p->res_mode >= 0 && p->res_mode <= 2, FD_E_INVAL, "fd_conv2d: res_mode %d", p->res_mode);
    if (p->res && p->res_mode == 2) {
        // The half-resolution addend has M / 4 rows: every kernel that is NOT an RUP instantiation would read `res` at all M output rows (out of bounds).  The whole
        // precondition therefore sits HERE, in front of every early return below (gate, x2, split-K, f16 / f16x3, gn_stats, tag 1 and the stem never reach an RUP kernel).
        FD_REQUIRE(p->tile == FD_TILE_AUTO || p->tile == FD_TILE_64x64 || p->tile == FD_TILE_128x64_SB || p->tile == FD_TILE_64x128_SB, FD_E_UNSUPPORTED,
                   "fd_conv2d: res_mode 2 is built for tiles 64x64, 128x64_SB, 64x128_SB (got %d)", p->tile);
        FD_REQUIRE(!stem && p->KH == 1 && p->KW == 1 && p->stride == 1 && p->pad == 0 && p->in.nseg == 1 && p->Cin % 32 == 0 && p->precision == FD_PREC_F32 &&
                       p->ksplit <= 1 && p->sc_H <= 0 && p->out_H <= 0 && p->out_W <= 0 && !p->gate && !p->gn_stats && !p->x2 && p->tag != 1 &&
                       p->in.H[0] % 2 == 0 && p->in.W[0] % 2 == 0 && p->Cout % 4 == 0 && p->y_cs % 4 == 0 && p->y_co % 4 == 0 && ((uintptr_t)p->y & 15) == 0 &&
                       p->res_cs % 4 == 0 && p->res_co % 4 == 0 && ((uintptr_t)p->res & 15) == 0,
                   FD_E_UNSUPPORTED, "fd_conv2d: res_mode 2 (half-resolution addend) needs an fp32 1x1 stride-1 unpadded single-level conv with even H, W, Cin %% 32 == 0, "
                                     "16-byte views, no split-K / scatter / gate / gn_stats / x2 / tag 1");
    }
    FD_REQUIRE(p->wg_count <= 0 || p->tile == FD_TILE_WINOGRAD4, FD_E_UNSUPPORTED, "fd_conv2d: wg_first / wg_count (a slice of the layer's grid) exist for FD_TILE_WINOGRAD4 only");
    FD_REQUIRE(p->sk_wgs <= 0 || p->tile == FD_TILE_WINOGRAD4, FD_E_UNSUPPORTED, "fd_conv2d: sk_wgs (the persistent stream-K form) exists for FD_TILE_WINOGRAD4 only");
    FD_REQUIRE(!p->io_f16 || (p->tile != FD_TILE_WINOGRAD && p->tile != FD_TILE_WINOGRAD4 && p->tile != FD_TILE_NARROW), FD_E_UNSUPPORTED,
               "fd_conv2d: io_f16 (f16 activation maps) is built for the FD_PREC_F16 implicit-GEMM tiles only");
    if (p->tile == FD_TILE_WINOGRAD) return fd_launch_conv_wino(p, stream);   // own argument block, own weight packing
    if (p->tile == FD_TILE_WINOGRAD4) return fd_launch_conv_wino4(p, stream);
    if (p->tile == FD_TILE_NARROW) return fd_launch_conv_narrow(p, stream);

    ConvArgs a;
    a.x16 = a.y16 = a.res16 = 0;
    a.x = p->x; a.w = p->w; a.scale = p->scale; a.shift = p->shift; a.res = p->res; a.y = p->y;
    a.x_cs = p->x_cs; a.x_co = p->x_co; a.res_cs = p->res_cs; a.res_co = p->res_co; a.y_cs = p->y_cs; a.y_co = p->y_co;
    a.Cin = p->Cin; a.Cout = p->Cout; a.KW = p->KW; a.stride = p->stride; a.pad = p->pad; a.dil = p->dil;
    a.act = p->act; a.act_c0 = p->act_c0;
    a.res_mask = (p->res && p->res_mode == 1) ? 1 : 0;
    a.res_up = 0;
    a.nseg = p->in.nseg;
    long mo = 0;
    for (int s = 0; s < FD_MAX_SEG; ++s) {
        if (s < p->in.nseg) {
            a.H[s] = p->in.H[s]; a.W[s] = p->in.W[s];
            a.Ho[s] = (p->in.H[s] + 2 * p->pad - p->dil * (p->KH - 1) - 1) / p->stride + 1;
            a.Wo[s] = (p->in.W[s] + 2 * p->pad - p->dil * (p->KW - 1) - 1) / p->stride + 1;
            if (p->out_H > 0 || p->out_W > 0) {   // explicit output size: taps past the input read as zero (parity classes of a strided dgrad)
                FD_REQUIRE(p->in.nseg == 1 && p->out_H > 0 && p->out_W > 0, FD_E_INVAL, "fd_conv2d: out_H / out_W need a single-level input");
                a.Ho[s] = p->out_H; a.Wo[s] = p->out_W;
            }
            FD_REQUIRE(a.Ho[s] >= 1 && a.Wo[s] >= 1, FD_E_INVAL, "fd_conv2d: empty output");
            a.m_in[s] = p->in.m_start[s];
            a.m_out[s] = (int)mo;
            mo += (long)p->in.batch * a.Ho[s] * a.Wo[s];
        } else {
            a.H[s] = a.W[s] = a.Ho[s] = a.Wo[s] = 1; a.m_in[s] = 0; a.m_out[s] = (int)mo;
        }
        a.seg_param[s] = p->seg_param[s];
    }
    a.m_out[FD_MAX_SEG] = (int)mo;
    FD_REQUIRE(mo > 0 && mo < (1L << 31), FD_E_INVAL, "fd_conv2d: row count out of range");
    a.M = (int)mo;
    a.sc_on = 0; a.sc_sy = a.sc_sx = 1; a.sc_oy = a.sc_ox = 0; a.sc_H = a.sc_W = 0;
    long y_rows = mo;
    if (p->sc_H > 0) {
        FD_REQUIRE(p->in.nseg == 1 && !stem && p->sc_W > 0 && p->sc_sy >= 1 && p->sc_sx >= 1 && p->sc_oy >= 0 && p->sc_ox >= 0 &&
                       p->sc_sy * (a.Ho[0] - 1) + p->sc_oy < p->sc_H && p->sc_sx * (a.Wo[0] - 1) + p->sc_ox < p->sc_W,
                   FD_E_INVAL, "fd_conv2d: output scatter does not fit its %d x %d target", p->sc_H, p->sc_W);
        a.sc_on = 1; a.sc_sy = p->sc_sy; a.sc_sx = p->sc_sx; a.sc_oy = p->sc_oy; a.sc_ox = p->sc_ox; a.sc_H = p->sc_H; a.sc_W = p->sc_W;
        y_rows = (long)p->in.batch * p->sc_H * p->sc_W;
    }
    FD_REQUIRE((long)p->in.m_start[p->in.nseg] * p->x_cs < (1L << 31) && y_rows * p->y_cs < (1L << 31), FD_E_UNSUPPORTED,
               "fd_conv2d: tensor exceeds 2^31 elements");
    if (stem) { a.KT = 7; a.ntaps = 7; a.Kpacked = 7 * 32; }
    else { const int cch = (p->Cin + 31) / 32; a.ntaps = p->KH * p->KW; a.KT = a.ntaps * cch; a.Kpacked = a.ntaps * cch * 32; }
    a.mtiles = a.ntiles = 0;

    a.vec_epi = (p->Cout % 4 == 0 && p->y_cs % 4 == 0 && p->y_co % 4 == 0 && ((uintptr_t)p->y & 15) == 0 &&
                 (!p->res || (p->res_cs % 4 == 0 && p->res_co % 4 == 0 && ((uintptr_t)p->res & 15) == 0)))
                    ? 1 : 0;

    a.is_gemm = (!stem && p->KH == 1 && p->KW == 1 && p->stride == 1 && p->pad == 0) ? 1 : 0;
    {
        const long xb = (long)p->in.m_start[p->in.nseg] * p->x_cs * 4, wb = (long)p->Cout * a.Kpacked * 4;
        FD_REQUIRE(xb < 0xC0000000L && wb < 0xC0000000L, FD_E_UNSUPPORTED, "fd_conv2d: input / weight buffer exceeds 3 GiB");
        a.x_bytes = (unsigned)xb; a.w_bytes = (unsigned)wb;
        for (int sg = 0; sg < p->in.nseg; ++sg)
            FD_REQUIRE((long)p->in.W[sg] * p->x_cs * 4 < (1L << 23), FD_E_UNSUPPORTED, "fd_conv2d: image row of %ld bytes exceeds the 24-bit row-stride path", (long)p->in.W[sg] * p->x_cs * 4);
    }
    a.Cout_epi = a.Cout; a.kt_per = a.KT; a.slice_stride = 0;
    a.p_halo = 0;
    a.x2 = nullptr; a.x2_cs = a.x2_co = 0; a.x2_stride = 1; a.x2_H = a.x2_W = 1; a.kt2 = 0; a.x2_bytes = 0;
    if (p->x2) {         // K-concatenated second source: a bottleneck's conv3 and its downsample conv as one GEMM
        FD_REQUIRE(!stem && p->KH == 1 && p->KW == 1 && p->pad == 0 && p->stride == 1 && p->in.nseg == 1 && p->Cin % 32 == 0 && p->x2_Cin >= 32 && p->x2_Cin % 32 == 0 &&
                       p->precision == FD_PREC_F32 && p->ksplit <= 1 && !a.sc_on && p->out_H <= 0 && !p->gate && !p->gn_stats && p->tag != 1,
                   FD_E_UNSUPPORTED, "fd_conv2d: `x2` needs an fp32 1x1 stride-1 unpadded single-level conv with Cin, x2_Cin multiples of 32, no split-K / scatter / gate / gn_stats");
        FD_REQUIRE(p->x2_stride >= 1 && p->x2_cs % 4 == 0 && p->x2_co % 4 == 0 && p->x2_cs >= p->x2_co + p->x2_Cin && ((uintptr_t)p->x2 & 15) == 0 &&
                       (p->in.H[0] - 1) * p->x2_stride < p->x2_H && (p->in.W[0] - 1) * p->x2_stride < p->x2_W,
                   FD_E_INVAL, "fd_conv2d: `x2` view must be 16-byte addressable and cover the strided samples of a %d x %d output", p->in.H[0], p->in.W[0]);
        const long x2b = (long)p->in.batch * p->x2_H * p->x2_W * p->x2_cs * 4;
        FD_REQUIRE(x2b < 0xC0000000L, FD_E_UNSUPPORTED, "fd_conv2d: `x2` buffer exceeds 3 GiB");
        a.x2 = p->x2; a.x2_cs = p->x2_cs; a.x2_co = p->x2_co; a.x2_stride = p->x2_stride; a.x2_H = p->x2_H; a.x2_W = p->x2_W;
        a.kt2 = p->Cin / 32; a.x2_bytes = (unsigned)x2b;
        a.KT = (p->Cin + p->x2_Cin) / 32; a.Kpacked = a.KT * 32; a.kt_per = a.KT;
        a.w_bytes = (unsigned)((long)p->Cout * a.Kpacked * 4);
        switch (p->tile) {      // (the single-buffer GEMM-addressed tiles the bottleneck expansions use)
            case FD_TILE_64x128_SB: return launch_conv<2, 2, 1, 2, false, true, 0, false, false, false, false, true, true>(a, stream);
            case FD_TILE_64x64: return launch_conv<2, 2, 1, 1, false, false, 0, false, false, false, false, true, true>(a, stream);
            case FD_TILE_128x128_SB: return launch_conv<2, 2, 2, 2, false, true, 0, false, false, false, false, true, true>(a, stream);
            case FD_TILE_AUTO: case FD_TILE_128x64_SB: return launch_conv<2, 2, 2, 1, false, true, 0, false, false, false, false, true, true>(a, stream);
            default: fd_set_error("fd_conv2d: `x2` is built for tiles 64x64, 128x64_SB, 64x128_SB, 128x128_SB (got %d)", p->tile); return FD_E_UNSUPPORTED;
        }
    }
    if (p->io_f16) {     // AMP activations stored as f16 (train.py:175-181 autocast): the single-plane f16 kernels read / write them directly
        FD_REQUIRE(p->precision == FD_PREC_F16 && !stem && !p->gate && !p->gn_stats && !p->x2 && (p->io_f16 & ~7) == 0 && p->tile != FD_TILE_WAVE64 && p->tile != FD_TILE_128x128_PATCH,
                   FD_E_UNSUPPORTED, "fd_conv2d: io_f16 (f16 activation maps) needs FD_PREC_F16 on the plain conv tiles, no gate / gn_stats / x2");
        FD_REQUIRE(a.vec_epi || !(p->io_f16 & 6), FD_E_UNSUPPORTED, "fd_conv2d: f16 output / residual maps need 4-channel aligned views (Cout, y_cs, y_co, res_cs, res_co multiples of 4)");
        FD_REQUIRE(!(p->io_f16 & 4) || p->res, FD_E_INVAL, "fd_conv2d: io_f16 names an f16 residual but res is NULL");
        a.x16 = p->io_f16 & 1; a.y16 = (p->io_f16 >> 1) & 1; a.res16 = (p->io_f16 >> 2) & 1;
        if (a.x16) a.x_bytes = (unsigned)((long)p->in.m_start[p->in.nseg] * p->x_cs * 2);
    }
    if (p->tile == FD_TILE_F16K64) return fd_launch_conv_f16k64(p, a, stream);      // the AMP step's f16 kernel on K-tiles of 64 channels (own weight packing)
    a.gate = nullptr; a.gate_cs = 0; a.gate_hw = 1; a.gate_b = nullptr; a.gate_act = FD_ACT_NONE; a.gate_batch = p->in.batch;
    a.gn_stats = nullptr; a.gn_G = 1; a.gn_cg = 4;
    if (p->gn_stats) {   // row-group statistics of the stored output (GroupNorm fused into the producer)
        FD_REQUIRE(p->gn_groups >= 1 && p->Cout % p->gn_groups == 0 && (p->Cout / p->gn_groups) % 4 == 0 && 32 % (p->Cout / p->gn_groups) == 0 &&
                       p->Cout % 32 == 0 && a.vec_epi && !a.sc_on && p->ksplit <= 1 && p->precision == FD_PREC_F32 && ((uintptr_t)p->gn_stats & 7) == 0,
                   FD_E_UNSUPPORTED, "fd_conv2d: gn_stats needs fp32, Cout %% 32 == 0, 4 | Cout / groups | 32, 16-byte output views, no split-K / scatter");
        a.gn_stats = p->gn_stats; a.gn_G = p->gn_groups; a.gn_cg = p->Cout / p->gn_groups;
    }
    if (p->gate) {       // input gate folded into the loader (MBConv: squeeze-excitation gate -> project conv; HISFCOSHead: GroupNorm + SiLU -> pw2)
        FD_REQUIRE(a.is_gemm && p->precision == FD_PREC_F32 && p->ksplit <= 1 && !a.sc_on && p->out_H <= 0 && !p->gn_stats, FD_E_UNSUPPORTED,
                   "fd_conv2d: `gate` needs an fp32 1x1 stride-1 unpadded conv, no split-K / scatter / gn_stats");
        FD_REQUIRE(p->gate_cs >= p->Cin && p->gate_cs % 4 == 0 && ((uintptr_t)p->gate & 15) == 0, FD_E_INVAL, "fd_conv2d: gate rows must be 16-byte aligned, gate_cs >= Cin");
        FD_REQUIRE(!p->gate_b || ((((uintptr_t)p->gate_b) & 15) == 0 && (p->gate_act == FD_ACT_NONE || p->gate_act == FD_ACT_RELU || p->gate_act == FD_ACT_SILU)),
                   FD_E_INVAL, "fd_conv2d: gate_b must be 16-byte aligned, gate_act in {NONE, RELU, SILU}");
        a.gate = p->gate; a.gate_cs = p->gate_cs; a.gate_hw = p->in.H[0] * p->in.W[0];
        a.gate_b = p->gate_b; a.gate_act = p->gate_b ? p->gate_act : FD_ACT_NONE;
        // the GroupNorm-fused form (HISFCOSHead pw2): single-LDS-buffer tiles at three workgroups per CU, as the tuned table picks for the plain layer
        if (p->gate_b && a.Cout > 64 && (long)((a.M + 63) / 64) * ((a.Cout + 127) / 128) >= 512) {   // 64 x 128 SB
            if (p->Cin % 32 == 0 && !a.sc_on) return launch_conv<2, 2, 1, 2, false, true, 0, false, true, false, false, true>(a, stream);      // (GEMM-addressed loader)
            return launch_conv<2, 2, 1, 2, false, true, 0, false, true>(a, stream);
        }
        if (a.Cout <= 32) return launch_conv<4, 1, 1, 1, false, false, 0, false, true>(a, stream);      // 128 x 32
        const long m128 = (a.M + 127) / 128;
        if (a.Cout <= 64) {
            if (m128 < 512) return launch_conv<2, 2, 1, 1, false, false, 0, false, true>(a, stream);    // 64 x 64: small maps need more workgroups
            return launch_conv<2, 2, 2, 1, false, false, 0, false, true>(a, stream);                    // 128 x 64
        }
        if (a.Cout <= 96 && m128 >= 512) return launch_conv<4, 1, 1, 3, false, false, 0, false, true>(a, stream);      // 128 x 96
        if (m128 * ((a.Cout + 127) / 128) < 640) return launch_conv<2, 2, 1, 2, false, false, 0, false, true>(a, stream);   // 64 x 128
        return launch_conv<2, 2, 2, 2, false, false, 0, false, true>(a, stream);                        // 128 x 128
    }
    if (p->tile == FD_TILE_WAVE64) {     // GEMM-addressed layers as wave-autonomous 64 x 64 tiles (fd_conv_wave.hip)
        FD_REQUIRE(a.is_gemm && p->Cin % 32 == 0 && p->Cout % 32 == 0 && p->precision == FD_PREC_F32 && p->ksplit <= 1 && !a.sc_on && p->out_H <= 0 &&
                       a.vec_epi && !p->gate && p->act_c0 % 32 == 0 && (p->act == FD_ACT_NONE || p->act == FD_ACT_RELU || p->act == FD_ACT_SILU),
                   FD_E_UNSUPPORTED, "fd_conv2d: FD_TILE_WAVE64 needs an fp32 1x1 stride-1 unpadded conv with Cin, Cout, act_c0 multiples of 32, 16-byte output "
                   "views, ReLU / SiLU / no activation, no split-K / scatter / gate");
        FD_REQUIRE(p->w_frag && ((uintptr_t)p->w_frag & 15) == 0, FD_E_UNSUPPORTED, "fd_conv2d: FD_TILE_WAVE64 needs fd_conv_params.w_frag (fd_pack_conv_weight_wave_f32)");
        return fd_launch_conv_wave(a, p->w_frag, stream);
    }
    if (p->tile == FD_TILE_128x128_PATCH) {     // 3x3 stride-1 'same' conv with the input patch staged in LDS (fd_conv_patch.hip)
        int wmax = 0;
        for (int sg = 0; sg < p->in.nseg; ++sg) wmax = p->in.W[sg] > wmax ? p->in.W[sg] : wmax;
        FD_REQUIRE(!stem && p->KH == 3 && p->KW == 3 && p->stride == 1 && p->pad == p->dil && p->ksplit <= 1 && !a.sc_on && !p->gn_stats &&
                       p->out_H <= 0 && FD_PATCH_BM + 2 * p->dil * (wmax + 1) <= FD_PATCH_MAXROWS,
                   FD_E_UNSUPPORTED, "fd_conv2d: FD_TILE_128x128_PATCH needs a 3x3 stride-1 'same' conv with 128 + 2*dil*(W+1) <= %d rows (W=%d dil=%d)",
                   FD_PATCH_MAXROWS, wmax, p->dil);
        a.p_halo = p->dil * (wmax + 1);
        return fd_launch_conv_patch(a, p->tag == 1, p->precision == FD_PREC_F16X3, stream);
    }

    const int ksplit = p->ksplit > 1 ? p->ksplit : 1;
    if (ksplit > 1) {
        FD_REQUIRE(!stem && ksplit <= 64 && a.KT >= ksplit, FD_E_INVAL, "fd_conv2d: ksplit=%d needs 1 < ksplit <= min(64, K-tiles=%d)", ksplit, a.KT);
        const ConvArgs orig = a;
        const int ldw = (a.Cout + 3) & ~3;
        const long slab = (long)a.M * ldw;
        FD_REQUIRE(p->workspace && ((uintptr_t)p->workspace & 15) == 0 && p->workspace_bytes >= (int64_t)ksplit * slab * 4, FD_E_INVAL,
                   "fd_conv2d: split-K needs a 16-byte aligned workspace of fd_conv_workspace_bytes() bytes");
        a.kt_per = (a.KT + ksplit - 1) / ksplit;
        a.y = (float*)p->workspace; a.y_cs = ldw; a.y_co = 0; a.slice_stride = slab; a.Cout_epi = ldw;
        a.scale = a.shift = a.res = nullptr; a.act = FD_ACT_NONE; a.vec_epi = 1; a.sc_on = 0;
        a.y16 = a.res16 = 0;                         // (the slabs are fp32; the combine launch writes the f16 map)
        const int rc = dispatch_conv(p, a, stem, stream);
        if (rc != FD_OK) return rc;
        return fd_launch_splitk_reduce(orig, (const float*)p->workspace, (a.KT + a.kt_per - 1) / a.kt_per, ldw, slab, stream);
    }
    return dispatch_conv(p, a, stem, stream);
}

static int dispatch_conv(const fd_conv_params* p, ConvArgs& a, bool stem, hipStream_t stream) {

    if (stem) { FD_REQUIRE(!a.gn_stats, FD_E_UNSUPPORTED, "fd_conv2d: gn_stats on the stem"); return launch_conv<2, 2, 2, 1, true>(a, stream); }       // 128 x 64
    if (p->precision == FD_PREC_F16X3) {   // split-f16: 3 f16 MFMAs per fp32 product (weights pre-split by the caller)
        FD_REQUIRE(!stem, FD_E_UNSUPPORTED, "fd_conv2d: the stem runs in exact fp32 only");
        const bool tg = p->tag == 1;
        switch (p->tile) {
            case FD_TILE_AUTO:
                if (a.Cout <= 32) return launch_conv<4, 1, 1, 1, false, false, 0, true>(a, stream);
                if (a.Cout <= 64) return launch_conv<2, 2, 1, 1, false, false, 0, true>(a, stream);
                if (a.Cout <= 96) return launch_conv<4, 1, 1, 3, false, false, 0, true>(a, stream);
                return launch_conv<2, 2, 1, 2, false, false, 0, true>(a, stream);
            case FD_TILE_128x128:
                return tg ? launch_conv<2, 2, 2, 2, false, false, 1, true>(a, stream) : launch_conv<2, 2, 2, 2, false, false, 0, true>(a, stream);
            case FD_TILE_128x128_SB:
                return tg ? launch_conv<2, 2, 2, 2, false, true, 1, true>(a, stream) : launch_conv<2, 2, 2, 2, false, true, 0, true>(a, stream);
            case FD_TILE_128x64: return launch_conv<2, 2, 2, 1, false, false, 0, true>(a, stream);
            case FD_TILE_64x128: return launch_conv<2, 2, 1, 2, false, false, 0, true>(a, stream);
            case FD_TILE_64x64: return launch_conv<2, 2, 1, 1, false, false, 0, true>(a, stream);
            case FD_TILE_128x64_SB: return launch_conv<2, 2, 2, 1, false, true, 0, true>(a, stream);
            case FD_TILE_64x128_SB: return launch_conv<2, 2, 1, 2, false, true, 0, true>(a, stream);
            case FD_TILE_128x32: return launch_conv<4, 1, 1, 1, false, false, 0, true>(a, stream);
            case FD_TILE_128x96: return launch_conv<4, 1, 1, 3, false, false, 0, true>(a, stream);
            case FD_TILE_128x96_SB: return launch_conv<4, 1, 1, 3, false, true, 0, true>(a, stream);
            case FD_TILE_256x128:
                return tg ? launch_conv<4, 2, 2, 2, false, false, 1, true>(a, stream) : launch_conv<4, 2, 2, 2, false, false, 0, true>(a, stream);
            case FD_TILE_256x128_SB:
                return tg ? launch_conv<4, 2, 2, 2, false, true, 1, true>(a, stream) : launch_conv<4, 2, 2, 2, false, true, 0, true>(a, stream);
            default: fd_set_error("fd_conv2d: tile id %d has no split-f16 kernel", p->tile); return FD_E_UNSUPPORTED;
        }
    }
    if (p->precision == FD_PREC_F16) {   // single-plane f16 products, fp32 accumulation (torch.autocast arithmetic); weights in the FD_PREC_F16X3 packing (hi plane used)
        FD_REQUIRE(!a.gn_stats && !a.gate, FD_E_UNSUPPORTED, "fd_conv2d: FD_PREC_F16 has no gn_stats / gate form");
        auto blocks = [&](int bm, int bn) { return (long)((a.M + bm - 1) / bm) * ((a.Cout + bn - 1) / bn); };
        switch (p->tile) {
            case FD_TILE_AUTO:
                if (a.Cout <= 32) return launch_conv<4, 1, 1, 1, false, false, 0, true, false, false, true>(a, stream);
                if (a.Cout <= 64) return blocks(128, 64) >= 512 ? launch_conv<2, 2, 2, 1, false, false, 0, true, false, false, true>(a, stream)
                                                                : launch_conv<2, 2, 1, 1, false, false, 0, true, false, false, true>(a, stream);
                if (a.Cout <= 96) return launch_conv<4, 1, 1, 3, false, false, 0, true, false, false, true>(a, stream);
                if (blocks(128, 128) >= 512) return launch_conv<2, 2, 2, 2, false, false, 0, true, false, false, true>(a, stream);
                if (blocks(64, 128) >= 256) return launch_conv<2, 2, 1, 2, false, false, 0, true, false, false, true>(a, stream);
                return launch_conv<2, 2, 1, 1, false, false, 0, true, false, false, true>(a, stream);
            case FD_TILE_128x128: return launch_conv<2, 2, 2, 2, false, false, 0, true, false, false, true>(a, stream);
            case FD_TILE_128x128_SB: return launch_conv<2, 2, 2, 2, false, true, 0, true, false, false, true>(a, stream);
            case FD_TILE_128x64: return launch_conv<2, 2, 2, 1, false, false, 0, true, false, false, true>(a, stream);
            case FD_TILE_64x128: return launch_conv<2, 2, 1, 2, false, false, 0, true, false, false, true>(a, stream);
            case FD_TILE_64x64: return launch_conv<2, 2, 1, 1, false, false, 0, true, false, false, true>(a, stream);
            case FD_TILE_128x32: return launch_conv<4, 1, 1, 1, false, false, 0, true, false, false, true>(a, stream);
            case FD_TILE_128x96: return launch_conv<4, 1, 1, 3, false, false, 0, true, false, false, true>(a, stream);
            default: fd_set_error("fd_conv2d: tile id %d has no f16 kernel", p->tile); return FD_E_UNSUPPORTED;
        }
    }
    FD_REQUIRE(p->precision == FD_PREC_F32, FD_E_INVAL, "fd_conv2d: unknown precision %d", p->precision);
    if (p->tag == 1) {  // profiling tag: same code under its own kernel symbol (TAG = 1) so rocprofv3 --stats isolates it
        switch (p->tile) {
            case FD_TILE_128x128: return launch_conv<2, 2, 2, 2, false, false, 1>(a, stream);
            case FD_TILE_128x128_SB: return launch_conv<2, 2, 2, 2, false, true, 1>(a, stream);
            case FD_TILE_64x128_SB: return launch_conv<2, 2, 1, 2, false, true, 1>(a, stream);
            case FD_TILE_128x64_SB: return launch_conv<2, 2, 2, 1, false, true, 1>(a, stream);
            default: break;  // other tiles keep the shared symbol
        }
    }
    if (a.gn_stats) {   // row-group statistics in the epilogue: dedicated instantiations of the one- / two-sub-tile tiles
        if (a.is_gemm && p->Cin % 32 == 0 && !a.sc_on) {     // (GEMM-addressed loader for the tiles the fused head uses)
            switch (p->tile) {
                case FD_TILE_64x128_SB: return launch_conv<2, 2, 1, 2, false, true, 0, false, false, true, false, true>(a, stream);
                case FD_TILE_AUTO: case FD_TILE_128x64_SB: return launch_conv<2, 2, 2, 1, false, true, 0, false, false, true, false, true>(a, stream);
                default: break;
            }
        }
        switch (p->tile) {
            case FD_TILE_64x64: return launch_conv<2, 2, 1, 1, false, false, 0, false, false, true>(a, stream);
            case FD_TILE_128x64: return launch_conv<2, 2, 2, 1, false, false, 0, false, false, true>(a, stream);
            case FD_TILE_64x128: return launch_conv<2, 2, 1, 2, false, false, 0, false, false, true>(a, stream);
            case FD_TILE_64x128_SB: return launch_conv<2, 2, 1, 2, false, true, 0, false, false, true>(a, stream);
            case FD_TILE_AUTO: case FD_TILE_128x64_SB: return launch_conv<2, 2, 2, 1, false, true, 0, false, false, true>(a, stream);
            default: fd_set_error("fd_conv2d: gn_stats is built for tiles 64x64, 128x64(_SB), 64x128(_SB), WAVE64 and WINOGRAD (got %d)", p->tile); return FD_E_UNSUPPORTED;
        }
    }
    if (p->res && p->res_mode == 2) {
        // the upsampled addend of an FPN lateral: 1x1 stride-1 unpadded fp32 layers on one level with even H, W, vector epilogue, no split-K
        FD_REQUIRE(a.is_gemm && p->in.nseg == 1 && p->Cin % 32 == 0 && p->precision == FD_PREC_F32 && p->ksplit <= 1 && !a.sc_on && p->out_H <= 0 && !p->gate &&
                       !p->gn_stats && !p->x2 && a.vec_epi && p->tag != 1 && p->in.H[0] % 2 == 0 && p->in.W[0] % 2 == 0,
                   FD_E_UNSUPPORTED, "fd_conv2d: res_mode 2 (half-resolution addend) needs an fp32 1x1 stride-1 unpadded single-level conv with even H, W, Cin %% 32 == 0, "
                                     "16-byte views, no split-K / scatter / gate / gn_stats / x2");
        FD_REQUIRE((long)p->in.batch * (p->in.H[0] / 2) * (p->in.W[0] / 2) * p->res_cs < (1L << 31), FD_E_UNSUPPORTED, "fd_conv2d: residual exceeds 2^31 elements");
        a.res_up = 1;
        switch (p->tile) {
            case FD_TILE_64x64: return launch_conv<2, 2, 1, 1, false, false, 0, false, false, false, false, true, false, true>(a, stream);
            case FD_TILE_128x64_SB: return launch_conv<2, 2, 2, 1, false, true, 0, false, false, false, false, true, false, true>(a, stream);
            case FD_TILE_AUTO: case FD_TILE_64x128_SB: return launch_conv<2, 2, 1, 2, false, true, 0, false, false, false, false, true, false, true>(a, stream);
            default: fd_set_error("fd_conv2d: res_mode 2 is built for tiles 64x64, 128x64_SB, 64x128_SB (got %d)", p->tile); return FD_E_UNSUPPORTED;
        }
    }
    // GEMM-addressed fp32 layers (1x1, stride 1, no padding, Cin % 32 == 0, no gate): the loader compiled without the tap / bounds arithmetic
    static const int gemm_on = getenv("FD_CONV_GEMM") ? atoi(getenv("FD_CONV_GEMM")) : 1;
    const bool pointwise = !stem && p->KH == 1 && p->KW == 1 && p->pad == 0;     // any stride: one input address per output row
    if (gemm_on && pointwise && p->Cin % 32 == 0 && !a.gate && !a.sc_on) {
        switch (p->tile) {
            case FD_TILE_128x128: return launch_conv<2, 2, 2, 2, false, false, 0, false, false, false, false, true>(a, stream);
            case FD_TILE_128x64: return launch_conv<2, 2, 2, 1, false, false, 0, false, false, false, false, true>(a, stream);
            case FD_TILE_64x128: return launch_conv<2, 2, 1, 2, false, false, 0, false, false, false, false, true>(a, stream);
            case FD_TILE_64x64: return launch_conv<2, 2, 1, 1, false, false, 0, false, false, false, false, true>(a, stream);
            case FD_TILE_128x128_SB: return launch_conv<2, 2, 2, 2, false, true, 0, false, false, false, false, true>(a, stream);
            case FD_TILE_128x64_SB: return launch_conv<2, 2, 2, 1, false, true, 0, false, false, false, false, true>(a, stream);
            case FD_TILE_64x128_SB: return launch_conv<2, 2, 1, 2, false, true, 0, false, false, false, false, true>(a, stream);
            default: break;
        }
    }
    switch (p->tile) {
        case FD_TILE_AUTO: break;
        case FD_TILE_128x128: return launch_conv<2, 2, 2, 2, false>(a, stream);
        case FD_TILE_128x64: return launch_conv<2, 2, 2, 1, false>(a, stream);
        case FD_TILE_64x128: return launch_conv<2, 2, 1, 2, false>(a, stream);
        case FD_TILE_64x64: return launch_conv<2, 2, 1, 1, false>(a, stream);
        case FD_TILE_128x32: return launch_conv<4, 1, 1, 1, false>(a, stream);
        case FD_TILE_128x96: return launch_conv<4, 1, 1, 3, false>(a, stream);
        case FD_TILE_128x96_SB: return launch_conv<4, 1, 1, 3, false, true>(a, stream);
        case FD_TILE_128x128_SB: return launch_conv<2, 2, 2, 2, false, true>(a, stream);
        case FD_TILE_128x64_SB: return launch_conv<2, 2, 2, 1, false, true>(a, stream);
        case FD_TILE_64x128_SB: return launch_conv<2, 2, 1, 2, false, true>(a, stream);
        case FD_TILE_256x128: return launch_conv<4, 2, 2, 2, false, false>(a, stream);
        case FD_TILE_256x128_SB: return launch_conv<4, 2, 2, 2, false, true>(a, stream);
        default: fd_set_error("fd_conv2d: unknown tile id %d", p->tile); return FD_E_INVAL;
    }
    if (a.Cout <= 32) return launch_conv<4, 1, 1, 1, false>(a, stream);   // 128 x 32
    // Largest tile that still yields >= 2 workgroups per CU (256 CUs); tiny maps fall through to 64 x 64.
    auto blocks = [&](int bm, int bn) { return (long)((a.M + bm - 1) / bm) * ((a.Cout + bn - 1) / bn); };
    const long want = 384;
    if (a.Cout > 64 && a.Cout <= 96 && blocks(128, 96) >= want) return launch_conv<4, 1, 1, 3, false>(a, stream);  // 128 x 96
    if (a.Cout > 64 && blocks(128, 128) >= want) return launch_conv<2, 2, 2, 2, false>(a, stream);
    if (blocks(128, 64) >= want || a.Cout <= 64) {
        if (a.Cout <= 64 && blocks(128, 64) < want) return launch_conv<2, 2, 1, 1, false>(a, stream);
        return launch_conv<2, 2, 2, 1, false>(a, stream);             // 128 x 64
    }
    return launch_conv<2, 2, 1, 1, false>(a, stream);                 // 64 x 64
}
