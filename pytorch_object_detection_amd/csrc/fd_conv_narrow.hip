// fd_conv_narrow.hip -- 3x3 stride-1 pad-1 convolution with at most 8 output channels, exact fp32 on the VECTOR unit (FD_TILE_NARROW).
//
// HISFCOS / FCOS predict centre-ness and the four box distances with one 3x3 conv of 1 + 4 = 5 output channels over the whole pyramid
// (cnt_logits + reg_pred, HISFcos.py:207-209,227-228; Fcos.py:113-117): 3.1 GFLOP at 16 x 640 x 640 behind a 140 MB input.  On the matrix
// pipe that layer pays for a 32-wide (MFMA) or 64-wide (Winograd workgroup) output tile whatever its width: the F(4x4) kernel ran a full
// 256-channel input transform for ONE live 32-cout block (0.19 ms, 16 TFLOP/s).  The fp32 FMA rate of the vector unit equals the fp32 MFMA
// rate on gfx950 (64 FLOP / clk / SIMD), so a layer this narrow belongs there: no padding of N, no transforms.
//
//   * one 256-thread workgroup = one 16 x 16 output tile of one (level, image), one thread per pixel, NCO accumulators in registers;
//   * the 18 x 18 x 16-channel input patch is staged in LDS per 16-channel chunk ([pixel][20 floats]: 16-lane groups of a ds_read_b128 walk
//     16 consecutive pixels at a 20-float pitch = 16 different bank quads), the next chunk's patch travels global -> registers under the FMAs;
//   * weights are wave-uniform.  Through the scalar cache (s_load + an SGPR operand per FMA) every (r, c4, q) step waited out a scalar-cache
//     round trip: 0.18 ms, no better than the Winograd kernel.  Now a chunk's weights sit in LDS as [step][k = 4 channels of the quad][8 couts]
//     and lane i of every QUAD reads the couts of channel k = i (one or two LDS reads per step, four distinct addresses per wave); the FMA takes
//     them through a DPP quad_perm:[k,k,k,k] broadcast: v_fmac_f32_dpp acc[co], W[co] (lane k of the quad), v.k -- no VGPR per weight, no
//     scalar round trips, three LDS instructions per 20 FMAs;
//   * per output one fp32 fma chain over (chunk, r, c4, q, component): a direct convolution's arithmetic (no Winograd rounding);
//   * pyramids are one launch (per-tile level decode); epilogue = scale / shift (bias) + activation on channels >= act_c0 (ScaleExp);
//   * GATE instantiations: the preceding GroupNorm's per-(level, image, channel) affine + activation (fd_conv_params.gate / gate_b / gate_act, the
//     coef of fd_groupnorm_from_rowstats) is applied to the patch on its way from the load registers to LDS -- padding stays zero, as a padded
//     conv of the normalised map has it -- so the normalise pass over the predictor's half of the tower map disappears.
// 2 560 waves of ~11 500 FMAs at batch 16: see DESIGN 4.1f for the measurement.
#include "fd_conv_common.h"

struct NarrowArgs {
    const float* x; const float* w; const float* scale; const float* shift; float* y;
    int x_cs, x_co, y_cs, y_co;
    int Cout, act, act_c0;
    int NCH;                      // 16-channel chunks
    int nseg, batch;
    int H[FD_MAX_SEG], W[FD_MAX_SEG], TH[FD_MAX_SEG], TW[FD_MAX_SEG];   // TH x TW tiles of 16 x 16 outputs per image
    int m0[FD_MAX_SEG];           // first row of the level
    int t0[FD_MAX_SEG + 1];       // first tile of the level
    float seg_param[FD_MAX_SEG];
    const float* gate; const float* gate_b;    // GATE: [nseg * batch][gate_cs] floats each, channel 0 = the view's first input channel
    int gate_cs, gate_act;
};

#define NR_T 16                    // tile edge
#define NR_P (NR_T + 2)            // patch edge
#define NR_KC 16                   // channels per chunk
#define NR_PITCH 20                // floats per patch pixel in LDS (16 + 4: conflict-free 16-byte reads at a one-pixel lane stride)
#define NR_NLD ((NR_P * NR_P * (NR_KC / 4) + 255) / 256)   // float4 loads per thread and chunk (6)

#define NR_WSTEP 32                // floats per (r, c4, q) step of the weight image: [4 k][8 couts]
#define NR_WCHUNK (36 * NR_WSTEP)  // floats per 16-channel chunk (4.5 KB)

// acc += w[lane k of this lane's quad] * v   (k = 0 .. 3): the weight operand is another lane's register, broadcast inside the quad by DPP
template <int K>
__device__ __forceinline__ void nr_fmac_quad(float& acc, float w, float v) {
    if constexpr (K == 0) asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(w), "v"(v));
    if constexpr (K == 1) asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(w), "v"(v));
    if constexpr (K == 2) asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(w), "v"(v));
    if constexpr (K == 3) asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(w), "v"(v));
}

template <int NCO, bool GATE>
__global__ __launch_bounds__(256) void conv3x3_narrow_kernel(NarrowArgs a) {
    __shared__ __attribute__((aligned(16))) float patch[NR_P * NR_P * NR_PITCH];
    __shared__ __attribute__((aligned(16))) float wts[NR_WCHUNK];
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    // ---- which tile ----
    int t = blockIdx.x, s = 0;
#pragma unroll
    for (int i = 1; i < FD_MAX_SEG; ++i)
        if (i < a.nseg && t >= a.t0[i]) s = i;
    t -= a.t0[s];
    const int H = a.H[s], W = a.W[s], tpi = a.TH[s] * a.TW[s];
    const int n = t / tpi, r_ = t - n * tpi;
    const int ti = r_ / a.TW[s], tj = r_ - ti * a.TW[s];
    const int h0 = ti * NR_T, w0 = tj * NR_T;
    const long row0 = (long)a.m0[s] + (long)n * H * W;

    // ---- patch loader: float4 i = tid + 256 u of the [324 pixels][4 quads] patch; weight loader: float4 tid (+ 256) of the chunk's 288 ----
    const float* src[NR_NLD];
    int dst[NR_NLD];
#pragma unroll
    for (int u = 0; u < NR_NLD; ++u) {
        const int i = tid + 256 * u, px = i >> 2, c4 = i & 3;
        const int pr = px / NR_P, pc = px - pr * NR_P;
        const int h = h0 - 1 + pr, w = w0 - 1 + pc;
        const bool ok = px < NR_P * NR_P && (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
        src[u] = ok ? a.x + (row0 + (long)h * W + w) * a.x_cs + a.x_co + c4 * 4 : nullptr;
        dst[u] = px < NR_P * NR_P ? px * NR_PITCH + c4 * 4 : -1;
    }
    float4 pf[NR_NLD], wf[2];
    // GATE: every float4 of this thread has channel quad tid & 3 (256 % 4 == 0): one affine pair per chunk
    float4 ga = make_float4(1.f, 1.f, 1.f, 1.f), gb = make_float4(0.f, 0.f, 0.f, 0.f);
    const long grow = GATE ? ((long)s * a.batch + n) * a.gate_cs + (tid & 3) * 4 : 0;
    const float4* wsrc = reinterpret_cast<const float4*>(a.w);
    auto load = [&](int ch) {
#pragma unroll
        for (int u = 0; u < NR_NLD; ++u)
            pf[u] = src[u] ? *reinterpret_cast<const float4*>(src[u] + ch * NR_KC) : make_float4(0.f, 0.f, 0.f, 0.f);
        wf[0] = wsrc[(long)ch * (NR_WCHUNK / 4) + tid];
        if (tid < NR_WCHUNK / 4 - 256) wf[1] = wsrc[(long)ch * (NR_WCHUNK / 4) + 256 + tid];
        if constexpr (GATE) {
            ga = *reinterpret_cast<const float4*>(a.gate + grow + ch * NR_KC);
            gb = *reinterpret_cast<const float4*>(a.gate_b + grow + ch * NR_KC);
        }
    };
    auto store = [&]() {
        if constexpr (GATE) {       // applied here, not in load(): the loads are still in flight under the previous chunk's FMAs there
#pragma clang fp contract(off)
            // (multiply, then add, each rounded -- contraction is off in this block: the arithmetic of the normalise pass this replaces, fd_layers.hip is
            //  built without FMA contraction.)  One uniform branch on the activation per chunk, not one per element.
            auto affine = [&](auto act_fn) {
#pragma unroll
                for (int u = 0; u < NR_NLD; ++u)
                    if (src[u]) {       // (padding pixels stay zero: the reference pads the NORMALISED map)
                        pf[u].x = act_fn(pf[u].x * ga.x + gb.x); pf[u].y = act_fn(pf[u].y * ga.y + gb.y);
                        pf[u].z = act_fn(pf[u].z * ga.z + gb.z); pf[u].w = act_fn(pf[u].w * ga.w + gb.w);
                    }
            };
            if (a.gate_act == FD_ACT_RELU) affine([](float v) { return fd_act(v, FD_ACT_RELU, 0.f); });
            else if (a.gate_act == FD_ACT_SILU) affine([](float v) { return fd_act(v, FD_ACT_SILU, 0.f); });
            else affine([](float v) { return v; });
        }
#pragma unroll
        for (int u = 0; u < NR_NLD; ++u)
            if (dst[u] >= 0) *reinterpret_cast<float4*>(patch + dst[u]) = pf[u];
        reinterpret_cast<float4*>(wts)[tid] = wf[0];
        if (tid < NR_WCHUNK / 4 - 256) reinterpret_cast<float4*>(wts)[256 + tid] = wf[1];
    };

    float acc[NCO];
#pragma unroll
    for (int co = 0; co < NCO; ++co) acc[co] = 0.f;
    const float* my = patch + (ty * NR_P + tx) * NR_PITCH;
    const float* myw = wts + (tid & 3) * 8;            // this lane's row of every step's [4 k][8 couts] block: the couts of channel k = lane & 3

    load(0);
    for (int ch = 0; ch < a.NCH; ++ch) {
        __syncthreads();                       // everyone has finished reading the previous chunk's patch and weights
        store();
        __syncthreads();
        if (ch + 1 < a.NCH) load(ch + 1);      // the next chunk's patch travels under this chunk's FMAs
        {
            // 36 steps (r, c4, q), software-pipelined by hand: the operands of step st + 1 are requested from LDS before the FMAs of step st go out
            // (the FMAs are asm statements: left alone the compiler reads a step's operands right in front of them and waits out the LDS latency 36 times)
            float4 va[2], wa[2], wb_[2];
            float w4_[2];
            auto rd = [&](int st, int slot) {
                const int r = st / 12, c4 = (st / 3) & 3, q = st % 3;
                va[slot] = *reinterpret_cast<const float4*>(my + (r * NR_P + q) * NR_PITCH + c4 * 4);
                const float* wk = myw + st * NR_WSTEP;
                wa[slot] = *reinterpret_cast<const float4*>(wk);
                if constexpr (NCO == 5) w4_[slot] = wk[4];
                if constexpr (NCO > 5) wb_[slot] = *reinterpret_cast<const float4*>(wk + 4);
            };
            rd(0, 0);
#pragma unroll
            for (int st = 0; st < 36; ++st) {
                const int sl = st & 1;
                if (st + 1 < 36) rd(st + 1, sl ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                const float4 v = va[sl];
                float wr[8];
                wr[0] = wa[sl].x; wr[1] = wa[sl].y; wr[2] = wa[sl].z; wr[3] = wa[sl].w;
                if constexpr (NCO == 5) wr[4] = w4_[sl];
                if constexpr (NCO > 5) { wr[4] = wb_[sl].x; wr[5] = wb_[sl].y; wr[6] = wb_[sl].z; wr[7] = wb_[sl].w; }
                asm volatile("s_nop 1");     // (gfx9: a VALU write of a register needs two wait states before a DPP read of it; the compiler does not see into the asm below)
#pragma unroll
                for (int co = 0; co < NCO; ++co) nr_fmac_quad<0>(acc[co], wr[co], v.x);
#pragma unroll
                for (int co = 0; co < NCO; ++co) nr_fmac_quad<1>(acc[co], wr[co], v.y);
#pragma unroll
                for (int co = 0; co < NCO; ++co) nr_fmac_quad<2>(acc[co], wr[co], v.z);
#pragma unroll
                for (int co = 0; co < NCO; ++co) nr_fmac_quad<3>(acc[co], wr[co], v.w);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // ---- epilogue ----
    const int h = h0 + ty, w = w0 + tx;
    if (h >= H || w >= W) return;
    float* yp = a.y + (row0 + (long)h * W + w) * a.y_cs + a.y_co;
    const float prm = a.seg_param[s];
#pragma unroll
    for (int co = 0; co < NCO; ++co) {
        if (co < a.Cout) {
            float v = acc[co];
            if (a.scale) v *= a.scale[co];
            if (a.shift) v += a.shift[co];
            if (a.act != FD_ACT_NONE && co >= a.act_c0) v = fd_act(v, a.act, prm);
            yp[co] = v;
        }
    }
}

extern "C" int32_t fd_conv_narrow_nco(int32_t Cout) {     // accumulators per pixel: the weight image itself is always 8 couts wide
    return Cout < 1 || Cout > 8 ? -1 : (Cout <= 4 ? 4 : (Cout == 5 ? 5 : 8));
}

int fd_launch_conv_narrow(const fd_conv_params* p, hipStream_t stream) {
    FD_REQUIRE(p->KH == 3 && p->KW == 3 && p->stride == 1 && p->dil == 1 && p->pad == 1 && p->Cout >= 1 && p->Cout <= 8 && p->Cin % NR_KC == 0 &&
                   p->precision == FD_PREC_F32 && !p->res && !p->gn_stats && p->ksplit <= 1 && p->out_H <= 0 && p->sc_H <= 0,
               FD_E_UNSUPPORTED, "fd_conv2d: FD_TILE_NARROW needs an fp32 3x3 stride-1 pad-1 conv with Cout <= 8, Cin %% 16 == 0, no residual / gn_stats / split-K / scatter");
    FD_REQUIRE(p->x_cs % 4 == 0 && p->x_co % 4 == 0, FD_E_UNSUPPORTED, "fd_conv2d: FD_TILE_NARROW needs a 16-byte addressable input view");
    FD_REQUIRE(!p->gate || (p->gate_b && p->gate_cs >= p->Cin && p->gate_cs % 4 == 0 && ((uintptr_t)p->gate & 15) == 0 && ((uintptr_t)p->gate_b & 15) == 0 &&
                            (p->gate_act == FD_ACT_NONE || p->gate_act == FD_ACT_RELU || p->gate_act == FD_ACT_SILU)),
               FD_E_INVAL, "fd_conv2d: FD_TILE_NARROW takes `gate` only with `gate_b` (a GroupNorm's affine), 16-byte aligned rows, gate_cs >= Cin, gate_act in {NONE, RELU, SILU}");
    NarrowArgs a;
    a.x = p->x; a.w = p->w; a.scale = p->scale; a.shift = p->shift; a.y = p->y;
    a.x_cs = p->x_cs; a.x_co = p->x_co; a.y_cs = p->y_cs; a.y_co = p->y_co;
    a.Cout = p->Cout; a.act = p->act; a.act_c0 = p->act_c0;
    a.NCH = p->Cin / NR_KC;
    a.gate = p->gate; a.gate_b = p->gate_b; a.gate_cs = p->gate_cs; a.gate_act = p->gate_act;
    a.nseg = p->in.nseg; a.batch = p->in.batch;
    long t = 0;
    for (int s = 0; s < FD_MAX_SEG; ++s) {
        a.t0[s] = (int)t;
        if (s < p->in.nseg) {
            a.H[s] = p->in.H[s]; a.W[s] = p->in.W[s];
            a.TH[s] = (p->in.H[s] + NR_T - 1) / NR_T; a.TW[s] = (p->in.W[s] + NR_T - 1) / NR_T;
            a.m0[s] = p->in.m_start[s];
            t += (long)p->in.batch * a.TH[s] * a.TW[s];
        } else {
            a.H[s] = a.W[s] = a.TH[s] = a.TW[s] = 1; a.m0[s] = 0;
        }
        a.seg_param[s] = p->seg_param[s];
    }
    a.t0[FD_MAX_SEG] = (int)t;
    FD_REQUIRE(t > 0 && t < (1L << 31), FD_E_INVAL, "fd_conv2d: tile count out of range");
    const long rows = p->in.m_start[p->in.nseg];
    FD_REQUIRE(rows * p->x_cs < (1L << 31) && rows * p->y_cs < (1L << 31), FD_E_UNSUPPORTED, "fd_conv2d: tensor exceeds 2^31 elements");
    const dim3 grid((unsigned)t), block(256);
    const int nco = fd_conv_narrow_nco(p->Cout);
    if (p->gate) {
        switch (nco) {
            case 4: hipLaunchKernelGGL((conv3x3_narrow_kernel<4, true>), grid, block, 0, stream, a); break;
            case 5: hipLaunchKernelGGL((conv3x3_narrow_kernel<5, true>), grid, block, 0, stream, a); break;
            default: hipLaunchKernelGGL((conv3x3_narrow_kernel<8, true>), grid, block, 0, stream, a); break;
        }
    } else {
        switch (nco) {
            case 4: hipLaunchKernelGGL((conv3x3_narrow_kernel<4, false>), grid, block, 0, stream, a); break;
            case 5: hipLaunchKernelGGL((conv3x3_narrow_kernel<5, false>), grid, block, 0, stream, a); break;
            default: hipLaunchKernelGGL((conv3x3_narrow_kernel<8, false>), grid, block, 0, stream, a); break;
        }
    }
    FD_CHECK_LAUNCH("fd_conv2d_nhwc_f32 (narrow 3x3 on the vector unit)");
    return FD_OK;
}
