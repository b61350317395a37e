// fd_conv_wino4.hip -- 3x3 stride-1 'same' convolution (dilation 1 or 2) as Winograd F(4x4, 3x3) on the fp32 MFMA of gfx950 (FD_TILE_WINOGRAD4).
//
// F(2x2, 3x3) (fd_conv_wino.hip) already runs its MFMA loop at the matrix pipe's full rate: in exact fp32 the only lever left is executing
// fewer multiplies again.  F(4x4, 3x3) computes a 4x4 output tile from a 6x6 input patch with 36 multiplies per (cin, cout) -- 2.25 per
// output instead of 4 (F(2x2)) or 9 (direct):
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A,   B^T 6x6, G 6x3, A^T 4x6 (Lavin & Gray's matrices, interpolation points 0, +-1, +-2, inf)
// so the conv becomes 36 independent GEMMs  M_f[tile][cout] = sum_c V_f[tile][c] * U_f[cout][c].  Everything stays fp32; the transforms now
// multiply by 4, 5, 8 as well, which costs a factor ~2 in rounding error against F(2x2): the whole 640 x 640 model stays within 2e-5 * (1 + |x|)
// of its fp64 evaluation (tools/wino44_emul.py), inside the 1e-4 parity bar.
//
// One workgroup (8 waves at 256 registers, one workgroup per CU) = 32 tiles (512 output pixels) x 64 output channels x all 36 frequencies:
//   * wave (g, ch) owns frequencies 9 g .. 9 g + 8 and output channels 32 ch .. + 31: 9 accumulators of 32 x 32 (144 VGPRs); the four waves of a
//     channel block that lies past Cout keep their loader role and skip the MFMAs;
//   * the input transform is a three-stage software pipeline over 8-channel chunks, ONE workgroup barrier per chunk, its stages cut into slices that
//     sit between the nine MFMA groups of the chunk (loader waves 0 .. 5: one patch line per WAVE, (tile, channel quad) per lane):
//       G(c)  fetches the 6 pixels of its patch row (6 x 16-byte raw buffer loads, zero outside the image);
//       R(c)  row pass of B^T d B in registers, written to an LDS scratch [tile][quad] blocks of [j'][i] (+ 1 float4 of padding: conflict-free);
//       C(c)  the same thread, now (tile, quad, column j'), reads its column of 6, does the column pass and writes V[f = 6 i' + j'][tile][8 c];
//       M(c)  the MFMAs.   Iteration cc runs M(cc), C(cc+1), R(cc+2), G(cc+3).
//     LDS: V 2 x 36 KB + scratch 2 x 37 KB = 146 KB; the epilogue reuses it for the frequency planes;
//   * U goes global -> registers in MFMA layout (packed [cout/32][chunk][36 f][32 cout][8 c], 1 KB per frequency block), the block of chunk
//     cc+1 into the registers the MFMAs of chunk cc have just consumed; every global / LDS address is a per-lane register set up once plus a scalar
//     or immediate offset per instruction (no address arithmetic on the VALU inside the loop);
//   * dilation 2 = four parity classes per image (address arithmetic only); split-K = blockIdx.y slices of the chunk loop, raw partial outputs to the
//     workspace, the direct kernel's combine launch finishes (fd_launch_splitk_reduce);
//   * epilogue: the two halves of the 32 tiles in turn -- all waves put that half of their accumulators into LDS planes, then every thread gathers the
//     36 frequencies of (tile, 4 couts), applies A^T . A for two of the four output rows and y = act(v * scale + shift (+ | mask) res) and stores
//     16 bytes per pixel.
// DESIGN 4.1d has the measurements (head tower 1.39 -> 0.92 ms against F(2x2)) and what did and did not matter on the way there.
#include "fd_conv_common.h"
#include <type_traits>
#include <vector>
#include <cstdio>

struct Wino4Args {
    const float* x; const float* u; const float* scale; const float* shift; const float* res; float* y;
    int x_cs, x_co, res_cs, res_co, y_cs, y_co;
    int Cin, Cout, act, act_c0, res_mask;
    int NC;                       // 8-channel chunks
    int nc_per;                   // chunks per split-K slice (blockIdx.y = slice; NC when split-K is off)
    long slice_stride;            // floats between the slices' raw partial outputs in the workspace (split-K), else 0
    int dil;                      // 1, or 2: four parity classes (ph, pw) per image, each a dense conv on the sub-grid h = 2 hs + ph, w = 2 ws + pw
    int nseg;
    int H[FD_MAX_SEG], W[FD_MAX_SEG], TH[FD_MAX_SEG], TW[FD_MAX_SEG];   // TH x TW tiles of 4 x 4 outputs
    int m0[FD_MAX_SEG];           // first row of the level
    int t0[FD_MAX_SEG + 1];       // first tile of the level
    float seg_param[FD_MAX_SEG];
    int T;                        // tiles in all
    int mtiles, ntiles, mt_per;   // M tiles (32 tiles each), N tiles (64 cout), M tiles per XCD
    int blk0;                     // first workgroup of this launch in the layer's grid (fd_conv_params.wg_first; a multiple of 8: the XCD of a workgroup is unchanged)
    unsigned x_bytes, u_bytes;
    // persistent stream-K form (SK kernels; fd_conv_params.sk_wgs): 8 * wpx workgroups, each walks a contiguous range of its XCD's (item, chunk) units
    int wpx;                      // workgroups per XCD
    int sk_P;                     // pieces a remainder item is queued as
    float4* ws;                   // partial outputs, one slot of [16][512] float4 per workgroup
    int* flags;                   // [1024] one per slot (1 = complete), [8] the XCDs' queue heads: all zero before and after every launch
    int dbg;                      // timing builds only (-DFD_W4_TIMING + FD_W4_DBG): 1 = no loader stages, 2 = no MFMAs, 4 = no epilogue (wrong results)
    long long* ts;                // timing builds only (FD_W4_TS=<file>): per workgroup, the 100 MHz wall clock at entry / set-up done / prologue done / chunk loop done / end
};
// The shipped library never skips parts of the kernel: the timing switches exist only in a build compiled with -DFD_W4_TIMING (tools/pmc_wino.sh).
#ifdef FD_W4_TIMING
#define W4_DBG(a) ((a).dbg)
// phase clocks: k = 0 entry, 1 .. 3 ACCUMULATE the time since the previous call over the segments of the workgroup (set-up | prologue | chunk loop), 4 = end: the record
// holds entry, entry + set-up, ... + prologue, ... + loop, end -- the rest up to `end` is epilogue (+ flag waits, segment barriers)
#define W4_TS(a, k) do { const long long n_ = wall_clock64(); if ((k) == 0) w4_t0 = n_; else if ((k) < 4) w4_acc[(k) - 1] += n_ - w4_last; w4_last = n_; \
        if ((k) == 4 && (a).ts && threadIdx.x == 0) { long long* r_ = (a).ts + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8; r_[0] = w4_t0; r_[1] = w4_t0 + w4_acc[0]; \
            r_[2] = r_[1] + w4_acc[1]; r_[3] = r_[2] + w4_acc[2]; r_[4] = n_; r_[6] = w4_segs; } } while (0)
#else
#define W4_DBG(a) 0
#define W4_TS(a, k) do { } while (0)
#endif

// workspace header of the persistent form -- FIXED, whatever the grid: launches with different sk_wgs may share one workspace (with a header sized by sk_wgs, the slots
// of a 240-workgroup launch lay over the queue heads of the 256-workgroup layout: garbage claims, a memory fault on the first mixed use)
#define W4_SK_MAX_WGS 1024
#define W4_TB 32
#define W4_KC 8
#define W4_PLANE (W4_TB * W4_KC)            // floats per frequency plane of V
#define W4_STAGE (36 * W4_PLANE)            // floats per V stage (36 KB)
#define W4_SBLK 37                          // float4s per (tile, channel quad) block of the scratch: 36 + 1, odd, so that 64 lanes = 64 blocks never share a bank
#define W4_SSTAGE (64 * W4_SBLK * 4)        // floats per scratch stage (37 KB)

struct Tile4 { int s, n, h0, w0, ph, pw; bool ok; };     // h0, w0: first output of the tile in sub-grid coordinates

__device__ __forceinline__ Tile4 wino4_decode(const Wino4Args& a, int t) {
    Tile4 p;
    p.ok = t < a.T;
    if (!p.ok) t = 0;
    int s = 0;
#pragma unroll
    for (int i = 1; i < FD_MAX_SEG; ++i)
        if (i < a.nseg && t >= a.t0[i]) s = i;
    const int local = t - a.t0[s];
    const int tpc = a.TH[s] * a.TW[s], tpi = tpc * a.dil * a.dil;
    const int n = local / tpi, rc = local - n * tpi;
    const int cls = rc / tpc, r = rc - cls * tpc;
    const int ti = r / a.TW[s], tj = r - ti * a.TW[s];
    p.s = s; p.n = n; p.h0 = 4 * ti; p.w0 = 4 * tj;
    p.ph = cls / a.dil; p.pw = cls - p.ph * a.dil;
    return p;
}

// B^T d for one line of six (per component of a float4): t = B^T d,
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
// in place: six inputs -> six outputs per component, four temporaries.  (Tried: the same on float2 halves with inline constants only --
// 4 d0 - 5 d2 + d4 as 4 (d0 - d2) + (d4 - d2) -- 26 fewer VALU operations, no faster, one more rounding per output: dropped.)
__device__ __forceinline__ void w4_bt_inplace(float4 (&d)[6]) {
#define W4_BT1(c)                                                              \
    {                                                                          \
        const float e42 = d[4].c - 4.f * d[2].c, e31 = d[3].c - 4.f * d[1].c;  \
        const float f42 = d[4].c - d[2].c, f31 = 2.f * (d[3].c - d[1].c);      \
        const float t0 = 4.f * d[0].c - 5.f * d[2].c + d[4].c;                 \
        const float t5 = 4.f * d[1].c - 5.f * d[3].c + d[5].c;                 \
        d[0].c = t0;                                                           \
        d[1].c = e42 + e31;                                                    \
        d[2].c = e42 - e31;                                                    \
        d[3].c = f42 + f31;                                                    \
        d[4].c = f42 - f31;                                                    \
        d[5].c = t5;                                                           \
    }
    W4_BT1(x) W4_BT1(y) W4_BT1(z) W4_BT1(w)
#undef W4_BT1
}

__device__ __forceinline__ void w4_bt(const float4 (&d)[6], float4 (&t)[6]) {
#define W4_BT1(c)                                                              \
    {                                                                          \
        const float e42 = d[4].c - 4.f * d[2].c, e31 = d[3].c - 4.f * d[1].c;  \
        const float f42 = d[4].c - d[2].c, f31 = 2.f * (d[3].c - d[1].c);      \
        t[0].c = 4.f * d[0].c - 5.f * d[2].c + d[4].c;                         \
        t[1].c = e42 + e31;                                                    \
        t[2].c = e42 - e31;                                                    \
        t[3].c = f42 + f31;                                                    \
        t[4].c = f42 - f31;                                                    \
        t[5].c = 4.f * d[1].c - 5.f * d[3].c + d[5].c;                         \
    }
    W4_BT1(x) W4_BT1(y) W4_BT1(z) W4_BT1(w)
#undef W4_BT1
}

// SK = false: one workgroup per (M tile, N tile) item (and split-K slice), grid = items.
// SK = true: a PERSISTENT grid of 8 * wpx workgroups (at most one per CU) that CLAIM their work from one queue per XCD (an atomic counter in the workspace):
//   * the queue of an XCD holds its `items` (M tile, N tile) items in the order of the plain launch (so the workgroups of an XCD always work on a window of neighbouring
//     items: same 64-cout slice of U, neighbouring input rows -- static contiguous ranges per workgroup put all N tiles' U slices, 19 MB on the head tower, through a 4 MB
//     L2 and were no faster than the plain launch; static round-robin lost the dynamic balance of the hardware dispatcher: CUs differ by ~5 % in speed);
//   * the last items % wpx items -- the round of the plain launch that leaves most of the chip idle -- are queued as P PIECES each (P = sk_P, chosen by the host so that
//     the pieces fill the chip once): piece pp = chunks [pp * NC / P, (pp + 1) * NC / P).  Pieces pp > 0 are PARTS: output-transformed, unscaled accumulators to the
//     workspace slot (item, pp), then a flag; piece 0 is queued LAST of its item and FINISHES it: waits for the item's flags, adds the slots in chunk order, applies the
//     epilogue.  A part never waits, and when piece 0 is claimed every part of its item has been claimed by a running workgroup: the grid drains whatever the dispatch
//     order or the number of CUs it got.  Which workgroup computes what varies from run to run; WHAT is computed, and the order of every sum, does not: deterministic, and
//     bit-identical to the plain launch for every item that is not cut.
//   * the next claim is issued at the start of an epilogue and read at its end: no atomic round trip between two items.
template <int TAG, bool SK>
__global__ __launch_bounds__(512, 1) void conv3x3_wino4_kernel(Wino4Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Vs = reinterpret_cast<float*>(smem);                  // [2][36][32][8]
    float* Ss = Vs + 2 * W4_STAGE;                               // [2][32 tiles][2 q] blocks of W4_SBLK float4: [6 j'][6 i] + pad
#ifdef FD_W4_TIMING
    long long w4_t0 = 0, w4_last = 0, w4_acc[3] = {0, 0, 0}, w4_segs = 0;
#endif
    W4_TS(a, 0);

    // XCD-aware order (as fd_conv_wino.hip): XCD x owns M tiles [x * mt_per, (x + 1) * mt_per) and walks them cout tile by cout tile
    const int bx = blockIdx.x + a.blk0;
    const int xcd = bx & 7, jx = bx >> 3;
    // (SK: the M tiles are dealt out evenly, mtiles / 8 or one more per XCD -- the plain grid's last XCD share is up to 7 tiles short, and the SK form's pieces are sized for all XCDs at once)
    const int mt_lo = SK ? xcd * (a.mtiles >> 3) + min(xcd, a.mtiles & 7) : xcd * a.mt_per;
    const int cnt = SK ? (a.mtiles >> 3) + (xcd < (a.mtiles & 7) ? 1 : 0) : min(a.mtiles - mt_lo, a.mt_per);
    // SK: the XCD's queue = sk_W whole items, then the pieces of the remaining items; u = the unit this workgroup holds
    int sk_W = 0, sk_units = 0, u = 0;
    int* const sk_ctr = a.flags + (SK ? W4_SK_MAX_WGS + xcd : 0);
    int* const claim_lds = reinterpret_cast<int*>(smem + (2 * W4_STAGE + 2 * W4_SSTAGE) * 4);
    if constexpr (SK) {
        const int items = max(cnt, 0) * a.ntiles;
        const int rem = items % a.wpx;
        sk_W = items - rem; sk_units = sk_W + rem * a.sk_P;
        // the first unit of workgroup j is unit j (no atomic before the first prologue); claim v of the queue is unit wpx + v.  Every workgroup that got a unit makes
        // exactly one claim that fails, so the queue sees exactly `sk_units` claims: the one that draws v = sk_units - 1 zeroes the counter for the next launch.
        u = jx;
        if (u >= sk_units) return;
    } else {
        if (cnt <= 0 || jx >= cnt * a.ntiles) return;
    }

    constexpr unsigned OOB = 0xC0000000u;
    // The input resource starts ONE patch pixel before the tensor: the six columns of a patch row are then column j = 1's byte offset (per lane, or OOB)
    // plus the non-negative scalar offset j * px_b + chunk -- no VALU per load.  (Nothing below a.x is touched: column 0 of a row at w0 = 0 is masked.)
    const int px_b = a.x_cs * 4 * a.dil;                         // bytes between neighbouring patch pixels of one row
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.x - px_b), (short)0, (int)(a.x_bytes + (unsigned)px_b), 0x00020000);
    const __amdgpu_buffer_rsrc_t ursrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.u, (short)0, (int)a.u_bytes, 0x00020000);
    // SK: the part slots.  Every access to them (and to the flags) is a DEVICE-scope access (sc1: served at the memory side of the L2s) -- the hand-over between two
    // workgroups then needs no cache-wide operation: an agent-scope release / acquire FENCE writes back and invalidates the whole L2 of the XCD, and with it the weights and
    // input rows all its workgroups re-use (measured: every layer 10 - 50 % slower than the plain launch).
    constexpr int SC1 = 16;                                      // gfx940+ cache policy: scope bit 1 = agent scope
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.ws, (short)0, SK ? (int)(gridDim.x * (16 * 512 * 16)) : 0, 0x00020000);

  for (;;) {                                                     // SK: the segments of this workgroup's range; else exactly one pass
    // Every per-thread role constant is derived HERE, from an opaque copy of the thread index: nothing but scalars is carried from one segment into the next, so the
    // epilogue (the most register-hungry part: 72 live accumulator registers beside the output pass) keeps no address of the next segment's loader alive.
    int tid = threadIdx.x;
    if constexpr (SK) asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int g = wave & 3, ch = wave >> 2;
    const int l31 = lane & 31, lh = lane >> 5;
    // ---- loader role (threads 0 .. 383): (tile lt, line pr, channel quad q); pr = patch row i in G / R, column j' in C ----
    const bool ldr = __builtin_amdgcn_readfirstlane(tid) < 384 && !(W4_DBG(a) & 1);       // wave-uniform: scalar branches around the loader slices
    // one line pr per WAVE, (tile, quad) per lane: every scratch / V instruction of a wave then walks 64 different blocks at one in-block offset
    const int q = tid & 1, lt = (tid >> 1) & 31, pr = min(tid >> 6, 5);
    // scratch: S[lt][q][j'][i] float4 -- R writes element (j', i = pr) for j' = 0..5; C reads (j' = pr, i = 0..5): 6 consecutive float4
    const int s_wr = ((lt * 2 + q) * W4_SBLK + pr) * 4;          // + j' * 24 floats
    const int s_rd = ((lt * 2 + q) * W4_SBLK + pr * 6) * 4;      // + i * 4 floats
    // V[f = 6 i' + j'][tile][8 c], plain: with one patch line per wave both the writes (64 lanes = one plane) and the MFMA reads are conflict-free
    const int v_wr = (pr * W4_TB + lt) * W4_KC + 4 * q;          // + i' * 6 * W4_PLANE.  (A wave's 64 lanes = the 64 float4s of one plane: no bank is hit twice.)
    const int v_half = 4 * lh;
    int idx, c0, NC;                                             // item of the XCD's list, first chunk, chunks of this segment
    bool part = false;                                           // SK: this segment is a PART (its item's head lies in an earlier workgroup)
    int nparts = 0;                                              // SK: parts the following workgroups hold of the item this segment is the head of
    int slot = 0;                                                // SK: this part's slot / the first slot of the item this segment finishes (per XCD)
    if constexpr (SK) {
        if ((unsigned)u >= (unsigned)sk_units) break;            // (unsigned: a queue head that was not zero at launch ends the workgroup instead of addressing tiles that do not exist)
        if (u < sk_W) { idx = u; c0 = 0; NC = a.NC; }            // a whole item
        else {                                                   // piece pp of the remainder item ri (pieces are queued tail first: ... 2, 1, 0)
            const int k = u - sk_W, ri = k / a.sk_P, pp = a.sk_P - 1 - (k - ri * a.sk_P);
            idx = sk_W + ri; c0 = pp * a.NC / a.sk_P; NC = (pp + 1) * a.NC / a.sk_P - c0;
            part = pp > 0;
            slot = ri * (a.sk_P - 1) + (part ? pp - 1 : 0);
            if (!part) nparts = a.sk_P - 1;
        }
    } else {
        idx = jx; c0 = blockIdx.y * a.nc_per;                    // split-K: first chunk of this slice
        NC = min(a.NC, c0 + a.nc_per) - c0;                      // this slice's chunks [c0, c0 + NC)
    }
    const int nt = idx / cnt, mt = mt_lo + (idx - nt * cnt);
    const int tile0 = mt * W4_TB, n0 = nt * 64;

    // patch row pr of tile lt: one base offset per thread; the six columns differ by a uniform pixel stride (buffer soffset) and a validity bit each
    unsigned a_off[6];                                           // a_base where the column is inside the image, OOB elsewhere
    {
        const Tile4 p = wino4_decode(a, tile0 + lt);
        const int H = a.H[p.s], W = a.W[p.s];
        const int hh = (p.h0 - 1 + pr) * a.dil + p.ph;           // (negative exactly when the sub-grid row is)
        const bool row_ok = ldr && p.ok && (unsigned)hh < (unsigned)H;
        const int rowbase = a.m0[p.s] + (p.n * H + hh) * W;
        const unsigned a_base = ((unsigned)(rowbase + p.w0 * a.dil + p.pw) * (unsigned)a.x_cs + (unsigned)(a.x_co + q * 4)) * 4u;    // column j = 1
#pragma unroll
        for (int j = 0; j < 6; ++j) a_off[j] = (row_ok && (unsigned)((p.w0 - 1 + j) * a.dil + p.pw) < (unsigned)W) ? a_base : OOB;
    }

    // ---- MFMA role ----
    const int nb = (n0 >> 5) + ch;
    const bool nb_ok = nb * 32 < ((a.Cout + 31) & ~31);
    const unsigned u_off0 = nb_ok ? ((unsigned)(nb * a.NC) * 36u + 9u * g) * 1024u + (unsigned)(l31 * 32 + lh * 16) : OOB;

    float4 pv[6], bq[9];
    auto load_G = [&](float4 (&dst)[6], int cc) {                 // global -> registers
        const int cb = (c0 + cc) * 32;
#pragma unroll
        for (int j = 0; j < 6; ++j)
            dst[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)a_off[j], j * px_b + cb, 0));
    };
    auto stage_G = [&](int cc) { load_G(pv, cc); };
    auto row_pass = [&](const float4 (&src)[6], int st) {          // row pass (along the patch row) -> scratch
        float4 t[6];
        w4_bt(src, t);
        float* d = Ss + st * W4_SSTAGE + s_wr;
#pragma unroll
        for (int j = 0; j < 6; ++j) *reinterpret_cast<float4*>(d + j * 24) = t[j];
    };
    auto stage_C = [&](int st) {                 // column pass (down column j' = pr) -> V
        float4 s[6], v[6];
        const float* r = Ss + st * W4_SSTAGE + s_rd;
#pragma unroll
        for (int i = 0; i < 6; ++i) s[i] = *reinterpret_cast<const float4*>(r + i * 4);
        w4_bt(s, v);
        float* d = Vs + st * W4_STAGE + v_wr;
#pragma unroll
        for (int i = 0; i < 6; ++i) *reinterpret_cast<float4*>(d + i * 6 * W4_PLANE) = v[i];
    };
    auto load_u = [&](int cc, int fi) {
        // per-lane part in the vector offset (OOB for a block past Cout: zeros), the chunk / frequency part in the scalar offset: no VALU per load
        bq[fi] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ursrc, (int)u_off0, ((c0 + cc) * 36 + fi) * 1024, 0));
    };

    W4_TS(a, 1);
    // ---- prologue: the patch rows of chunks 0, 1, 2 are requested together (ONE global round trip, not three: the accumulators are not live yet,
    // registers are plenty), then R(0) R(1) | C(0): V[0] holds chunk 0, scratch[1] chunk 1's row pass, the patch registers chunk 2 ----
    // (Both paths define every patch register: inside the segment loop of the SK form a register that one path leaves undefined is "the value of the previous
    // segment" to the compiler -- 72 registers carried through the epilogue and spilled.)
    if (ldr) {
        float4 p0[6], p1[6];
        load_G(p0, 0); load_G(p1, min(1, NC - 1)); stage_G(min(2, NC - 1));
#pragma unroll
        for (int fi = 0; fi < 9; ++fi) load_u(0, fi);
        row_pass(p0, 0); row_pass(p1, 1);
    } else {
#pragma unroll
        for (int j = 0; j < 6; ++j) pv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int fi = 0; fi < 9; ++fi) load_u(0, fi);
    }
    __syncthreads();
    if (ldr) stage_C(0);
    __syncthreads();
    W4_TS(a, 2);
    f32x16 acc[9];                                               // (zeroed behind the prologue: its three chunks of patch rows need the registers)
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    // Main loop: one 8-channel chunk per iteration and workgroup barrier.  The loader stages of the NEXT chunks are cut into slices that sit between
    // the nine MFMA groups of this chunk (pinned with sched_barriers): issued in the shadow of the 64-cycle MFMAs instead of in front of them
    // (as one block in front, loader time and MFMA time simply added: 0.34 + 0.90 ms on the head tower).
    //   group 0: R(cc+2) row pass in place in the patch registers   group 1: -> scratch[st]
    //   group 2: C(cc+1) reads its column from scratch[st^1] into the same registers   group 3: column pass in place   group 4: -> V[st^1]
    //   group 5: G(cc+3) issues the patch loads (four groups and a barrier before group 0 of the next chunk consumes them)
    // The loop exists twice, for the loader waves and for the two waves without a loader role: s_waitcnt counts are in-order counts, and where the
    // two paths joined after every slice the compiler had to assume the shorter queue -- the loader waves then waited for their own slice's LDS
    // traffic before every MFMA group.
    auto main_loop = [&](auto ldr_c, auto mm_c) {
    constexpr bool LDR = decltype(ldr_c)::value;
    constexpr bool MM = decltype(mm_c)::value;         // false: this wave's 32-cout block lies past Cout (Cout % 64 in 1..32): no U loads, no MFMAs, only its loader role
    for (int cc = 0; cc < NC; ++cc) {
        const int st = cc & 1;
        const float* Vb = Vs + st * W4_STAGE + (9 * g) * W4_PLANE + v_half;
        const int cn = min(cc + 1, NC - 1);
        float4 fa[2];
        if constexpr (MM) fa[0] = *reinterpret_cast<const float4*>(Vb + (0 * W4_TB + l31) * W4_KC);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int fi = 0; fi < 9; ++fi) {
            if constexpr (MM) {
                if (fi + 1 < 9) fa[(fi + 1) & 1] = *reinterpret_cast<const float4*>(Vb + ((fi + 1) * W4_TB + l31) * W4_KC);
                __builtin_amdgcn_sched_barrier(0);                     // the next group's V fragment is requested BEFORE this group's MFMAs go out
                const float4 va = fa[fi & 1], fb = bq[fi];
                acc[fi] = __builtin_amdgcn_mfma_f32_32x32x2f32(va.x, fb.x, acc[fi], 0, 0, 0);
                acc[fi] = __builtin_amdgcn_mfma_f32_32x32x2f32(va.y, fb.y, acc[fi], 0, 0, 0);
                acc[fi] = __builtin_amdgcn_mfma_f32_32x32x2f32(va.z, fb.z, acc[fi], 0, 0, 0);
                acc[fi] = __builtin_amdgcn_mfma_f32_32x32x2f32(va.w, fb.w, acc[fi], 0, 0, 0);
                load_u(cn, fi);                                        // next chunk's block into the registers just consumed
            }
            if constexpr (LDR) {
                if (fi == 0) w4_bt_inplace(pv);
                if (fi == 1) {
                    float* d = Ss + st * W4_SSTAGE + s_wr;
#pragma unroll
                    for (int j = 0; j < 6; ++j) *reinterpret_cast<float4*>(d + j * 24) = pv[j];
                }
                if (fi == 2) {
                    const float* r = Ss + (st ^ 1) * W4_SSTAGE + s_rd;
#pragma unroll
                    for (int i = 0; i < 6; ++i) pv[i] = *reinterpret_cast<const float4*>(r + i * 4);
                }
                if (fi == 3) w4_bt_inplace(pv);
                if (fi == 4) {
                    float* d = Vs + (st ^ 1) * W4_STAGE + v_wr;
#pragma unroll
                    for (int i = 0; i < 6; ++i) *reinterpret_cast<float4*>(d + i * 6 * W4_PLANE) = pv[i];
                }
                if (fi == 5) stage_G(min(cc + 3, NC - 1));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();
    }
    };
    const bool mm = __builtin_amdgcn_readfirstlane((int)nb_ok) != 0 && !(W4_DBG(a) & 2);
    if (ldr) { if (mm) main_loop(std::true_type{}, std::true_type{}); else main_loop(std::true_type{}, std::false_type{}); }
    else     { if (mm) main_loop(std::false_type{}, std::true_type{}); else main_loop(std::false_type{}, std::false_type{}); }

    // ---- epilogue: two halves of the 32 tiles in turn through the (now free) LDS, both channel blocks and all eight waves at once ----
    // Accumulator rows 0..15 are registers e = 0..7 of every f32x16, rows 16..31 registers 8..15: half h is dead in the register file once written, so
    // the output pass of half 0 runs beside 72 live accumulator registers only (the whole-tile variant spilled, the block-by-block one idled four waves).
    W4_TS(a, 3);
#ifdef FD_W4_TIMING
    ++w4_segs;
#endif
    int sk_next = 0;
    if (W4_DBG(a) & 4) { W4_TS(a, 4); return; }
    if constexpr (SK) {
        // the next claim goes out now and is read behind the epilogue
        if (tid == 0) sk_next = __hip_atomic_fetch_add(sk_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // piece 0 of a cut item: wait for its parts (all claimed before this piece was; the flags are lowered again here: zero after the launch)
        if (nparts && tid == 0) {
#ifdef FD_W4_TIMING
            const long long tw = wall_clock64();
#endif
            for (int p = 0; p < nparts; ++p) {
                int* f = a.flags + xcd + 8 * (slot + p);
                while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(8);
                __hip_atomic_store(f, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#ifdef FD_W4_TIMING
            if (a.ts) a.ts[(size_t)blockIdx.x * 8 + 5] = wall_clock64() - tw;
#endif
        }
    }
    float* Ms = reinterpret_cast<float*>(smem);                          // [2 ch][36 f][16 tiles][32 cout]
    // output role: thread = (channel block, row pair of the 4 x 4 outputs, tile of the half, cout quad)
    int te = tid;
    asm volatile("" : "+v"(te));                                         // (opaque: nothing of the output role is computed, and kept in registers, ahead of the main loop)
    const int ech = te >> 8, eh = (te >> 7) & 1, et = (te >> 3) & 15, eq = te & 7;
    const int nn = n0 + ech * 32 + eq * 4;
    const float4 sc = (a.scale && nn < a.Cout) ? *reinterpret_cast<const float4*>(a.scale + nn) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 sf = (a.shift && nn < a.Cout) ? *reinterpret_cast<const float4*>(a.shift + nn) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (h) __syncthreads();                                          // (the output pass of half 0 has read its planes)
#pragma unroll
        for (int fi = 0; fi < 9; ++fi) {
            float* d = Ms + ((ch * 36 + 9 * g + fi) * 16) * 32 + l31;
#pragma unroll
            for (int e = 0; e < 8; ++e) d[((e & 3) + 8 * (e >> 2) + 4 * lh) * 32] = acc[fi][8 * h + e];
        }
        __syncthreads();
        const Tile4 ep = wino4_decode(a, tile0 + 16 * h + et);
        const int eH = a.H[ep.s], eW = a.W[ep.s];
        const float eprm = a.seg_param[ep.s];
        if (ep.ok && nn < a.Cout && !(W4_DBG(a) & 16)) {       // (timing builds, 16: dump + barriers only)
            // Y = A^T M A, A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]: per frequency row i the column pass r_i[y], then
            // Y[x][y] += A^T[x][i] * r_i[y] for this thread's two rows x = 2 eh, 2 eh + 1 (eh is wave-uniform; the zero entries of A^T cost nothing)
            float4 Y[2][4];
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y) Y[x][y] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                float4 m[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) m[j] = *reinterpret_cast<const float4*>(Ms + ((ech * 36 + 6 * i + j) * 16 + et) * 32 + eq * 4);
                float4 r[4];
#define W4_AT1(c)                                                                     \
    {                                                                                 \
        const float p12 = m[1].c + m[2].c, m12 = m[1].c - m[2].c;                      \
        const float p34 = m[3].c + m[4].c, m34 = m[3].c - m[4].c;                      \
        r[0].c = m[0].c + p12 + p34;                                                  \
        r[1].c = m12 + 2.f * m34;                                                     \
        r[2].c = p12 + 4.f * p34;                                                     \
        r[3].c = m12 + 8.f * m34 + m[5].c;                                            \
    }
                W4_AT1(x) W4_AT1(y) W4_AT1(z) W4_AT1(w)
#undef W4_AT1
                constexpr float AT[4][6] = {{1.f, 1.f, 1.f, 1.f, 1.f, 0.f}, {0.f, 1.f, -1.f, 2.f, -2.f, 0.f}, {0.f, 1.f, 1.f, 4.f, 4.f, 0.f}, {0.f, 1.f, -1.f, 8.f, -8.f, 1.f}};
                const float c0 = eh ? AT[2][i] : AT[0][i], c1 = eh ? AT[3][i] : AT[1][i];
                if (AT[0][i] != 0.f || AT[2][i] != 0.f) {
#pragma unroll
                    for (int y = 0; y < 4; ++y) {
                        Y[0][y].x += c0 * r[y].x; Y[0][y].y += c0 * r[y].y; Y[0][y].z += c0 * r[y].z; Y[0][y].w += c0 * r[y].w;
                    }
                }
                if (AT[1][i] != 0.f || AT[3][i] != 0.f) {
#pragma unroll
                    for (int y = 0; y < 4; ++y) {
                        Y[1][y].x += c1 * r[y].x; Y[1][y].y += c1 * r[y].y; Y[1][y].z += c1 * r[y].z; Y[1][y].w += c1 * r[y].w;
                    }
                }
            }
            if constexpr (SK) {
                // slot layout [half][row x][column y][512 threads] float4: the writer and the finisher are the same thread of their workgroups
                if (part) {
                    const int so = ((xcd + 8 * slot) * 16 + h * 8) * (512 * 16);
#pragma unroll
                    for (int x = 0; x < 2; ++x)
#pragma unroll
                        for (int y = 0; y < 4; ++y)
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, Y[x][y]), wrsrc, te * 16,
                                                                   so + (x * 4 + y) * (512 * 16), SC1);
                    continue;
                }
                for (int p = 0; p < nparts; ++p) {
                    const int so = ((xcd + 8 * (slot + p)) * 16 + h * 8) * (512 * 16);
#pragma unroll
                    for (int x = 0; x < 2; ++x)
#pragma unroll
                        for (int y = 0; y < 4; ++y) {
                            const float4 o = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, te * 16, so + (x * 4 + y) * (512 * 16), SC1));
                            Y[x][y].x += o.x; Y[x][y].y += o.y; Y[x][y].z += o.z; Y[x][y].w += o.w;
                        }
                }
            }
            // The eight stores of this thread as TWO copies of the loop: FAST = no activation or ReLU / SiLU on every channel of the tile -- a dozen instructions
            // per store, contiguous; the general copy carries the residual fetch and fd_act's whole switch (ScaleExp, SiLU, sigmoid) per channel.  As ONE loop the
            // ReLU layers hopped through ~8 000 instructions of mostly skipped code per output pass (the finding of DESIGN 4.3c on the AMP kernel: 40 % there).
            auto store_sites = [&](auto fast_c) {
                constexpr bool FAST = decltype(fast_c)::value;
#pragma unroll
                for (int x = 0; x < 2; ++x) {
                    const int hh = (ep.h0 + 2 * eh + x) * a.dil + ep.ph;
                    if (hh >= eH) continue;
#pragma unroll
                    for (int y = 0; y < 4; ++y) {
                        const int w = (ep.w0 + y) * a.dil + ep.pw;
                        if (w >= eW) continue;
                        const size_t m_ = (size_t)(a.m0[ep.s] + (ep.n * eH + hh) * eW + w);
                        float4 v = make_float4(Y[x][y].x * sc.x + sf.x, Y[x][y].y * sc.y + sf.y, Y[x][y].z * sc.z + sf.z, Y[x][y].w * sc.w + sf.w);
                        if constexpr (FAST) {
                            if (a.res) {                       // (uniform) the training step's data gradients: ReLU mask of the layer below, or an addend
                                const float4 rr = *reinterpret_cast<const float4*>(a.res + m_ * a.res_cs + a.res_co + nn);
                                if (a.res_mask) {
                                    v.x = rr.x > 0.f ? v.x : 0.f; v.y = rr.y > 0.f ? v.y : 0.f; v.z = rr.z > 0.f ? v.z : 0.f; v.w = rr.w > 0.f ? v.w : 0.f;
                                } else {
                                    v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
                                }
                            }
                            if (a.act == FD_ACT_RELU) {        // (uniform)
                                v.x = fd_act(v.x, FD_ACT_RELU, 0.f); v.y = fd_act(v.y, FD_ACT_RELU, 0.f); v.z = fd_act(v.z, FD_ACT_RELU, 0.f); v.w = fd_act(v.w, FD_ACT_RELU, 0.f);
                            } else if (a.act == FD_ACT_SILU) {
                                v.x = fd_act(v.x, FD_ACT_SILU, 0.f); v.y = fd_act(v.y, FD_ACT_SILU, 0.f); v.z = fd_act(v.z, FD_ACT_SILU, 0.f); v.w = fd_act(v.w, FD_ACT_SILU, 0.f);
                            }
                        } else {
                            if (a.res) {
                                const float4 rr = *reinterpret_cast<const float4*>(a.res + m_ * a.res_cs + a.res_co + nn);
                                if (a.res_mask) {
                                    v.x = rr.x > 0.f ? v.x : 0.f; v.y = rr.y > 0.f ? v.y : 0.f; v.z = rr.z > 0.f ? v.z : 0.f; v.w = rr.w > 0.f ? v.w : 0.f;
                                } else {
                                    v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
                                }
                            }
                            if (a.act != FD_ACT_NONE) {
                                if (nn + 0 >= a.act_c0) v.x = fd_act(v.x, a.act, eprm);
                                if (nn + 1 >= a.act_c0) v.y = fd_act(v.y, a.act, eprm);
                                if (nn + 2 >= a.act_c0) v.z = fd_act(v.z, a.act, eprm);
                                if (nn + 3 >= a.act_c0) v.w = fd_act(v.w, a.act, eprm);
                            }
                        }
                        if ((W4_DBG(a) & 8) && v.x != 12345.678f) continue;          // (timing builds: the epilogue without its global stores)
                        *reinterpret_cast<float4*>(a.y + (size_t)blockIdx.y * a.slice_stride + m_ * a.y_cs + a.y_co + nn) = v;
                    }
                }
            };
            if (a.act == FD_ACT_NONE || ((a.act == FD_ACT_RELU || a.act == FD_ACT_SILU) && a.act_c0 <= n0)) store_sites(std::integral_constant<bool, true>{});       // (uniform)
            else store_sites(std::integral_constant<bool, false>{});
        }
    }
    if constexpr (!SK) break;
    if (part) __builtin_amdgcn_s_waitcnt(0x0f70);                        // vmcnt(0): this wave's device-scope stores to the slot have been acknowledged
    if (tid == 0) {
        if (sk_next == sk_units - 1) __hip_atomic_store(sk_ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // the queue's last claim: ready for the next launch
        *claim_lds = sk_next + a.wpx;
    }
    __syncthreads();                                                     // (also: the output pass has read its planes -- the next segment's prologue reuses the LDS)
    if (part && tid == 0) __hip_atomic_store(a.flags + xcd + 8 * slot, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // the slot is complete: raise its flag
    u = __builtin_amdgcn_readfirstlane(*claim_lds);
#ifdef FD_W4_TIMING
    w4_last = wall_clock64();                                            // (the epilogue is not part of the next segment's set-up time)
#endif
  }
#ifdef FD_W4_TIMING
    __syncthreads();
    W4_TS(a, 4);
#endif
}

// U = G g G^T per filter (fd_wino4_pack_one, fd_conv_common.h): one (n, k) filter per thread.
__global__ __launch_bounds__(256) void wino4_pack_kernel(const float* __restrict__ w, const float* __restrict__ scale, float* __restrict__ out,
                                                         int N, int K, int mode) {
    const long total = (long)((N + 31) & ~31) * K;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int n = (int)(i / K), k = (int)(i - (long)n * K);
        fd_wino4_pack_one(w, scale, out, N, K, mode, n, k);
    }
}

extern "C" int64_t fd_wino4_weight_bytes(int32_t N, int32_t K) {
    if (N < 1 || K < 8 || K % 8) return -1;
    return (int64_t)((N + 31) & ~31) * K * 36 * 4;
}

extern "C" int32_t fd_wino4_pack_weights_f32(const float* w, const float* scale, float* out, int32_t Cout, int32_t Cin, int32_t mode, fd_stream_t stream) {
    FD_REQUIRE(w && out && Cout >= 1 && Cin >= 1 && (mode == 0 || mode == 1), FD_E_INVAL, "fd_wino4_pack_weights: bad arguments");
    const int N = mode == 0 ? Cout : Cin, K = mode == 0 ? Cin : Cout;
    FD_REQUIRE(K % 8 == 0, FD_E_UNSUPPORTED, "fd_wino4_pack_weights: the reduction width (%d) must be a multiple of 8", K);
    const long total = (long)((N + 31) & ~31) * K;
    long gsz = (total + 255) / 256;
    if (gsz > 8192) gsz = 8192;
    hipLaunchKernelGGL(wino4_pack_kernel, dim3((unsigned)gsz), dim3(256), 0, (hipStream_t)stream, w, scale, out, N, K, mode);
    FD_CHECK_LAUNCH("fd_wino4_pack_weights_f32");
    return FD_OK;
}

// workspace of the persistent form: [8 KB header: 1024 slot flags, then the 8 XCDs' queue heads (ints)][slots: 16 x 512 float4 = 128 KB per workgroup]
#define W4_SK_FLAG_BYTES(wgs) 8192L
static int64_t fd_wino4_sk_workspace_bytes_impl(int wgs) { return W4_SK_FLAG_BYTES(wgs) + (int64_t)wgs * 16 * 512 * 16; }
extern "C" int64_t fd_conv_sk_workspace_bytes(int32_t sk_wgs) {
    if (sk_wgs < 8 || sk_wgs % 8 || sk_wgs > 1024) return -1;
    return fd_wino4_sk_workspace_bytes_impl(sk_wgs);
}

// workgroups of the layer's F(4x4) launch (grid.x; with split-K: per slice): what fd_conv_params.wg_first / wg_count index
int fd_wino4_workgroups(const fd_conv_params* p) {
    long t = 0;
    for (int s = 0; s < p->in.nseg; ++s)
        t += (long)p->in.batch * p->dil * p->dil * (((p->in.H[s] + p->dil - 1) / p->dil + 3) / 4) * (((p->in.W[s] + p->dil - 1) / p->dil + 3) / 4);
    const long mtiles = (t + W4_TB - 1) / W4_TB;
    return (int)(8 * ((mtiles + 7) / 8) * ((p->Cout + 63) / 64));
}

// ... and how many of the workgroups [first, first + count) of that grid are non-empty (the grid is padded to 8 XCDs x mt_per M tiles: the last XCD's share ends early)
int fd_wino4_workgroups_live(const fd_conv_params* p, int first, int count) {
    long t = 0;
    for (int s = 0; s < p->in.nseg; ++s)
        t += (long)p->in.batch * p->dil * p->dil * (((p->in.H[s] + p->dil - 1) / p->dil + 3) / 4) * (((p->in.W[s] + p->dil - 1) / p->dil + 3) / 4);
    const int mtiles = (int)((t + W4_TB - 1) / W4_TB), mt_per = (mtiles + 7) / 8, ntiles = (p->Cout + 63) / 64;
    int live = 0;
    for (int b = first; b < first + count; ++b) {
        const int xcd = b & 7, idx = b >> 3;
        const int cnt = min(mtiles - xcd * mt_per, mt_per);
        if (cnt > 0 && idx < cnt * ntiles) ++live;
    }
    return live;
}

int fd_launch_conv_wino4(const fd_conv_params* p, hipStream_t stream) {
    FD_REQUIRE(p->KH == 3 && p->KW == 3 && p->stride == 1 && (p->dil == 1 || p->dil == 2) && p->pad == p->dil && p->Cin % 8 == 0 && p->Cout % 4 == 0 &&
                   p->precision == FD_PREC_F32 && p->out_H <= 0 && p->sc_H <= 0 && !p->gate && !p->gn_stats,
               FD_E_UNSUPPORTED, "fd_conv2d: FD_TILE_WINOGRAD4 needs an fp32 3x3 stride-1 'same' conv of dilation 1 or 2 with Cin %% 8 == 0, Cout %% 4 == 0, no scatter / gate / gn_stats");
    FD_REQUIRE(p->x_cs % 4 == 0 && p->x_co % 4 == 0 && p->y_cs % 4 == 0 && p->y_co % 4 == 0 && ((uintptr_t)p->y & 15) == 0 &&
                   (!p->res || (p->res_cs % 4 == 0 && p->res_co % 4 == 0 && ((uintptr_t)p->res & 15) == 0)) &&
                   (!p->scale || ((uintptr_t)p->scale & 15) == 0) && (!p->shift || ((uintptr_t)p->shift & 15) == 0),
               FD_E_UNSUPPORTED, "fd_conv2d: FD_TILE_WINOGRAD4 needs 16-byte addressable input / output / residual / scale / shift views");
    FD_REQUIRE(p->sk_wgs <= 0 || (p->ksplit <= 1 && p->wg_count <= 0 && p->sk_wgs % 8 == 0 && p->sk_wgs <= 1024), FD_E_INVAL,
               "fd_conv2d: sk_wgs = %d must be a multiple of 8, at most 1024, without split-K / wg_count", p->sk_wgs);
    Wino4Args a;
    a.x = p->x; a.u = p->w; a.scale = p->scale; a.shift = p->shift; a.res = p->res; a.y = p->y;
    a.x_cs = p->x_cs; a.x_co = p->x_co; a.res_cs = p->res_cs; a.res_co = p->res_co; a.y_cs = p->y_cs; a.y_co = p->y_co;
    a.Cin = p->Cin; a.Cout = p->Cout; a.act = p->act; a.act_c0 = p->act_c0;
    a.res_mask = (p->res && p->res_mode == 1) ? 1 : 0;
    a.NC = p->Cin / 8;
    a.dil = p->dil;
    a.nseg = p->in.nseg;
    long t = 0;
    for (int s = 0; s < FD_MAX_SEG; ++s) {
        a.t0[s] = (int)t;
        if (s < p->in.nseg) {
            a.H[s] = p->in.H[s]; a.W[s] = p->in.W[s];
            a.TH[s] = ((p->in.H[s] + p->dil - 1) / p->dil + 3) / 4; a.TW[s] = ((p->in.W[s] + p->dil - 1) / p->dil + 3) / 4;   // per parity class
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
    const long xb = rows * p->x_cs * 4, ub = (long)((p->Cout + 31) & ~31) * p->Cin * 36 * 4;
    FD_REQUIRE(xb < 0xC0000000L - 65536 && ub < 0xC0000000L, FD_E_UNSUPPORTED, "fd_conv2d: input / weight buffer exceeds 3 GiB");
    a.x_bytes = (unsigned)xb; a.u_bytes = (unsigned)ub;
#ifdef FD_W4_TIMING
    static const int dbg = getenv("FD_W4_DBG") ? atoi(getenv("FD_W4_DBG")) : 0;
    a.dbg = dbg;
#else
    a.dbg = 0;
#endif
    a.ts = nullptr;
    a.mtiles = (a.T + W4_TB - 1) / W4_TB;
    a.ntiles = (p->Cout + 63) / 64;
    a.mt_per = (a.mtiles + 7) / 8;
    constexpr int lds = (2 * W4_STAGE + 2 * W4_SSTAGE) * 4 + 16;   // 146 KB + the SK form's claim word
    a.nc_per = a.NC; a.slice_stride = 0;
    const int ksplit = p->ksplit > 1 ? p->ksplit : 1;
    ConvArgs o = {};
    int ldw = 0; long slab = 0;
    if (ksplit > 1) {
        // split-K (maps with too few tiles to fill 256 CUs, e.g. layer4's 20 x 20): the chunk loop is divided over `ksplit` workgroups per tile, raw
        // partial outputs (the output transform is linear) go to the workspace, the direct kernel's combine launch adds the slabs in slice order and
        // applies the epilogue (deterministic) -- as FD_TILE_WINOGRAD
        FD_REQUIRE(ksplit <= 64 && a.NC >= 2 * ksplit, FD_E_INVAL, "fd_conv2d: ksplit=%d needs 1 < ksplit <= min(64, Cin / 16 = %d)", ksplit, a.NC / 2);
        ldw = (p->Cout + 3) & ~3;
        slab = rows * ldw;
        FD_REQUIRE(p->workspace && ((uintptr_t)p->workspace & 15) == 0 && p->workspace_bytes >= (int64_t)ksplit * slab * 4, FD_E_INVAL,
                   "fd_conv2d: split-K needs a 16-byte aligned workspace of fd_conv_workspace_bytes() bytes");
        a.nc_per = (a.NC + ksplit - 1) / ksplit;
        o.scale = p->scale; o.shift = p->shift; o.res = p->res; o.y = p->y;
        o.res_cs = p->res_cs; o.res_co = p->res_co; o.y_cs = p->y_cs; o.y_co = p->y_co;
        o.Cout = p->Cout; o.act = p->act; o.act_c0 = p->act_c0; o.M = (int)rows; o.nseg = p->in.nseg;
        o.res_mask = (p->res && p->res_mode == 1) ? 1 : 0;
        for (int sg = 0; sg <= FD_MAX_SEG; ++sg) o.m_out[sg] = p->in.m_start[sg < p->in.nseg ? sg : p->in.nseg];
        for (int sg = 0; sg < FD_MAX_SEG; ++sg) { o.seg_param[sg] = p->seg_param[sg]; o.Ho[sg] = o.Wo[sg] = 1; }
        o.sc_on = 0;
        a.y = (float*)p->workspace; a.y_cs = ldw; a.y_co = 0; a.slice_stride = slab;
        a.scale = a.shift = a.res = nullptr; a.act = FD_ACT_NONE; a.res_mask = 0;
    }
    const int nslice = (a.NC + a.nc_per - 1) / a.nc_per;
    dim3 grid((unsigned)(8 * a.mt_per * a.ntiles), (unsigned)nslice);
    a.blk0 = 0;
    a.wpx = 0; a.sk_P = 1; a.ws = nullptr; a.flags = nullptr;
    if (p->sk_wgs > 0) {
        // persistent stream-K form: sk_wgs workgroups (a multiple of 8, one per CU at most) share the layer's (item, chunk) units evenly -- no last round with most of the chip
        // idle (cls_logits: 538 items on 256 CUs ran as 3 rounds for 2.1 rounds of work), no workgroup dispatch / set-up between the items of a CU
        a.wpx = p->sk_wgs / 8;
        FD_REQUIRE((long)a.mt_per * a.ntiles * 16 < (1L << 30), FD_E_UNSUPPORTED, "fd_conv2d: sk_wgs: too many items per XCD");
        {   // pieces per remainder item: the choice that gets the remainder of the fullest XCD through in the least time -- ceil(rem * P / wpx) passes of
            // (NC / P chunks + the fixed cost of a segment), in units of one chunk's time (fixed cost ~ 6.5 chunks: profiles/r05_wino4_fixed_cost.txt); the slots of an XCD
            // (rem * (P - 1)) must fit the workspace's wpx
            static const int force_p = getenv("FD_W4_SK_P") ? atoi(getenv("FD_W4_SK_P")) : 0;
            // (every XCD has its own remainder -- the last one owns fewer M tiles: the slot bound must hold for each, the time is the slowest XCD's)
            int best = 1; double best_t = 1e30; bool any_rem = false;
            for (int P = 1; P <= 4 && P <= a.NC; ++P) {          // (more than 4 pieces: the finisher's serial slot reads cost more than the pieces save -- layer1.conv2 at P = 8: 0.176 against 0.145 ms)
                double t = 0.0; bool ok = true;
                for (int x = 0; x < 8; ++x) {
                    const int cx = (a.mtiles >> 3) + (x < (a.mtiles & 7) ? 1 : 0);
                    if (cx <= 0) continue;
                    const int items = cx * a.ntiles, rem = items % a.wpx;
                    any_rem = any_rem || rem > 0;
                    if (P > 1 && rem * (P - 1) > a.wpx) ok = false;
                    const double tx = (double)(items / a.wpx) * (a.NC + 6.5) + (double)((rem * P + a.wpx - 1) / a.wpx) * ((double)a.NC / P + 6.5 + 1.5 * (P - 1));
                    t = tx > t ? tx : t;
                }
                if (ok && t < best_t - 1e-9) { best_t = t; best = P; }
            }
            a.sk_P = any_rem ? best : 1;
            bool force_ok = force_p > 0 && force_p <= a.NC && force_p <= 16;
            for (int x = 0; x < 8 && force_ok; ++x) {
                const int cx = (a.mtiles >> 3) + (x < (a.mtiles & 7) ? 1 : 0);
                if (cx > 0 && force_p > 1 && (cx * a.ntiles % a.wpx) * (force_p - 1) > a.wpx) force_ok = false;
            }
            if (force_ok) a.sk_P = force_p;
        }
        const int64_t need = fd_wino4_sk_workspace_bytes_impl(p->sk_wgs);
        FD_REQUIRE(p->workspace && ((uintptr_t)p->workspace & 255) == 0 && p->workspace_bytes >= need, FD_E_INVAL,
                   "fd_conv2d: sk_wgs = %d needs a 256-byte aligned workspace of fd_conv_sk_workspace_bytes() = %ld bytes whose first 8192 bytes are zero", p->sk_wgs, (long)need);
        a.flags = (int*)p->workspace;
        a.ws = (float4*)((char*)p->workspace + W4_SK_FLAG_BYTES(p->sk_wgs));
        grid.x = (unsigned)p->sk_wgs;
    }
    if (p->wg_count > 0) {
        // a slice of the layer's grid (fd_conv_workgroups): the head tower's 2 152 workgroups are 8.4 rounds on 256 CUs -- launched as 8 whole rounds + a tail
        // launch, the caller can let other work in beside the tail instead of idling 60 % of the chip for a round (pipeline.TwoLanePipeline)
        FD_REQUIRE(ksplit <= 1 && p->wg_first >= 0 && p->wg_first % 8 == 0 && (long)p->wg_first + p->wg_count <= (long)grid.x, FD_E_INVAL,
                   "fd_conv2d: wg_first = %d (a multiple of 8), wg_count = %d must lie inside the layer's %u workgroups, without split-K", p->wg_first, p->wg_count, grid.x);
        a.blk0 = p->wg_first;
        grid.x = (unsigned)p->wg_count;
    }
    if (ksplit > 1) {
        static std::atomic<unsigned> ms{0};
        fd_set_max_lds_once(ms, reinterpret_cast<const void*>(conv3x3_wino4_kernel<0, false>), lds);
        hipLaunchKernelGGL((conv3x3_wino4_kernel<0, false>), grid, dim3(512), lds, stream, a);
        FD_CHECK_LAUNCH("fd_conv2d_nhwc_f32 (Winograd F(4x4,3x3), split-K)");
        return fd_launch_splitk_reduce(o, (const float*)p->workspace, nslice, ldw, slab, stream);
    }
#ifdef FD_W4_TIMING
    // FD_W4_TS=<file>: every launch is followed by a device synchronisation and one line of per-workgroup phase times (development builds only)
    static const char* ts_path = getenv("FD_W4_TS");
    static long long* ts_dev = nullptr;
    const size_t ts_n = (size_t)grid.x * grid.y * 8;
    if (ts_path) {
        if (!ts_dev) (void)hipMalloc((void**)&ts_dev, (size_t)(1 << 16) * 8 * sizeof(long long));
        if (ts_n <= (size_t)(1 << 16) * 8) { a.ts = ts_dev; (void)hipMemsetAsync(ts_dev, 0, ts_n * sizeof(long long), stream); }
    }
    struct TsDump {
        const char* path; long long* dev; size_t n; hipStream_t st; const Wino4Args& a;
        ~TsDump() {
            if (!path || !a.ts) return;
            (void)hipStreamSynchronize(st);
            std::vector<long long> h(n);
            (void)hipMemcpy(h.data(), dev, n * sizeof(long long), hipMemcpyDeviceToHost);
            long long first = 0, last = 0; double ph[4] = {0, 0, 0, 0}, tot = 0, totmax = 0; size_t live = 0;
            for (size_t b = 0; b < n / 8; ++b) {
                const long long* t = &h[b * 8];
                if (!t[0] || !t[4]) continue;
                if (!live || t[0] < first) first = t[0];
                if (!live || t[4] > last) last = t[4];
                for (int k = 0; k < 4; ++k) ph[k] += (double)(t[k + 1] - t[k]);
                tot += (double)(t[4] - t[0]); if ((double)(t[4] - t[0]) > totmax) totmax = (double)(t[4] - t[0]);
                ++live;
            }
            if (FILE* f = fopen(path, "a")) {
                const double u = 0.01 / (live ? live : 1);     // 100 MHz ticks -> us, mean over the live workgroups
                double wait = 0, waitmax = 0, segs = 0;
                for (size_t b = 0; b < n / 8; ++b) { wait += (double)h[b * 8 + 5]; if ((double)h[b * 8 + 5] > waitmax) waitmax = (double)h[b * 8 + 5]; segs += (double)h[b * 8 + 6]; }
                fprintf(f, "w4ts Cin %d Cout %d T %d wgs %zu live %zu dbg %d sk %d | span_us %.2f | per-wg us (sum over %.2f segments): setup %.2f prologue %.2f loop %.2f epilogue+rest %.2f total %.2f max %.2f | flag wait %.2f max %.2f\n",
                        a.Cin, a.Cout, a.T, n / 8, live, a.dbg, a.wpx * 8, (double)(last - first) * 0.01, segs / (live ? live : 1), ph[0] * u, ph[1] * u, ph[2] * u, ph[3] * u, tot * u, totmax * 0.01, wait * u, waitmax * 0.01);
                fclose(f);
            }
        }
    } ts_dump{ts_path, ts_dev, ts_n, stream, a};
#endif
    if (p->sk_wgs > 0) {
        if (p->tag == 1) {
            static std::atomic<unsigned> k1{0};
            fd_set_max_lds_once(k1, reinterpret_cast<const void*>(conv3x3_wino4_kernel<1, true>), lds);
            hipLaunchKernelGGL((conv3x3_wino4_kernel<1, true>), grid, dim3(512), lds, stream, a);
        } else {
            static std::atomic<unsigned> k0{0};
            fd_set_max_lds_once(k0, reinterpret_cast<const void*>(conv3x3_wino4_kernel<0, true>), lds);
            hipLaunchKernelGGL((conv3x3_wino4_kernel<0, true>), grid, dim3(512), lds, stream, a);
        }
    } else if (p->tag == 1) {
        static std::atomic<unsigned> m1{0};
        fd_set_max_lds_once(m1, reinterpret_cast<const void*>(conv3x3_wino4_kernel<1, false>), lds);
        hipLaunchKernelGGL((conv3x3_wino4_kernel<1, false>), grid, dim3(512), lds, stream, a);
    } else {
        static std::atomic<unsigned> m0{0};
        fd_set_max_lds_once(m0, reinterpret_cast<const void*>(conv3x3_wino4_kernel<0, false>), lds);
        hipLaunchKernelGGL((conv3x3_wino4_kernel<0, false>), grid, dim3(512), lds, stream, a);
    }
    FD_CHECK_LAUNCH("fd_conv2d_nhwc_f32 (Winograd F(4x4,3x3))");
    return FD_OK;
}
