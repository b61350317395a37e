// Torch-free host of the C-ABI: plain HIP runtime + include/fcosdet.h, the way a non-Python caller (the cgo / JNI /
// ctypes stub of INTEGRATION.md) would drive it.  Test infrastructure: it links the oracle's C restatement
// (oracle/postproc_ref.c) as the checker.  Exit code 0 = every check passed.
//
//   hipcc -O2 --offload-arch=gfx950 -I include tests/abi/abi_host_check.cpp oracle/postproc_ref.c \
//         -L pytorch_object_detection_amd/csrc -lfcosdet_hip -Wl,-rpath,... -o oracle/_build/abi_host_check
//
// Checks: fd_version, error path (null pointer -> negative code + fd_last_error text), fd_batched_nms kept indices
// bit-identical to ref_post_process on 4 images x 1000 candidates (crowded boxes, 80 classes), fd_clip_boxes,
// fd_pairwise_iou vs ref_pairwise_iou bit for bit, one fused fd_conv2d_nhwc_f32 launch vs a naive fp64 loop.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "fcosdet.h"

extern "C" {
int ref_post_process(const float* scores, const int64_t* classes, const float* boxes, int K, float score_thr, double iou_thr,
                     int32_t* keep);
void ref_pairwise_iou(const float* a, const float* b, int Na, int Nb, int plus_one, float* out);
void ref_clip_boxes(float* boxes, int n, int img_h, int img_w);
}

#define HIPOK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define CHECK(c, ...) do { if (!(c)) { std::printf("FAILED: " __VA_ARGS__); std::printf("\n"); return 1; } } while (0)

static uint32_t lcg_state = 12345u;
static float urand() { lcg_state = lcg_state * 1664525u + 1013904223u; return (float)(lcg_state >> 8) / 16777216.0f; }

template <class T> static T* dev_copy(const std::vector<T>& h) {
    T* d = nullptr;
    if (hipMalloc((void**)&d, h.size() * sizeof(T)) != hipSuccess) return nullptr;
    if (hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
}

int main() {
    CHECK(fd_version() >= 100, "fd_version() = %d", fd_version());
    CHECK(fd_clip_boxes(nullptr, 4, 10, 10, nullptr) < 0 && std::strlen(fd_last_error()) > 0, "null pointer not rejected");

    const int N = 4, K = 1000;
    std::vector<float> scores((size_t)N * K), boxes((size_t)N * K * 4);
    std::vector<int64_t> classes((size_t)N * K);
    for (int n = 0; n < N; ++n) {
        // score-descending rows (what fd_fcos_topk hands over), clustered boxes so that suppression really happens
        float s = 0.999f;
        for (int i = 0; i < K; ++i) {
            s *= 0.9965f;                                   // ~0.03 at the end: the 0.05 threshold cuts the tail
            scores[(size_t)n * K + i] = s;
            classes[(size_t)n * K + i] = 1 + (int64_t)(urand() * 80.0f) % 80;
            const float cx = 40.f + 560.f * (float)((i * 7 + n) % 97) / 97.f + 6.f * urand();
            const float cy = 40.f + 560.f * (float)((i * 13 + n) % 89) / 89.f + 6.f * urand();
            const float w = 16.f + 200.f * urand() * urand(), h = 16.f + 200.f * urand() * urand();
            float* b = &boxes[((size_t)n * K + i) * 4];
            b[0] = cx - w / 2; b[1] = cy - h / 2; b[2] = cx + w / 2; b[3] = cy + h / 2;
        }
    }
    float *d_scores = dev_copy(scores), *d_boxes = dev_copy(boxes);
    int64_t* d_classes = dev_copy(classes);
    CHECK(d_scores && d_boxes && d_classes, "hipMalloc");
    float *o_scores, *o_boxes; int64_t* o_classes; int32_t *keep, *counts; void* ws;
    HIPOK(hipMalloc((void**)&o_scores, (size_t)N * K * 4)); HIPOK(hipMalloc((void**)&o_boxes, (size_t)N * K * 16));
    HIPOK(hipMalloc((void**)&o_classes, (size_t)N * K * 8)); HIPOK(hipMalloc((void**)&keep, (size_t)N * K * 4));
    HIPOK(hipMalloc((void**)&counts, N * 4));
    const int64_t wsb = fd_nms_workspace_bytes(N, K);
    CHECK(wsb > 0, "fd_nms_workspace_bytes = %lld", (long long)wsb);
    HIPOK(hipMalloc(&ws, (size_t)wsb));
    hipStream_t st;
    HIPOK(hipStreamCreate(&st));
    const int rc = fd_batched_nms(d_scores, d_classes, d_boxes, N, K, 0.05f, 0.6, o_scores, o_classes, o_boxes, keep, counts, ws, st);
    CHECK(rc == FD_OK, "fd_batched_nms rc=%d: %s", rc, fd_last_error());
    HIPOK(hipStreamSynchronize(st));
    std::vector<int32_t> h_keep((size_t)N * K), h_counts(N), r_keep(K);
    HIPOK(hipMemcpy(h_keep.data(), keep, h_keep.size() * 4, hipMemcpyDeviceToHost));
    HIPOK(hipMemcpy(h_counts.data(), counts, N * 4, hipMemcpyDeviceToHost));
    long kept_total = 0;
    for (int n = 0; n < N; ++n) {
        const int nk = ref_post_process(&scores[(size_t)n * K], &classes[(size_t)n * K], &boxes[(size_t)n * K * 4], K, 0.05f, 0.6, r_keep.data());
        CHECK(nk == h_counts[n], "image %d: kept %d, oracle %d", n, h_counts[n], nk);
        CHECK(nk > 50 && nk < K, "image %d: degenerate test data (kept %d)", n, nk);
        for (int i = 0; i < nk; ++i)
            CHECK(h_keep[(size_t)n * K + i] == r_keep[i], "image %d: kept[%d] = %d, oracle %d", n, i, h_keep[(size_t)n * K + i], r_keep[i]);
        CHECK(nk == K || h_keep[(size_t)n * K + nk] == -1, "image %d: padding is not -1", n);
        kept_total += nk;
    }

    // clip the kept boxes to a 512 x 640 image, in place, and compare with the oracle on the host copy
    CHECK(fd_clip_boxes(o_boxes, N * K, 512, 640, st) == FD_OK, "fd_clip_boxes: %s", fd_last_error());
    std::vector<float> h_ob((size_t)N * K * 4), r_ob;
    HIPOK(hipStreamSynchronize(st));
    HIPOK(hipMemcpy(h_ob.data(), o_boxes, h_ob.size() * 4, hipMemcpyDeviceToHost));
    for (int n = 0; n < N; ++n) {
        r_ob.assign((size_t)h_counts[n] * 4, 0.f);
        for (int i = 0; i < h_counts[n]; ++i) std::memcpy(&r_ob[(size_t)i * 4], &boxes[((size_t)n * K + h_keep[(size_t)n * K + i]) * 4], 16);
        ref_clip_boxes(r_ob.data(), h_counts[n], 512, 640);
        CHECK(std::memcmp(r_ob.data(), &h_ob[(size_t)n * K * 4], r_ob.size() * 4) == 0, "image %d: clipped boxes differ", n);
    }

    // pairwise IoU (no +1), first 200 x 300 boxes of image 0 vs image 1
    const int Na = 200, Nb = 300;
    float* d_iou;
    HIPOK(hipMalloc((void**)&d_iou, (size_t)Na * Nb * 4));
    CHECK(fd_pairwise_iou(d_boxes, d_boxes + (size_t)K * 4, Na, Nb, 0, d_iou, st) == FD_OK, "fd_pairwise_iou: %s", fd_last_error());
    std::vector<float> h_iou((size_t)Na * Nb), r_iou((size_t)Na * Nb);
    HIPOK(hipStreamSynchronize(st));
    HIPOK(hipMemcpy(h_iou.data(), d_iou, h_iou.size() * 4, hipMemcpyDeviceToHost));
    ref_pairwise_iou(boxes.data(), &boxes[(size_t)K * 4], Na, Nb, 0, r_iou.data());
    CHECK(std::memcmp(h_iou.data(), r_iou.data(), h_iou.size() * 4) == 0, "pairwise IoU differs from the oracle");

    // one fused conv launch: 3x3, 64 -> 96 channels, folded-BN scale / shift, residual, ReLU on a 2 x 9 x 7 map, against a
    // naive double-precision loop (tolerance 1e-4: north_star's conv bar); weights packed as the header documents
    {
        const int B = 2, H = 9, W = 7, Cin = 64, Cout = 96, KH = 3;
        const int M = B * H * W;
        std::vector<float> x((size_t)M * Cin), wt((size_t)Cout * Cin * 9), wp(wt.size()), sc(Cout), sf(Cout), res((size_t)M * Cout);
        for (auto& v : x) v = urand() * 2.f - 1.f;
        for (auto& v : wt) v = (urand() * 2.f - 1.f) / 24.f;          // [co][ci][r][q]
        for (auto& v : sc) v = 0.5f + urand();
        for (auto& v : sf) v = urand() - 0.5f;
        for (auto& v : res) v = urand() - 0.5f;
        for (int co = 0; co < Cout; ++co)                              // -> [Cout][Cin/32][KH][KW][32]
            for (int ci = 0; ci < Cin; ++ci)
                for (int t = 0; t < 9; ++t)
                    wp[(((size_t)co * (Cin / 32) + ci / 32) * 9 + t) * 32 + ci % 32] = wt[((size_t)co * Cin + ci) * 9 + t];
        float *dx = dev_copy(x), *dw = dev_copy(wp), *dsc = dev_copy(sc), *dsf = dev_copy(sf), *dres = dev_copy(res), *dy;
        HIPOK(hipMalloc((void**)&dy, (size_t)M * Cout * 4));
        fd_conv_params p;
        std::memset(&p, 0, sizeof p);
        p.x = dx; p.w = dw; p.scale = dsc; p.shift = dsf; p.res = dres; p.y = dy;
        p.x_cs = Cin; p.res_cs = Cout; p.y_cs = Cout;
        p.Cin = Cin; p.Cout = Cout; p.KH = p.KW = KH; p.stride = 1; p.pad = 1; p.dil = 1;
        p.act = FD_ACT_RELU; p.mode = FD_CONV_GENERIC; p.precision = FD_PREC_F32;
        p.in.nseg = 1; p.in.batch = B; p.in.H[0] = H; p.in.W[0] = W; p.in.m_start[0] = 0;
        for (int s = 1; s <= FD_MAX_SEG; ++s) p.in.m_start[s] = M;
        CHECK(fd_conv2d_nhwc_f32(&p, st) == FD_OK, "fd_conv2d_nhwc_f32: %s", fd_last_error());
        std::vector<float> y((size_t)M * Cout);
        HIPOK(hipStreamSynchronize(st));
        HIPOK(hipMemcpy(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int n = 0; n < B; ++n)
            for (int h = 0; h < H; ++h)
                for (int w = 0; w < W; ++w)
                    for (int co = 0; co < Cout; ++co) {
                        double a = 0;
                        for (int r = 0; r < 3; ++r)
                            for (int q = 0; q < 3; ++q) {
                                const int hi = h + r - 1, wi = w + q - 1;
                                if (hi < 0 || hi >= H || wi < 0 || wi >= W) continue;
                                const float* px = &x[((size_t)(n * H + hi) * W + wi) * Cin];
                                for (int ci = 0; ci < Cin; ++ci) a += (double)px[ci] * wt[((size_t)co * Cin + ci) * 9 + r * 3 + q];
                            }
                        const size_t m = (size_t)(n * H + h) * W + w;
                        double v = a * sc[co] + sf[co] + res[m * Cout + co];
                        if (v < 0) v = 0;
                        const double d = v - y[m * Cout + co];
                        if (d > worst) worst = d;
                        if (-d > worst) worst = -d;
                    }
        CHECK(worst < 1e-4, "conv differs from the naive loop by %g", worst);
    }

    std::printf("abi_host_check ok: fused conv within 1e-4 of a naive fp64 loop; %d images x %d candidates, %ld boxes kept, indices / clipped boxes / IoU identical to the oracle\n", N, K, kept_total);
    return 0;
}
