// Write-bandwidth probe (diagnostic): the conv epilogue's store pattern against contiguous ones on a [M][N] fp32 map.
// hipcc --offload-arch=gfx950 -O3 store_probe.hip -o /tmp/store_probe && /tmp/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// A: the igemm epilogue: workgroup tile 64 rows x 128 cols, wave = 32 rows x 64 cols as 2 sub-tiles of 32x32; one instruction = 8 rows x 128 B
__global__ __launch_bounds__(256) void pat_epilogue(float* y, int M, int N, int ntn) {
    const int bid = blockIdx.x, mt = bid / ntn, nt = bid - mt * ntn;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, wm = wave >> 1, wn = wave & 1;
    const float4 v = make_float4(1.f, 2.f, 3.f, (float)bid);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int m = mt * 64 + wm * 32 + (lane >> 3) + 8 * p, n = nt * 128 + (wn * 2 + j) * 32 + (lane & 7) * 4;
            if (m < M) *reinterpret_cast<float4*>(y + (size_t)m * N + n) = v;
        }
}
// B: same workgroup tile, but one instruction = 2 rows x 512 B (the tile's whole row width)
__global__ __launch_bounds__(256) void pat_rows512(float* y, int M, int N, int ntn) {
    const int bid = blockIdx.x, mt = bid / ntn, nt = bid - mt * ntn;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float4 v = make_float4(1.f, 2.f, 3.f, (float)bid);
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int m = mt * 64 + wave * 16 + 2 * p + (lane >> 5), n = nt * 128 + (lane & 31) * 4;
        if (m < M) *reinterpret_cast<float4*>(y + (size_t)m * N + n) = v;
    }
}
// C: fully linear: each workgroup writes 32 KB contiguous, one instruction = 1 KB contiguous
__global__ __launch_bounds__(256) void pat_linear(float* y, long total4) {
    const long base = (long)blockIdx.x * 2048 + threadIdx.x;
    const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const long i = base + p * 256;
        if (i < total4) reinterpret_cast<float4*>(y)[i] = v;
    }
}
// D: the epilogue pattern with nontemporal stores
__global__ __launch_bounds__(256) void pat_epilogue_nt(float* y, int M, int N, int ntn) {
    const int bid = blockIdx.x, mt = bid / ntn, nt = bid - mt * ntn;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, wm = wave >> 1, wn = wave & 1;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int m = mt * 64 + wm * 32 + (lane >> 3) + 8 * p, n = nt * 128 + (wn * 2 + j) * 32 + (lane & 7) * 4;
            if (m < M) {
                float* q = y + (size_t)m * N + n;
                __builtin_nontemporal_store(1.f, q); __builtin_nontemporal_store(2.f, q + 1);
                __builtin_nontemporal_store(3.f, q + 2); __builtin_nontemporal_store(4.f, q + 3);
            }
        }
}

// E: the transposed-accumulator epilogue (no LDS stage): one instruction = 32 rows x 32 B
__global__ __launch_bounds__(256) void pat_rows32B(float* y, int M, int N, int ntn) {
    const int bid = blockIdx.x, mt = bid / ntn, nt = bid - mt * ntn;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, wm = wave >> 1, wn = wave & 1;
    const float4 v = make_float4(1.f, 2.f, 3.f, (float)bid);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int m = mt * 64 + wm * 32 + (lane & 31), n = nt * 128 + (wn * 2 + j) * 32 + 8 * g + 4 * (lane >> 5);
            if (m < M) *reinterpret_cast<float4*>(y + (size_t)m * N + n) = v;
        }
}

int main() {
    const int M = 102400, N = 512, ntn = N / 128, nblk = (M / 64) * ntn;
    float* y; CK(hipMalloc(&y, (size_t)M * N * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = (double)M * N * 4;
    for (int which = 0; which < 5; ++which) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(e0));
            for (int k = 0; k < 10; ++k) {
                if (which == 0) hipLaunchKernelGGL(pat_epilogue, dim3(nblk), dim3(256), 0, 0, y, M, N, ntn);
                if (which == 1) hipLaunchKernelGGL(pat_rows512, dim3(nblk), dim3(256), 0, 0, y, M, N, ntn);
                if (which == 2) hipLaunchKernelGGL(pat_linear, dim3(nblk), dim3(256), 0, 0, y, (long)M * N / 4);
                if (which == 4) hipLaunchKernelGGL(pat_rows32B, dim3(nblk), dim3(256), 0, 0, y, M, N, ntn);
                if (which == 3) hipLaunchKernelGGL(pat_epilogue_nt, dim3(nblk), dim3(256), 0, 0, y, M, N, ntn);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
            if (ms < best) best = ms;
        }
        const char* nm[] = {"epilogue pattern (8 rows x 128 B per instruction)", "2 rows x 512 B per instruction", "linear, 1 KB per instruction", "epilogue pattern, nontemporal dword stores", "32 rows x 32 B per instruction (transposed accumulators)"};
        printf("%-52s %7.1f us  %6.2f TB/s\n", nm[which], best * 1e3, bytes / best / 1e9);
    }
    return 0;
}
