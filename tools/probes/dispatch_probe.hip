// Workgroup dispatch-rate probe (diagnostic): empty workgroups of 256 threads with the conv kernel's footprint (24 KB LDS, ~128 VGPRs).
// hipcc --offload-arch=gfx950 -O3 dispatch_probe.hip -o /tmp/dispatch_probe && /tmp/dispatch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int SLEEP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(128))) void empty_kernel(float* out) {
    extern __shared__ float lds[];
    if (SLEEP) for (int i = 0; i < SLEEP; ++i) __builtin_amdgcn_s_sleep(100);      // ~2.9 us each
    if (out && threadIdx.x == 1024) out[blockIdx.x] = lds[threadIdx.x];            // (never true: keeps the LDS allocation alive)
}

int main() {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float* out; CK(hipMalloc(&out, 1 << 20));
    const int lds_sizes[] = {0, 24576, 65536};
    for (int li = 0; li < 3; ++li)
        for (int sl = 0; sl < 2; ++sl)
            for (int n : {1024, 6400, 12800, 25600}) {
                float best = 1e9f;
                for (int rep = 0; rep < 5; ++rep) {
                    CK(hipEventRecord(e0));
                    for (int k = 0; k < 10; ++k) {
                        if (sl == 0) hipLaunchKernelGGL(empty_kernel<0>, dim3(n), dim3(256), lds_sizes[li], 0, out);
                        else hipLaunchKernelGGL(empty_kernel<2>, dim3(n), dim3(256), lds_sizes[li], 0, out);
                    }
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
                    if (ms < best) best = ms;
                }
                printf("LDS %6d B, body %s, %6d workgroups: %7.1f us  -> %6.1f workgroups/us\n", lds_sizes[li], sl ? "5.8 us sleep" : "empty      ", n, best * 1e3, n / (best * 1e3));
            }
    return 0;
}
