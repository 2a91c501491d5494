// Host cost of a kernel launch as a function of the kernel-argument size (build: hipcc --offload-arch=gfx950).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Small { int a[4]; };
struct Big { int a[600]; };   // 2.4 KB, like the pyramid kernel's image list + frame pointers
__global__ void k_small(Small s, int* out) { if (s.a[0] == 12345) out[0] = 1; }
__global__ void k_big(Big b, int* out) { if (b.a[0] == 12345) out[0] = 1; }
int main() {
    int* d; hipMalloc(&d, 4);
    hipStream_t q; hipStreamCreateWithFlags(&q, hipStreamNonBlocking);
    Small s{}; Big b{};
    for (int rep = 0; rep < 3; rep++) {
        for (int which = 0; which < 2; which++) {
            hipStreamSynchronize(q);
            auto t0 = std::chrono::steady_clock::now();
            const int N = 2000;
            for (int i = 0; i < N; i++) {
                if (which == 0) hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, q, s, d);
                else hipLaunchKernelGGL(k_big, dim3(64), dim3(256), 0, q, b, d);
                if ((i & 31) == 31) hipStreamSynchronize(q);
            }
            hipStreamSynchronize(q);
            auto t1 = std::chrono::steady_clock::now();
            std::printf("%s args: %.2f us per launch (incl. a sync every 32)\n", which ? "2.4 KB" : "16 B", std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
        }
    }
    return 0;
}
