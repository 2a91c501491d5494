// Round trip "launch a short kernel, learn on the host that it has finished", three ways:
//   event  : hipEventRecord after the kernel + hipEventSynchronize
//   stream : hipStreamSynchronize
//   flag   : the kernel's last instruction stores a sequence number to pinned host memory; the host spins on it
// (build: hipcc --offload-arch=gfx950 -O2 completion_latency.hip -o completion_latency)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <atomic>
__global__ void k_plain(int* out, int v) { if (v == -1) out[0] = 1; }
__global__ void k_flag(volatile unsigned* flag, unsigned seq) {
    if (threadIdx.x == 0) { __threadfence_system(); *flag = seq; }
}
int main() {
    int* d; hipMalloc(&d, 4);
    unsigned* hflag; hipHostMalloc(&hflag, 64, hipHostMallocDefault);
    unsigned* dflag; hipHostGetDevicePointer((void**)&dflag, hflag, 0);
    *hflag = 0;
    hipStream_t q; hipStreamCreateWithFlags(&q, hipStreamNonBlocking);
    hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    const int N = 2000;
    for (int rep = 0; rep < 3; rep++) {
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; i++) { hipLaunchKernelGGL(k_plain, dim3(1), dim3(64), 0, q, d, i); hipEventRecord(ev, q); hipEventSynchronize(ev); }
        auto t1 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; i++) { hipLaunchKernelGGL(k_plain, dim3(1), dim3(64), 0, q, d, i); hipStreamSynchronize(q); }
        auto t2 = std::chrono::steady_clock::now();
        unsigned seq = *hflag;
        for (int i = 0; i < N; i++) {
            seq++;
            hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, q, dflag, seq);
            while (*(volatile unsigned*)hflag != seq) { }
        }
        auto t3 = std::chrono::steady_clock::now();
        hipStreamSynchronize(q);
        auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count() / N; };
        std::printf("event %.2f us   stream sync %.2f us   flag in pinned memory %.2f us   (launch + completion, per round trip)\n", us(t0, t1), us(t1, t2), us(t2, t3));
    }
    return 0;
}
