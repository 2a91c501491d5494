// Can the host write a tracking batch's input block straight into device memory (fine-grained allocation reached through
// the PCIe BAR), so that no kernel or copy has to fetch it from pinned host memory?  Prints whether the pointer is
// host-writable, the time of a 40 KB host write, and whether a kernel then reads what was written.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdint>
#include <vector>

__global__ void sum_kernel(const uint32_t* p, int n, unsigned long long* out) {
    unsigned long long s = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += p[i];
    atomicAdd(out, s);
}

int main() {
    const size_t bytes = 40 * 1024;
    void* dev = nullptr;
    hipError_t e = hipExtMallocWithFlags(&dev, bytes, hipDeviceMallocFinegrained);
    std::printf("hipExtMallocWithFlags(finegrained): %s\n", hipGetErrorString(e));
    if (e != hipSuccess) return 0;
    hipPointerAttribute_t at;
    e = hipPointerGetAttributes(&at, dev);
    std::printf("attributes: %s type %d device ptr %p host ptr %p\n", hipGetErrorString(e), (int)at.type, at.devicePointer, at.hostPointer);
    int large_bar = -1;
    (void)hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, 0);
    std::printf("large BAR attribute: %d\n", large_bar);
    if (large_bar != 1) { std::printf("no large BAR: host writes not attempted\n"); return 0; }
    unsigned long long* out = nullptr;
    hipMalloc(&out, 8);
    std::vector<uint32_t> src(bytes / 4);
    for (size_t i = 0; i < src.size(); i++) src[i] = (uint32_t)(i * 2654435761u);
    unsigned long long want = 0;
    for (uint32_t v : src) want += v;
    for (int rep = 0; rep < 5; rep++) {
        hipMemset(out, 0, 8);
        hipDeviceSynchronize();
        const auto t0 = std::chrono::steady_clock::now();
        std::memcpy(dev, src.data(), bytes);
        __sync_synchronize();
        const auto t1 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, 0, (const uint32_t*)dev, (int)src.size(), out);
        unsigned long long got = 0;
        hipMemcpy(&got, out, 8, hipMemcpyDeviceToHost);
        std::printf("host write of %zu bytes: %.2f us, kernel read %s\n", bytes, std::chrono::duration<double, std::micro>(t1 - t0).count(), got == want ? "matches" : "DIFFERS");
        src[rep] += 1; want += 1;
    }
    return 0;
}
