// Bandwidth of a kernel that reads pinned host memory over the bus and writes device memory, by access shape
// (build: hipcc --offload-arch=gfx950 -O2 zero_copy_read.hip -o zero_copy_read)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
__global__ void __launch_bounds__(256) copy16(const uint4* __restrict__ s, uint4* __restrict__ d, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) d[i] = s[i];
}
__global__ void __launch_bounds__(256) copy8(const uint2* __restrict__ s, uint2* __restrict__ d, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) d[i] = s[i];
}
// rows of `w` bytes read as 144-byte pieces (18 lanes x 8 bytes), three rows per wave: the tile staging of a pyramid pass
__global__ void __launch_bounds__(256) copy_tiles(const uint8_t* __restrict__ s, uint8_t* __restrict__ d, int w, int h, int n_img) {
    const int tiles_x = (w + 127) / 128, tiles_y = (h + 31) / 32;
    const int item = blockIdx.x;
    if (item >= tiles_x * tiles_y * n_img) return;
    const int img = item / (tiles_x * tiles_y), t = item % (tiles_x * tiles_y);
    const int X0 = (t % tiles_x) * 128, Y0 = (t / tiles_x) * 32;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int rg = lane / 18, c8 = lane % 18;
    if (lane >= 54) return;
    for (int k = 0; k < 3; k++) {
        const int row = Y0 + wv * 3 + rg + 12 * k, x = X0 - 8 + 8 * c8;
        if (row < h && x >= 0 && x + 8 <= w && (c8 >= 1 && c8 <= 16)) {
            const size_t o = ((size_t)img * h + row) * w + x;
            *reinterpret_cast<uint2*>(d + o) = *reinterpret_cast<const uint2*>(s + o);
        }
    }
}
int main() {
    const int w = 752, h = 480, n = 64;
    const size_t bytes = (size_t)w * h * n;
    uint8_t *hs, *dd; hipHostMalloc((void**)&hs, bytes, hipHostMallocDefault); hipMalloc((void**)&dd, bytes);
    for (size_t i = 0; i < bytes; i++) hs[i] = (uint8_t)i;
    uint8_t* hsd; hipHostGetDevicePointer((void**)&hsd, hs, 0);
    hipStream_t q; hipStreamCreateWithFlags(&q, hipStreamNonBlocking);
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; i++) launch();
        hipStreamSynchronize(q);
        auto t0 = std::chrono::steady_clock::now();
        const int R = 20;
        for (int i = 0; i < R; i++) launch();
        hipStreamSynchronize(q);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / R;
        std::printf("%-34s %8.1f us  %6.1f GB/s\n", name, us, bytes / us / 1e3);
    };
    for (int blocks : {256, 1024, 4096})
        run(blocks == 256 ? "16 B/lane linear, 256 WGs" : blocks == 1024 ? "16 B/lane linear, 1024 WGs" : "16 B/lane linear, 4096 WGs",
            [&] { hipLaunchKernelGGL(copy16, dim3(blocks), dim3(256), 0, q, (const uint4*)hsd, (uint4*)dd, bytes / 16); });
    run("8 B/lane linear, 1024 WGs", [&] { hipLaunchKernelGGL(copy8, dim3(1024), dim3(256), 0, q, (const uint2*)hsd, (uint2*)dd, bytes / 8); });
    run("128x32 tiles, 8 B/lane", [&] { hipLaunchKernelGGL(copy_tiles, dim3(6 * 15 * n), dim3(256), 0, q, hsd, dd, w, h, n); });
    run("hipMemcpyAsync (one copy)", [&] { hipMemcpyAsync(dd, hs, bytes, hipMemcpyHostToDevice, q); });
    return 0;
}
