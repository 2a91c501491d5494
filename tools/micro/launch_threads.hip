// Host cost of kernel launches from several threads at once, each on a queue of its own: does the runtime serialise them?
// (build: hipcc --offload-arch=gfx950 -pthread)
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
struct Args { int a[100]; };   // 400 B, like a tracking launch
__global__ void k(Args s, int* out) { if (s.a[0] == 12345) out[0] = 1; }
int main() {
    int* d; hipMalloc(&d, 4);
    for (int T : {1, 2, 4, 8}) {
        std::vector<hipStream_t> q(T);
        for (auto& s : q) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        std::atomic<int> go{0};
        std::vector<double> us(T);
        std::vector<std::thread> th;
        const int N = 4000;
        for (int t = 0; t < T; t++) th.emplace_back([&, t] {
            Args a{};
            for (int i = 0; i < 64; i++) hipLaunchKernelGGL(k, dim3(16), dim3(64), 0, q[t], a, d);
            hipStreamSynchronize(q[t]);
            go.fetch_add(1);
            while (go.load() < T) {}
            double in_call = 0;
            for (int i = 0; i < N; i++) {
                auto t0 = std::chrono::steady_clock::now();
                hipLaunchKernelGGL(k, dim3(16), dim3(64), 0, q[t], a, d);
                in_call += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                if ((i & 15) == 15) hipStreamSynchronize(q[t]);
            }
            us[t] = in_call / N;
        });
        auto t0 = std::chrono::steady_clock::now();
        for (auto& x : th) x.join();
        const double wall = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        double m = 0; for (double u : us) m += u / T;
        std::printf("%d threads: %.2f us inside a launch call (mean), %.2f M launches/s in all (with a sync every 16)\n", T, m, T * (double)N / wall);
        for (auto& s : q) hipStreamDestroy(s);
    }
    return 0;
}
