// How long does hipMalloc take for the sizes a whole-image-resident run asks for?   hipcc -O2 --offload-arch=gfx950 -o /tmp/malloc_probe tools/malloc_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipFree(nullptr);
    const double GB = 1e9;
    for (int round = 0; round < 2; round++) {
        for (double gb : {1.0, 8.0, 32.0, 64.0, 100.0, 200.0, 250.0}) {
            void* p = nullptr;
            double t0 = now();
            hipError_t e = hipMalloc(&p, (size_t)(gb * GB));
            double t1 = now();
            if (e != hipSuccess) { printf("%.0f GB: %s\n", gb, hipGetErrorString(e)); continue; }
            hipMemsetAsync(p, 0, 64, 0); hipDeviceSynchronize();
            double t2 = now();
            hipFree(p);
            double t3 = now();
            printf("round %d: %5.0f GB  malloc %8.1f ms  first touch %6.1f ms  free %8.1f ms\n", round, gb, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3);
        }
        // the same 200 GB as 4 pieces of 50 GB
        std::vector<void*> ps(4, nullptr);
        double t0 = now();
        for (auto& p : ps) hipMalloc(&p, (size_t)(50 * GB));
        double t1 = now();
        for (auto& p : ps) hipFree(p);
        printf("round %d: 4 x 50 GB malloc %8.1f ms free %8.1f ms\n", round, (t1 - t0) * 1e3, (now() - t1) * 1e3);
    }
}
