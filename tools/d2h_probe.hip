// Raw device-to-host copy rate into pinned memory, by piece size and copies in flight (what the streaming sink can hope for).
//   hipcc -O2 --offload-arch=gfx950 -o /tmp/d2h_probe tools/d2h_probe.hip && /tmp/d2h_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t total = 8ull << 30;
    uint8_t* d; CK(hipMalloc(&d, total)); CK(hipMemset(d, 7, total));
    for (size_t piece : {8ull << 20, 33ull << 20, 128ull << 20, 512ull << 20}) {
        for (int inflight : {1, 2, 4}) {
            std::vector<uint8_t*> h(inflight); std::vector<hipStream_t> st(inflight);
            for (int i = 0; i < inflight; i++) { CK(hipHostMalloc(&h[i], piece, hipHostMallocDefault)); CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking)); }
            CK(hipDeviceSynchronize());
            const double t0 = now();
            size_t off = 0; int k = 0;
            while (off + piece <= total) { CK(hipMemcpyAsync(h[k % inflight], d + off, piece, hipMemcpyDeviceToHost, st[k % inflight])); off += piece; k++; }
            CK(hipDeviceSynchronize());
            const double dt = now() - t0;
            printf("piece %4zu MB, %d in flight: %.1f GB/s\n", piece >> 20, inflight, off / 1e9 / dt);
            fflush(stdout);
            for (int i = 0; i < inflight; i++) { CK(hipHostFree(h[i])); CK(hipStreamDestroy(st[i])); }
        }
    }
    return 0;
}
