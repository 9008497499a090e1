// jk_common.h -- error type, HIP call check, device buffer, small helpers shared by the host code
// (part of the one translation unit jk_api.hip; see the include list there)
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "../../include/jackalope_hip.h"
#include "jk_host.h"

namespace jk {

static thread_local std::string g_last_error;

#define JK_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            throw Error(JK_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_));        \
    } while (0)

// The device arena.  Large buffers (read pools, FASTQ images, per-launch scratch, genome) are not handed back to the driver
// when a session closes but parked per device, and the next session -- the next haplotype of a sep_files call, the next
// job of the process -- takes a parked buffer of a fitting size instead of hipMalloc: the driver clears VRAM it hands out
// that is not known to be clean, at ~45 GB/s (tools/malloc_probe.hip), so a 100 GB set of pools costs seconds per
// session when allocated afresh.  Parked memory counts as free in this library's own planning (dev_mem_info); when
// hipMalloc fails the arena is emptied and the call repeated; jk_device_arena_trim() empties it on request, JK_ARENA=0
// turns it off, JK_ARENA_POISON=1 fills every reused buffer with 0xa5, =2 (or any larger number: a seed) with pseudo-random
// words; either mode also fills what hipMalloc returns (tests: nothing may rely on fresh memory being zero, or on anything
// else about it).
struct DevArena {
    struct Item { void* p; size_t n; int dev; };
    std::mutex m;
    std::vector<Item> parked;
    uint64_t hits = 0, misses = 0;
    static DevArena& get() { static DevArena* a = new DevArena(); return *a; }      // (never destroyed: the runtime may be gone first)
    static bool enabled() { static const bool on = !(std::getenv("JK_ARENA") && std::atoi(std::getenv("JK_ARENA")) == 0); return on; }
    static constexpr size_t MIN_BYTES = 8u << 20;
    // smallest parked buffer of the device with want <= n <= want + want / 4 + 64 MB
    void* take(int dev, size_t want, size_t* got) {
        std::lock_guard<std::mutex> l(m);
        int best = -1;
        for (int i = 0; i < (int)parked.size(); i++) {
            const Item& it = parked[i];
            if (it.dev != dev || it.n < want || it.n > want + want / 4 + (64u << 20)) continue;
            if (best < 0 || it.n < parked[best].n) best = i;
        }
        if (best < 0) { misses++; return nullptr; }
        void* p = parked[best].p; *got = parked[best].n;
        parked.erase(parked.begin() + best);
        hits++;
        return p;
    }
    void park(int dev, void* p, size_t n) { std::lock_guard<std::mutex> l(m); parked.push_back(Item{p, n, dev}); }
    uint64_t bytes(int dev) {
        std::lock_guard<std::mutex> l(m);
        uint64_t t = 0;
        for (const Item& it : parked) if (dev < 0 || it.dev == dev) t += it.n;
        return t;
    }
    void trim(int dev) {
        std::vector<Item> out;
        {
            std::lock_guard<std::mutex> l(m);
            std::vector<Item> keep;
            for (const Item& it : parked) (dev < 0 || it.dev == dev ? out : keep).push_back(it);
            parked.swap(keep);
        }
        int cur = 0;
        const bool have = hipGetDevice(&cur) == hipSuccess;
        for (const Item& it : out) { (void)hipSetDevice(it.dev); (void)hipFree(it.p); }
        if (have) (void)hipSetDevice(cur);
    }
};

// JK_ARENA_POISON=2: a reused buffer is filled with pseudo-random words, different for every hand-out (0xa5 bytes make every
// dword equal, which hides a reader that only trips over words that differ from each other)
__global__ void arena_poison_kernel(uint32_t* p, uint64_t n_words, uint32_t seed) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u ^ seed ^ (uint32_t)(i >> 32);
        x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12; x *= 0x297a2d39u; x ^= x >> 15;
        p[i] = x;
    }
}

struct DevBuf {
    void* p = nullptr;
    size_t n = 0;          // bytes the owner asked for
    size_t cap = 0;        // bytes of the allocation (>= n when it came out of the arena)
    int dev = -1;
    DevBuf() {}
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (!p) return;
        if (DevArena::enabled() && cap >= DevArena::MIN_BYTES && dev >= 0) DevArena::get().park(dev, p, cap);
        else (void)hipFree(p);
        p = nullptr; n = 0; cap = 0;
    }
    void alloc(size_t bytes) {
        release();
        if (bytes == 0) bytes = 16;
        int d = 0;
        JK_HIP(hipGetDevice(&d));
        dev = d;
        if (DevArena::enabled() && bytes >= DevArena::MIN_BYTES) {
            size_t got = 0;
            if (void* q = DevArena::get().take(d, bytes, &got)) {
                p = q; n = bytes; cap = got;
                poison_fill();
                return;
            }
        }
        // (JK_TIMING: large allocations are timed on their own -- the driver clears VRAM it hands out that is not
        // known to be clean, at ~45 GB/s, so a run that needs more than the clean part of HBM waits seconds here:
        // tools/malloc_probe.hip)
        const bool timed = bytes >= (1ull << 30) && std::getenv("JK_TIMING") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess && DevArena::get().bytes(d) > 0) {        // the arena holds what this allocation needs
            (void)hipGetLastError();
            DevArena::get().trim(d);
            e = hipMalloc(&p, bytes);
        }
        if (e != hipSuccess) { p = nullptr; throw Error(JK_ERR_DEVICE, std::string("hipMalloc of ") + std::to_string(bytes >> 20) + " MiB: " + hipGetErrorString(e)); }
        if (timed) std::fprintf(stderr, "[jk timing]   hipMalloc of %6.1f GB      %8.1f ms\n", bytes / 1e9, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        n = bytes; cap = bytes;
        poison_fill();              // (small buffers never see the arena, and hipMalloc hands back what this process freed a moment ago)
    }
    void poison_fill() {
        static const int poison = std::getenv("JK_ARENA_POISON") ? std::atoi(std::getenv("JK_ARENA_POISON")) : 0;
        if (!poison) return;
        if (poison == 1) JK_HIP(hipMemset(p, 0xa5, cap));
        else {
            static std::atomic<uint32_t> round{0};
            const uint64_t nw = (uint64_t)cap / 4;
            hipLaunchKernelGGL(arena_poison_kernel, dim3((unsigned)std::min<uint64_t>(4096, (nw + 255) / 256 + 1)), dim3(256), 0, 0, static_cast<uint32_t*>(p), nw,
                               0x9e3779b9u * (round.fetch_add(1) + (uint32_t)poison));
            JK_HIP(hipGetLastError());
        }
        JK_HIP(hipDeviceSynchronize());
    }
    void swap(DevBuf& o) { std::swap(p, o.p); std::swap(n, o.n); std::swap(cap, o.cap); std::swap(dev, o.dev); }
    template <typename T> T* as() const { return static_cast<T*>(p); }
    template <typename T> void upload(const std::vector<T>& v) { upload(v.data(), v.size()); }
    template <typename T> void upload(const T* v, size_t count) {
        alloc(count * sizeof(T));
        if (count) JK_HIP(hipMemcpy(p, v, count * sizeof(T), hipMemcpyHostToDevice));
    }
};
// free / total device memory as this library's planning sees it: what is parked in the arena is as good as free
static inline void dev_mem_info(size_t* free_b, size_t* total_b) {
    JK_HIP(hipMemGetInfo(free_b, total_b));
    int d = 0;
    if (DevArena::enabled() && hipGetDevice(&d) == hipSuccess) *free_b += (size_t)DevArena::get().bytes(d);
}

// JK_TIMING=1: wall-clock of the set-up phases on stderr (where does open() spend its time?)
struct PhaseTimer {
    const char* what; std::chrono::steady_clock::time_point t0; bool on;
    explicit PhaseTimer(const char* w) : what(w), t0(std::chrono::steady_clock::now()), on(std::getenv("JK_TIMING") != nullptr) {}
    ~PhaseTimer() {
        if (on) std::fprintf(stderr, "[jk timing] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
};

constexpr int JK_ERR_RETRY = 1000;   // internal: PacBio pools were too small, regenerate with larger ones
constexpr int JK_ERR_RETRY_IMAGE = 1001;   // internal: the PacBio image (sized for the expected read length) was too small

static inline uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }
static inline uint32_t n_digits(uint64_t v) { uint32_t d = 1; while (v >= 10) { v /= 10; d++; } return d; }

struct Batch {
    uint64_t lane0;        // first lane (relative to the shard)
    uint32_t n_lanes;
    uint64_t pool_bytes;   // per read end
    uint64_t n_reads;      // planned reads of its lanes (progress reporting)
};

template <typename F>
static int guarded(F f) {
    try { f(); g_last_error.clear(); return JK_OK; }
    catch (const Error& e) { g_last_error = e.what(); return e.code; }
    catch (const std::bad_alloc&) { g_last_error = "out of host memory"; return JK_ERR_DEVICE; }
    catch (const std::exception& e) { g_last_error = e.what(); return JK_ERR_ARG; }
}


}  // namespace jk
