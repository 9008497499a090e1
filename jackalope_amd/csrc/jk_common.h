// jk_common.h -- error type, HIP call check, device buffer, small helpers shared by the host code
// (part of the one translation unit jk_api.hip; see the include list there)
#pragma once
#include <hip/hip_runtime.h>
#include <chrono>
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

struct DevBuf {
    void* p = nullptr;
    size_t n = 0;
    DevBuf() {}
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() { if (p) { (void)hipFree(p); p = nullptr; n = 0; } }
    void alloc(size_t bytes) {
        release();
        if (bytes == 0) bytes = 16;
        // (JK_TIMING: large allocations are timed on their own -- the driver clears VRAM it hands out that is not
        // known to be clean, at ~45 GB/s, so a run that needs more than the clean part of HBM waits seconds here:
        // tools/malloc_probe.hip)
        const bool timed = bytes >= (1ull << 30) && std::getenv("JK_TIMING") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        JK_HIP(hipMalloc(&p, bytes));
        if (timed) std::fprintf(stderr, "[jk timing]   hipMalloc of %6.1f GB      %8.1f ms\n", bytes / 1e9, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        n = bytes;
    }
    template <typename T> T* as() const { return static_cast<T*>(p); }
    template <typename T> void upload(const std::vector<T>& v) { upload(v.data(), v.size()); }
    template <typename T> void upload(const T* v, size_t count) {
        alloc(count * sizeof(T));
        if (count) JK_HIP(hipMemcpy(p, v, count * sizeof(T), hipMemcpyHostToDevice));
    }
};

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
