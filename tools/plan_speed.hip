// Host-side planner speed probe (no device work): BASELINE configs[2] shape, 2^21 lanes x 4 haplotypes x 24 chromosomes,
// 143 pairs per lane, deferred chromosome splits.   hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -o /tmp/plan_speed tools/plan_speed.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include "../jackalope_amd/csrc/jk_plan.h"
using namespace jk;
int main(int argc, char** argv) {
    const uint64_t T = 1ull << 21, nh = 4, nc = 24;
    std::vector<uint32_t> words(8 * T * (3 + 2 * nh) + 64);
    uint64_t x = 88172645463325252ull;
    for (auto& w : words) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; w = (uint32_t)(x >> 11); }
    QuotaModel Q; Q.hap = true; Q.n_ends = 2; Q.maker_halves = true; Q.n_haps = nh; Q.n_chroms = nc;
    Q.hap_chain = GroupChain(std::vector<double>(nh, 1.0));
    for (uint64_t h = 0; h < nh; h++) Q.chrom_chain.emplace_back(std::vector<double>(nc, 125e6));
    std::vector<uint64_t> per_lane(T, 286);
    printf("hardware_concurrency %u\n", std::thread::hardware_concurrency());
    for (int rep = 0; rep < 2; rep++) {
        jk_seed_source src{words.data(), words.size(), nullptr, nullptr};
        SeedReader r{src};
        auto t0 = std::chrono::steady_clock::now();
        LanePlan lp = plan_lane_quotas(Q, per_lane, 0, T, r, false, 0, true);
        auto t1 = std::chrono::steady_clock::now();
        printf("threads %s: %.3f s (tasks %llu)\n", getenv("JK_HOST_THREADS") ? getenv("JK_HOST_THREADS") : "all", std::chrono::duration<double>(t1 - t0).count(), (unsigned long long)lp.n_tasks());
    }
}
