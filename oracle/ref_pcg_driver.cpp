// TEST INFRASTRUCTURE ONLY -- never linked into the product.
//
// Tiny driver around the reference's own, self-contained PCG headers
// (/root/reference/inst/include/pcg/pcg_random.hpp, header-only C++11).  It is
// compiled by oracle/Makefile *from the sources where they lie* into
// oracle/_ref/libref_pcg.so, and is used to pin the oracle's pcg64 restatement
// (reference engine: pcg_random.hpp:352-477, typedef pcg64 at :1675; seeding as
// in src/pcg.h:48-85 `fill_seeds` + `seeded_pcg`).
//
// Nothing else of the reference compiles without Rcpp/RcppArmadillo/RcppProgress/
// Rhtslib (absent from this image), see DESIGN.md "Oracle".
#include <cstdint>
#include <pcg/pcg_random.hpp>

extern "C" {

// `sub_seeds` are the 8 32-bit words the reference draws with
// Rcpp::runif(8, 0, 4294967296) (src/pcg.h:37-46); they are combined exactly as
// src/pcg.h:48-61 does.
void ref_pcg64_outputs(const uint32_t* sub_seeds, uint64_t n, uint64_t* out) {
    typedef pcg_extras::pcg128_t u128;
    u128 s64_1 = (static_cast<u128>(sub_seeds[0]) << 32) + sub_seeds[1];
    u128 s64_2 = (static_cast<u128>(sub_seeds[2]) << 32) + sub_seeds[3];
    u128 s64_3 = (static_cast<u128>(sub_seeds[4]) << 32) + sub_seeds[5];
    u128 s64_4 = (static_cast<u128>(sub_seeds[6]) << 32) + sub_seeds[7];
    u128 seed1 = (s64_1 << 64) + s64_2;
    u128 seed2 = (s64_3 << 64) + s64_4;
    pcg64 eng(seed1, seed2);
    for (uint64_t i = 0; i < n; i++) out[i] = eng();
}

uint64_t ref_pcg64_max(void) { return pcg64::max(); }

// The reference engine's own jump-ahead (engine::advance, pcg_random.hpp:419-434): seed, advance by
// delta = delta_hi * 2^64 + delta_lo steps, then n outputs.  Pins the advance tables create_genome uses.
void ref_pcg64_advance_outputs(const uint32_t* sub_seeds, uint64_t delta_hi, uint64_t delta_lo, uint64_t n, uint64_t* out) {
    typedef pcg_extras::pcg128_t u128;
    u128 s64_1 = (static_cast<u128>(sub_seeds[0]) << 32) + sub_seeds[1];
    u128 s64_2 = (static_cast<u128>(sub_seeds[2]) << 32) + sub_seeds[3];
    u128 s64_3 = (static_cast<u128>(sub_seeds[4]) << 32) + sub_seeds[5];
    u128 s64_4 = (static_cast<u128>(sub_seeds[6]) << 32) + sub_seeds[7];
    pcg64 eng((s64_1 << 64) + s64_2, (s64_3 << 64) + s64_4);
    eng.advance((static_cast<u128>(delta_hi) << 64) + delta_lo);
    for (uint64_t i = 0; i < n; i++) out[i] = eng();
}

}
