// test scaffolding -- see README.md (not Rcpp)
#pragma once
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
typedef void* SEXP;
namespace rcpp_stub {
// what stands in for R's RNG: a queue of 32-bit words handed out by runif(n, 0, 2^32) as doubles with a fraction
extern std::vector<uint32_t> seed_queue;
extern size_t seed_pos;
extern int runif_thread_violations;      // runif called off the "R main thread"
extern int warnings;                     // Rcpp::warning calls
void mark_main_thread();
bool on_main_thread();
}
namespace Rcpp {
class exception : public std::runtime_error {
public:
    exception(const char* m, bool = true) : std::runtime_error(m) {}
};
inline void stop(const std::string& m) { throw exception(m.c_str(), false); }
template <typename... A> inline void warning(const char*, A...) { rcpp_stub::warnings++; }
struct NumericVector {
    std::vector<double> v;
    double operator[](int i) const { return v[(size_t)i]; }
};
inline NumericVector runif(int n, double lo, double hi) {
    (void)lo; (void)hi;
    if (!rcpp_stub::on_main_thread()) rcpp_stub::runif_thread_violations++;
    NumericVector out;
    for (int i = 0; i < n; i++) {
        if (rcpp_stub::seed_pos >= rcpp_stub::seed_queue.size()) throw exception("stub RNG exhausted", false);
        out.v.push_back((double)rcpp_stub::seed_queue[rcpp_stub::seed_pos++] + 0.25);
    }
    return out;
}
template <typename T> class XPtr {
    T* p;
public:
    explicit XPtr(SEXP s) : p(static_cast<T*>(s)) {}
    T& operator*() const { return *p; }
    T* operator->() const { return p; }
};
}  // namespace Rcpp
