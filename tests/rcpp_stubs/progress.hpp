// test scaffolding -- see README.md (not RcppProgress)
#pragma once
#include <cstdint>
namespace rcpp_stub {
extern unsigned long long progress_max, progress_shown;
extern int progress_thread_violations;
extern unsigned long long abort_after;        // check_abort() turns true once this much progress was shown (0 = never)
bool on_main_thread();
}
class Progress {
public:
    Progress(unsigned long long max, bool) { rcpp_stub::progress_max = max; rcpp_stub::progress_shown = 0; }
    void increment(unsigned long long n) { if (!rcpp_stub::on_main_thread()) rcpp_stub::progress_thread_violations++; rcpp_stub::progress_shown += n; }
    static bool check_abort() {
        if (!rcpp_stub::on_main_thread()) rcpp_stub::progress_thread_violations++;
        return rcpp_stub::abort_after && rcpp_stub::progress_shown >= rcpp_stub::abort_after;
    }
};
