#!/bin/bash
# The host side of the library under AddressSanitizer, UBSan and (planner threads) ThreadSanitizer -- the GPU pool allows
# sanitizers on the CPU build only:
# builds variants/lib_asan.so and lib_ubsan.so with the sanitizer on the host compilation, and runs the CPU test suite
# (planner, host primitives, mutation-table builder, boundary shims, ABI) with each.      tools/sanitize_cpu.sh
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/variants"
rt=$(dirname "$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)")
cd "$root/jackalope_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -ffp-contract=off -fPIC -shared -Wno-literal-range \
    -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer -o "$root/variants/lib_asan.so" jk_api.hip -lz -lpthread
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -ffp-contract=off -fPIC -shared -Wno-literal-range \
    -Xarch_host -fsanitize=undefined -Xarch_host -fno-sanitize-recover=undefined -o "$root/variants/lib_ubsan.so" jk_api.hip -lz -lpthread
cd "$root"
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 LD_PRELOAD=$rt/libclang_rt.asan-x86_64.so JK_HIP_LIB=$root/variants/lib_asan.so python -m pytest tests -x -q -m "not gpu"
LD_PRELOAD=$rt/libclang_rt.ubsan_standalone-x86_64.so JK_HIP_LIB=$root/variants/lib_ubsan.so python -m pytest tests -x -q -m "not gpu"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -ffp-contract=off -fPIC -shared -Wno-literal-range \
    -Xarch_host -fsanitize=thread -o "$root/variants/lib_tsan.so" "$root/jackalope_amd/csrc/jk_api.hip" -lz -lpthread
TSAN_OPTIONS=report_signal_unsafe=0 LD_PRELOAD=$rt/libclang_rt.tsan-x86_64.so JK_HIP_LIB=$root/variants/lib_tsan.so JK_HOST_THREADS=6 \
    python -m pytest tests/test_plan_cpu.py tests/test_host_primitives.py -x -q 2>&1 | grep -E "ThreadSanitizer|passed|failed"
