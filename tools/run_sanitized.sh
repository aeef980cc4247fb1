#!/bin/bash
# tools/run_sanitized.sh -- the CPU suite under AddressSanitizer + UBSan.  CPU box ONLY (gpurun refuses GPU sanitizer runs).
#
#   make -C oracle asan            -> oracle/libzk_oracle_asan.so   (the checker: 1,1k lines of C)
#   make -C zk_amd/csrc asan       -> zk_amd/libzk_amd_asan.so      (host side of the product library; device code not instrumented)
#   LD_PRELOAD=clang's libclang_rt.asan-x86_64.so  ZK_ORACLE_LIB=...  ZK_AMD_LIB=...  python -m pytest tests -m "not gpu"
#
# The python interpreter is not instrumented, so the runtime is preloaded; leak detection is off (CPython keeps its arenas) and
# tests/test_gpu_cpp_host.py is left out (it shells out to `make`, which refuses the preload, and its binaries link the normal
# library).  Output: profiles/r05_asan_cpu.log (or $1).  Exit code = pytest's; any sanitizer report aborts the run
# (-fno-sanitize-recover, halt_on_error).
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
LOG="${1:-$ROOT/profiles/r05_asan_cpu.log}"
RT="$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)"
make -C "$ROOT/oracle" asan
make -j2 -C "$ROOT/zk_amd/csrc" asan
cd "$ROOT"
{
  echo "# $(date -u +%FT%TZ)  HEAD $(git rev-parse --short HEAD)  runtime $RT"
  echo "# oracle: clang -O1 -g -fsanitize=address,undefined; library host side: hipcc -O1 -g -fsanitize=address,undefined -fno-gpu-sanitize"
  env LD_PRELOAD="$RT" ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0:detect_odr_violation=0" \
      UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1" \
      ZK_ORACLE_LIB="$ROOT/oracle/libzk_oracle_asan.so" ZK_AMD_LIB="$ROOT/zk_amd/libzk_amd_asan.so" \
      python3 -m pytest tests -q -m "not gpu" -p no:cacheprovider --ignore tests/test_gpu_cpp_host.py 2>&1
  rc=$?
  echo "# pytest exit code $rc; sanitizer reports: $(grep -c 'ERROR: AddressSanitizer\|runtime error:' "$LOG" 2>/dev/null || true)"
  exit $rc
} | tee "$LOG"
