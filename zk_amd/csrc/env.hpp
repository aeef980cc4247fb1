// env.hpp -- the library's ZK_* environment switches, parsed in one place.
//
// Every switch is a tuning / A-B / debug override read ONCE per process (function-local statics at the call sites); none of
// them selects a different implementation of the arithmetic -- only thresholds and kernel choices that all produce the same
// bits (tests/skip1_check.py, tests/test_gpu_parity.py force each of them in child processes).  The list with defaults and
// accepted ranges is in INTEGRATION.md section 7 and at the end of include/zk_amd.h.
//
// A value that does not parse as a whole decimal number, or lies outside the accepted range, is IGNORED (the default
// applies) and reported once on stderr: a mistyped switch must not silently change the kernel selection.
#pragma once
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

namespace zk {

inline uint64_t env_u64(const char *name, uint64_t dflt, uint64_t lo, uint64_t hi) {
    const char *e = getenv(name);
    if (!e) return dflt;
    char *end = nullptr;
    errno = 0;
    const unsigned long long v = strtoull(e, &end, 10);
    const bool numeric = *e != '\0' && *e != '-' && *e != '+' && end && *end == '\0' && errno == 0;
    if (!numeric || v < lo || v > hi) {
        fprintf(stderr, "zk_amd: %s=\"%s\" ignored (accepted: whole numbers %llu..%llu; default %llu applies)\n", name, e,
                (unsigned long long)lo, (unsigned long long)hi, (unsigned long long)dflt);
        return dflt;
    }
    return (uint64_t)v;
}
// set (to anything but "0" or the empty string) = on
inline bool env_flag(const char *name) {
    const char *e = getenv(name);
    return e && e[0] != '\0' && !(e[0] == '0' && e[1] == '\0');
}

}  // namespace zk
