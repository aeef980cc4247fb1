// mb_tail.hip -- latency of the per-round serial step (k_round_tail) and its parts (tuning harness).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "host_field.hpp"
#include "kernels.cuh"
using namespace zk;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void k_empty() {}
// only the Keccak permutation, lane-parallel, `reps` times
__global__ void k_keccak_only(uint64_t* out, int reps) {
    const LaneKeccak L = lane_keccak_init();
    uint64_t a = L.index >= 0 ? (uint64_t)L.index * 0x9E3779B97F4A7C15ull : 0;
    for (int i = 0; i < reps; ++i) a = lane_keccak_f1600(a, L);
    if (L.index >= 0) out[L.index] = a;
}
// only the field conversions of the transcript step
__global__ void k_convert_only(uint64_t* out, FieldParams P, Fe x) {
    Fe c = fe_to_canonical(x, P);
    Fe xr = fe_reduce_u256(c.v, P);
    Mul29 k0, k1;
    for (int i = 0; i < 9; ++i) { k0.l[i] = P.r2_29[i]; k1.l[i] = P.r2s_29[i]; }
    Fe ch = fe_mul29(xr, k0, P), chs = fe_mul29(xr, k1, P);
    Mul29 m; split29(chs.v, m.l);
    if (threadIdx.x == 0) { fe_store(out, 0, ch); out[8] = m.l[3]; }
}

int main() {
    const FieldInfo* fi = field_info(0);
    const FieldParams P = fi->P;
    uint64_t *partials, *out; WordSponge* sp;
    CK(hipMalloc(&partials, 2048 * 3 * 32)); CK(hipMalloc(&out, 4096)); CK(hipMalloc(&sp, sizeof(WordSponge)));
    CK(hipMemset(partials, 1, 2048 * 3 * 32)); CK(hipMemset(sp, 0, sizeof(WordSponge)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto f, int reps) {
        f(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-44s %.2f us per launch\n", name, ms * 1e3 / reps);
    };
    run("empty kernel (launch-to-launch floor)", [&] { k_empty<<<1, 64>>>(); }, 200);
    run("tail: reduce 1 block x3, no transcript", [&] { k_round_tail<<<1, 256>>>(partials, 1, 3, nullptr, out, nullptr, nullptr, nullptr, P); }, 200);
    run("tail: reduce 2048 blocks x3, no transcript", [&] { k_round_tail<<<1, 256>>>(partials, 2048, 3, nullptr, out, nullptr, nullptr, nullptr, P); }, 200);
    run("tail: reduce 1 block x3 + transcript", [&] { k_round_tail<<<1, 256>>>(partials, 1, 3, sp, out, out + 64, out + 128, nullptr, P); }, 200);
    run("tail: reduce 2048 blocks x3 + transcript", [&] { k_round_tail<<<1, 256>>>(partials, 2048, 3, sp, out, out + 64, out + 128, nullptr, P); }, 200);
    run("keccak-f[1600] lane-parallel x1", [&] { k_keccak_only<<<1, 64>>>(out, 1); }, 200);
    run("keccak-f[1600] lane-parallel x11", [&] { k_keccak_only<<<1, 64>>>(out, 11); }, 200);
    run("field conversions only", [&] { k_convert_only<<<1, 64>>>(out, P, fi->two_adic_root); }, 200);
    return 0;
}
