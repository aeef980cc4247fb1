// zeta_sort.hip -- the term list of CoeffMultilinearPolynomial::to_evaluation_form (coefficient_form.rs:340-347) ordered on the device.
//
// k_zeta_first (zeta_kernels.cuh) wants the terms ascending by TABLE INDEX (key bit v <-> variable v <-> index bit n-1-v: the bit-reversed
// key).  Up to a few thousand terms the host orders them while it merges duplicates (capi.hip); a long list (2^16 terms: 3.5 ms of
// std::sort + merge + pageable upload on the host, six times the transform itself) is uploaded AS GIVEN and ordered here: one kernel
// turns keys into table indices, rocPRIM's radix sort orders (index, position) pairs, and k_zeta_first reads the coefficients through the
// permutation, adding up runs of equal indices itself (BTreeMap semantics: duplicate terms are summed, coefficient_form.rs:164-171).
// Own translation unit: rocPRIM's headers are heavy and nothing else needs them.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

namespace zk {

__global__ void k_term_indices(const uint64_t *__restrict__ keys, uint64_t n, uint32_t n_vars, uint64_t *__restrict__ idx, uint32_t *__restrict__ pos) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    idx[t] = __brevll(keys[t]) >> (64 - n_vars);
    pos[t] = (uint32_t)t;
}

// keys (device, as the caller listed them) -> idx_sorted (ascending table indices) + perm (position of each in the caller's list).
// temp == nullptr: only *temp_bytes is set (the scratch the sort needs).  Returns a hipError_t as int.
int zeta_sort_terms(hipStream_t stream, const uint64_t *d_keys, uint64_t n, uint32_t n_vars, uint64_t *d_idx_unsorted, uint32_t *d_pos_unsorted,
                    uint64_t *d_idx_sorted, uint32_t *d_perm, void *temp, size_t *temp_bytes) {
    if (!temp) {
        return (int)rocprim::radix_sort_pairs(nullptr, *temp_bytes, d_idx_unsorted, d_idx_sorted, d_pos_unsorted, d_perm, (size_t)n, 0u, n_vars, stream);
    }
    k_term_indices<<<(uint32_t)((n + 255) / 256), 256, 0, stream>>>(d_keys, n, n_vars, d_idx_unsorted, d_pos_unsorted);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    return (int)rocprim::radix_sort_pairs(temp, *temp_bytes, d_idx_unsorted, d_idx_sorted, d_pos_unsorted, d_perm, (size_t)n, 0u, n_vars, stream);
}

}  // namespace zk
