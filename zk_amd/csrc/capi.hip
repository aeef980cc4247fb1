// capi.hip -- the C ABI of libzk_amd.so (include/zk_amd.h): host-side protocol logic + kernel launches.
//
// Host logic restated from the reference (paths relative to the reference checkout):
//   sumcheck/src/prover.rs:33-73 (round loop, absorb order), sumcheck/src/lib.rs:23-29 (32-byte BE elements),
//   transcript/src/lib.rs:16-30, sumcheck/src/verifier.rs:15-78, polynomial/src/univariate_poly.rs:29-80,
//   fft/src/lib.rs:4-19.  There is no CPU compute fallback: every table operation is a gfx950 kernel.
#include <hip/hip_runtime.h>
#include <sched.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <cstdio>
#include <map>
#include <new>
#include <set>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/zk_amd.h"
#include "host_field.hpp"
#include "env.hpp"
#include "kernels.cuh"
#include "eval_kernels.cuh"
#include "zeta_kernels.cuh"
#include "gkr_kernels.cuh"
#include "launch.hpp"
#include "ntt_kernels.cuh"
#include "keccak.hpp"

using namespace zk;

// ------------------------------------------------------------------------------------------------------------
// objects
// ------------------------------------------------------------------------------------------------------------
struct zk_ctx {
    int field;
    int device;
    const FieldInfo *fi;
    hipStream_t own_stream;
    hipStream_t stream;
    uint64_t *d_partials;   // per-block partial sums of a round: kMaxGrid * kMaxSums elements
    uint64_t *d_sums;       // final round sums (kMaxSums elements) + lanes area
    uint64_t *h_pinned;     // pinned staging: kMaxSums*8 u64
    uint32_t *h_flag;       // completion word in pinned memory: the last kernel of a call stores flag_seq there (host_flag_wait)
    uint32_t flag_seq;
    uint8_t *h_results;     // pinned staging for proofs (grown on demand)
    std::map<uint32_t, uint64_t *> lagrange_w;   // interp_weights(D) in device memory, cached (a field inversion per node)
    size_t h_results_bytes;
    hipEvent_t ev0, ev1;
    std::map<std::pair<uint32_t, int>, uint64_t *> twiddles;   // (log_n, inverse) -> omega^i table, i < n/2 (n < 2^8 path)
    std::map<std::pair<uint32_t, int>, NttPlan> ntt_plans;     // (log_n, inverse) -> pass plan + two-level twiddle tables
    std::map<size_t, std::vector<void *>> pool;                // freed device blocks by exact size (stream-ordered reuse)
    size_t pool_bytes, pool_checked;
    Fe inv2;                // 1/2 (pipelined rounds interpolate on the nodes 0, 1, -1, inf)
    PipeConsts pipe_consts; // its prepared multiplier + the constant 2^266 mod p (pipe_kernels.cuh pipe_eval_canon)
    uint64_t *d_dbg;        // ZK_PIPE_DEBUG: phase timestamps of the pipelined launches (64 launches x 32 slots + finisher)
    uint32_t dbg_launch;
    uint8_t *h_absorb[2];   // pinned staging of absorb_tables (prove / verify), kept across calls
    size_t h_absorb_bytes;
    hipEvent_t ev_absorb[2];
};
struct zk_mle {
    zk_ctx *ctx;
    uint64_t n_vars;
    uint64_t *d;
};
struct zk_transcript {
    Sponge sp;
};

static constexpr uint32_t kMaxGrid = 2048;    // round kernels: 8 workgroups per CU on 256 CUs (partials are sized for it)
static constexpr uint32_t kMaxGridStream = 16384;   // pure streaming kernels (fold): measured +10% over 2048 at 2^24
static constexpr uint32_t kMaxSums = 256;     // max_var_degree is a u8 in the reference (prover.rs:9)
static constexpr uint64_t kMaxVars = 40;

static thread_local std::string g_hip_err;

#define HIPCHK(expr)                                                                      \
    do {                                                                                  \
        hipError_t e__ = (expr);                                                          \
        if (e__ != hipSuccess) {                                                          \
            g_hip_err = std::string(#expr) + ": " + hipGetErrorString(e__);               \
            return ZK_ERR_HIP;                                                            \
        }                                                                                 \
    } while (0)
#define ZKCHK(expr)                        \
    do {                                   \
        int32_t rc__ = (expr);             \
        if (rc__ != ZK_OK) return rc__;    \
    } while (0)

static inline uint32_t grid_for(uint64_t items) {
    uint64_t b = (items + kBlock - 1) / kBlock;
    if (b < 1) b = 1;
    if (b > kMaxGrid) b = kMaxGrid;
    return (uint32_t)b;
}
// Wait for the stream.  (Until round 3 this spun on hipStreamQuery first; tools/mb/mb_flag.hip: that costs 2-4 us MORE than
// hipStreamSynchronize for kernels of 1 us .. 1.2 ms, and a completion word in pinned memory -- host_flag_wait below -- 5 us less.)
static inline hipError_t stream_wait(hipStream_t s) { return hipStreamSynchronize(s); }
// ---- completion word in pinned host memory -----------------------------------------------------------------------------------
// The last kernel of a call copies the results into pinned host memory itself and then stores a sequence number next to them
// (system-scope fence in between); the host spins on that word instead of waiting for the stream's completion signal, which
// arrives ~5 us later (tools/mb/mb_flag.hip: launch + wait of a 1-us kernel 11.7 us with hipStreamSynchronize, 6.9 us with the
// word).  The stream itself is checked every few thousand spins so that a failed launch ends the wait with its error.
__global__ __launch_bounds__(64) void k_publish_host(const uint64_t *__restrict__ src, uint64_t *__restrict__ dst_host, uint32_t n_u64,
                                                     volatile uint32_t *flag, uint32_t seq) {
    // ONE wave: its lanes' stores, then one system-scope fence for the whole wave, then the word -- no workgroup barrier
    for (uint32_t i = 2 * threadIdx.x; i < n_u64; i += 2 * 64)   // n_u64 is even: elements are 4 words
        *reinterpret_cast<uint4 *>(dst_host + i) = *reinterpret_cast<const uint4 *>(src + i);
    __threadfence_system();
    if (threadIdx.x == 0) *flag = seq;
}
static constexpr unsigned kPolledHostFlags = hipHostMallocCoherent | hipHostMallocMapped;
// next completion sequence number; 0 is reserved ("this launch writes no word"), so it is skipped when the counter wraps
static inline uint32_t next_flag_seq(zk_ctx *c) {
    if (++c->flag_seq == 0) ++c->flag_seq;
    return c->flag_seq;
}
static int32_t host_flag_wait(zk_ctx *c, uint32_t seq, uint32_t slot = 0) {
    volatile uint32_t *flag = c->h_flag + 16 * slot;
    for (uint32_t spins = 1;; ++spins) {
        if (*flag == seq) break;
        if ((spins & 0x3FFF) == 0) {
            const hipError_t e = hipStreamQuery(c->stream);
            if (e == hipSuccess) break;                  // the stream has drained: the kernel's stores are visible
            if (e != hipErrorNotReady) {
                g_hip_err = std::string("stream failed while waiting for the completion word: ") + hipGetErrorString(e);
                return ZK_ERR_HIP;
            }
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return ZK_OK;
}
static inline int32_t use_device(const zk_ctx *ctx) {
    HIPCHK(hipSetDevice(ctx->device));
    return ZK_OK;
}
// Device blocks are recycled through a per-context free list keyed by size (tables are powers of two, so the hit
// rate is high): hipMalloc/hipFree of a 256 MiB block costs milliseconds and synchronises the device, which is more
// than a whole 2^24 fold.  All use is ordered on the context's stream, so a recycled block is safe to hand out again
// without a synchronisation.
static void pool_trim(zk_ctx *c) {   // hipFree synchronises the device, so blocks still in use by queued work are safe
    for (auto &kv : c->pool)
        for (void *q : kv.second) (void)hipFree(q);
    c->pool.clear();
    c->pool_bytes = 0;
}
static int32_t pool_alloc(zk_ctx *c, size_t bytes, void **out) {
    if (bytes == 0) bytes = 32;
    auto it = c->pool.find(bytes);
    if (it != c->pool.end() && !it->second.empty()) {
        *out = it->second.back();
        it->second.pop_back();
        c->pool_bytes -= bytes;
        return ZK_OK;
    }
    hipError_t e = hipMalloc(out, bytes);
    if (e != hipSuccess) {   // out of memory: drop the cache and retry once
        (void)hipGetLastError();
        pool_trim(c);
        e = hipMalloc(out, bytes);
    }
    if (e != hipSuccess) {
        g_hip_err = std::string("hipMalloc: ") + hipGetErrorString(e);
        return ZK_ERR_ALLOC;
    }
    return ZK_OK;
}
static void pool_free(zk_ctx *c, void *ptr, size_t bytes) {
    if (!ptr) return;
    if (bytes == 0) bytes = 32;
    c->pool[bytes].push_back(ptr);
    c->pool_bytes += bytes;
    // cap: idle blocks never hold more than half of what the device has left (other contexts, other libraries and this
    // library's few raw hipMallocs must not fail while gigabytes sit here).  Checked only past 1 GiB: hipMemGetInfo is slow.
    // (pool_alloc lowers pool_bytes without touching pool_checked: compare before subtracting, the difference is unsigned)
    if (c->pool_bytes > ((size_t)1 << 30) && c->pool_bytes > c->pool_checked && c->pool_bytes - c->pool_checked > ((size_t)1 << 30)) {
        size_t fr = 0, tot = 0;
        c->pool_checked = c->pool_bytes;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess && c->pool_bytes > fr / 2) {
            pool_trim(c);
            c->pool_checked = 0;
        }
    }
}
// scratch that does not fit the pool's power-of-two habits still goes through it, so an allocation failure drops the cache
// and retries instead of failing while the pool sits on idle gigabytes
static int32_t raw_alloc(zk_ctx *c, size_t bytes, void **out) {
    hipError_t e = hipMalloc(out, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        pool_trim(c);
        e = hipMalloc(out, bytes);
    }
    if (e != hipSuccess) {
        g_hip_err = std::string("hipMalloc: ") + hipGetErrorString(e);
        return ZK_ERR_ALLOC;
    }
    return ZK_OK;
}

// ------------------------------------------------------------------------------------------------------------
// library
// ------------------------------------------------------------------------------------------------------------
extern "C" int32_t zk_abi_version(void) { return ZK_AMD_ABI_VERSION; }

extern "C" const char *zk_strerror(int32_t s) {
    switch (s) {
        case ZK_OK: return "ok";
        case ZK_ERR_EVAL_LEN: return "evaluation vec len should equal 2^n_vars";
        case ZK_ERR_EVAL_ARITY: return "evaluate must assign to all variables";
        case ZK_ERR_EMPTY_PRODUCT: return "cannot create product polynomial from empty polynomials";
        case ZK_ERR_ARITY_MISMATCH:
            return "cannot create product polynomial from polynomial that don't share the same number of variables";
        case ZK_ERR_PANIC_INDEX: return "reference panics: index arithmetic underflow (initial_var / assignments out of range)";
        case ZK_ERR_FFT_NOT_POW2: return "values must be a power of 2";
        case ZK_ERR_FFT_NO_ROOT: return "reference panics: get_root_of_unity returned None";
        case ZK_ERR_VERIFY_ROUNDS: return "invalid proof: require 1 round poly for each variable in poly";
        case ZK_ERR_VERIFY_SUM: return "verifier check failed: claimed_sum != p(0) + p(1)";
        case ZK_ERR_COEFF_RANGE: return "coefficient map represents more than specificed number of variables";
        case ZK_ERR_BAD_ARG: return "bad argument";
        case ZK_ERR_BAD_FIELD: return "unknown field id";
        case ZK_ERR_NO_DEVICE: return "no usable gfx950 device (libzk_amd has no CPU fallback)";
        case ZK_ERR_HIP: return "HIP runtime error (see zk_last_hip_error)";
        case ZK_ERR_ALLOC: return "allocation failed";
        case ZK_ERR_UNSUPPORTED: return "unsupported configuration";
        case ZK_ERR_CONTEXT_MISMATCH: return "handle belongs to a different context";
        case ZK_ERR_GKR_REJECT: return "gkr verifier check failed: layer wiring / input claim mismatch";
        case ZK_ERR_COMM: return "collective failed (see zk_last_hip_error)";
        default: return "unknown status";
    }
}
extern "C" const char *zk_last_hip_error(void) { return g_hip_err.c_str(); }

extern "C" int32_t zk_device_count(int32_t *out) {
    if (!out) return ZK_ERR_BAD_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *out = n;
    return ZK_OK;
}

// ------------------------------------------------------------------------------------------------------------
// field helpers (host)
// ------------------------------------------------------------------------------------------------------------
extern "C" int32_t zk_field_modulus(int32_t field, uint64_t out[4]) {
    const uint64_t *m = field_modulus_limbs(field);
    if (!m || !out) return m ? ZK_ERR_BAD_ARG : ZK_ERR_BAD_FIELD;
    memcpy(out, m, 32);
    return ZK_OK;
}
extern "C" int32_t zk_field_two_adicity(int32_t field, int32_t *out) {
    const FieldInfo *fi = field_info(field);
    if (!fi) return ZK_ERR_BAD_FIELD;
    if (!out) return ZK_ERR_BAD_ARG;
    *out = (int32_t)fi->two_adicity;
    return ZK_OK;
}
// F::get_root_of_unity(2^log_n) (fft/src/lib.rs:6): None -> ZK_ERR_FFT_NO_ROOT
extern "C" int32_t zk_field_root_of_unity(int32_t field, uint64_t log_n, uint64_t out[4]) {
    const FieldInfo *fi = field_info(field);
    if (!fi) return ZK_ERR_BAD_FIELD;
    if (!out) return ZK_ERR_BAD_ARG;
    Fe w;
    if (log_n > 64 || !field_root_of_unity(*fi, (uint32_t)log_n, w)) return ZK_ERR_FFT_NO_ROOT;
    fe_to_u64limbs(w, out);
    return ZK_OK;
}
extern "C" int32_t zk_fe_from_u64(int32_t field, uint64_t v, uint64_t out[4]) {
    const FieldInfo *fi = field_info(field);
    if (!fi) return ZK_ERR_BAD_FIELD;
    uint64_t l[4] = {v, 0, 0, 0};
    fe_to_u64limbs(fe_from_canonical(fe_from_u64limbs(l), fi->P), out);
    return ZK_OK;
}
extern "C" int32_t zk_fe_from_canonical(int32_t field, const uint64_t limbs[4], uint64_t out[4]) {
    const FieldInfo *fi = field_info(field);
    if (!fi) return ZK_ERR_BAD_FIELD;
    Fe c = fe_from_u64limbs(limbs), d;
    if (!sub8(d.v, c.v, fi->P.p)) return ZK_ERR_BAD_ARG;   // not < p
    fe_to_u64limbs(fe_from_canonical(c, fi->P), out);
    return ZK_OK;
}
extern "C" int32_t zk_fe_to_canonical(int32_t field, const uint64_t a[4], uint64_t out[4]) {
    const FieldInfo *fi = field_info(field);
    if (!fi) return ZK_ERR_BAD_FIELD;
    fe_to_u64limbs(fe_to_canonical(fe_from_u64limbs(a), fi->P), out);
    return ZK_OK;
}
extern "C" int32_t zk_fe_from_be_bytes_mod_order(int32_t field, const uint8_t *bytes, size_t len, uint64_t out[4]) {
    const FieldInfo *fi = field_info(field);
    if (!fi) return ZK_ERR_BAD_FIELD;
    fe_to_u64limbs(fe_from_be_bytes_mod_order(bytes, len, fi->P), out);
    return ZK_OK;
}

// ------------------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------------------
extern "C" int32_t zk_ctx_create(int32_t field, int32_t device, zk_ctx **out) {
    if (!out) return ZK_ERR_BAD_ARG;
    const FieldInfo *fi = field_info(field);
    if (!fi) return ZK_ERR_BAD_FIELD;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return ZK_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return ZK_ERR_NO_DEVICE;
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos) return ZK_ERR_NO_DEVICE;   // gfx950 code object only
    zk_ctx *c = new (std::nothrow) zk_ctx();
    if (!c) return ZK_ERR_ALLOC;
    c->field = field;
    c->device = device;
    c->fi = fi;
    c->own_stream = nullptr;
    c->d_partials = c->d_sums = c->h_pinned = nullptr;
    c->h_flag = nullptr;
    c->flag_seq = 0;
    c->h_results = nullptr;
    c->h_results_bytes = 0;
    c->pool_bytes = c->pool_checked = 0;
    c->inv2 = fe_inverse(fe_from_u32(2, fi->P), fi->P);
    c->pipe_consts.inv2p = mul29_prepare(c->inv2, fi->P);
    {
        Fe v = {{1, 0, 0, 0, 0, 0, 0, 0}};   // the integer 1; doubled 266 times modulo p (fe_add works on any residues < p)
        for (int i = 0; i < 266; ++i) v = fe_add(v, v, fi->P);
        split29(v.v, c->pipe_consts.k266.l);
    }
    c->d_dbg = nullptr;
    c->dbg_launch = 0;
    c->h_absorb[0] = c->h_absorb[1] = nullptr;
    c->h_absorb_bytes = 0;
    c->ev_absorb[0] = c->ev_absorb[1] = nullptr;
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    HIPCHK(hipMalloc(&c->d_partials, (size_t)kMaxGrid * kMaxSums * 32));
    HIPCHK(hipMalloc(&c->d_sums, (size_t)kMaxSums * 32 * 3));
    // completion word + result staging are POLLED by the host while the kernel that writes them is still running: ask for
    // coherent (fine-grained) mapped memory explicitly instead of relying on HIP_HOST_COHERENT's default
    HIPCHK(hipHostMalloc(&c->h_pinned, (size_t)kMaxSums * 32 * 3, kPolledHostFlags));
    HIPCHK(hipHostMalloc((void **)&c->h_flag, 64 * kMaxBatch, kPolledHostFlags));   // one 64-byte line per proof of a batch (slot 0: every other call)
    memset(c->h_flag, 0, 64 * kMaxBatch);
    HIPCHK(hipEventCreate(&c->ev0));
    HIPCHK(hipEventCreate(&c->ev1));
    *out = c;
    return ZK_OK;
}
extern "C" int32_t zk_ctx_destroy(zk_ctx *c) {
    if (!c) return ZK_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto &kv : c->twiddles) (void)hipFree(kv.second);
    for (auto &kv : c->lagrange_w) (void)hipFree(kv.second);
    for (auto &kv : c->ntt_plans) {
        (void)hipFree((void *)kv.second.w_lo);
        (void)hipFree((void *)kv.second.w_hi);
        for (int p = 0; p < 4; ++p)
            if (kv.second.w_full[p]) (void)hipFree((void *)kv.second.w_full[p]);
    }
    for (auto &kv : c->pool)
        for (void *q : kv.second) (void)hipFree(q);
    (void)hipFree(c->d_partials);
    (void)hipFree(c->d_sums);
    (void)hipHostFree(c->h_pinned);
    if (c->h_flag) (void)hipHostFree(c->h_flag);
    if (c->h_results) (void)hipHostFree(c->h_results);
    for (int b = 0; b < 2; ++b) {
        if (c->h_absorb[b]) (void)hipHostFree(c->h_absorb[b]);
        if (c->ev_absorb[b]) (void)hipEventDestroy(c->ev_absorb[b]);
    }
    (void)hipEventDestroy(c->ev0);
    (void)hipEventDestroy(c->ev1);
    (void)hipStreamDestroy(c->own_stream);
    delete c;
    return ZK_OK;
}
extern "C" int32_t zk_ctx_synchronize(zk_ctx *c) {
    if (!c) return ZK_ERR_BAD_ARG;
    ZKCHK(use_device(c));
    HIPCHK(hipStreamSynchronize(c->stream));
    return ZK_OK;
}
extern "C" int32_t zk_ctx_set_stream(zk_ctx *c, void *s) {
    if (!c) return ZK_ERR_BAD_ARG;
    ZKCHK(use_device(c));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->stream = (hipStream_t)s;   // NULL is the legacy default stream (what torch uses unless told otherwise)
    return ZK_OK;
}
extern "C" int32_t zk_ctx_use_own_stream(zk_ctx *c) {
    if (!c) return ZK_ERR_BAD_ARG;
    ZKCHK(use_device(c));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->stream = c->own_stream;
    return ZK_OK;
}
extern "C" int32_t zk_ctx_trim(zk_ctx *c) {
    if (!c) return ZK_ERR_BAD_ARG;
    ZKCHK(use_device(c));
    HIPCHK(hipStreamSynchronize(c->stream));
    pool_trim(c);
    c->pool_checked = 0;
    return ZK_OK;
}
extern "C" int32_t zk_ctx_field(const zk_ctx *c, int32_t *out) {
    if (!c || !out) return ZK_ERR_BAD_ARG;
    *out = c->field;
    return ZK_OK;
}

// ------------------------------------------------------------------------------------------------------------
// MultiLinearPolynomial
// ------------------------------------------------------------------------------------------------------------
static int32_t mle_alloc(zk_ctx *c, uint64_t n_vars, zk_mle **out) {
    if (n_vars > kMaxVars) return ZK_ERR_UNSUPPORTED;
    zk_mle *t = new (std::nothrow) zk_mle();
    if (!t) return ZK_ERR_ALLOC;
    t->ctx = c;
    t->n_vars = n_vars;
    t->d = nullptr;
    int32_t rc = pool_alloc(c, (size_t)32 << n_vars, (void **)&t->d);
    if (rc != ZK_OK) {
        delete t;
        return rc;
    }
    *out = t;
    return ZK_OK;
}
static void mle_release(zk_mle *t) {
    if (!t) return;
    pool_free(t->ctx, t->d, (size_t)32 << t->n_vars);
    delete t;
}
extern "C" int32_t zk_mle_alloc(zk_ctx *c, uint64_t n_vars, zk_mle **out) {
    if (!c || !out) return ZK_ERR_BAD_ARG;
    ZKCHK(use_device(c));
    return mle_alloc(c, n_vars, out);
}
extern "C" int32_t zk_mle_upload(zk_ctx *c, uint64_t n_vars, const uint64_t *evals, uint64_t len, zk_mle **out) {
    if (!c || !out || (!evals && len)) return ZK_ERR_BAD_ARG;
    if (n_vars >= 64 || len != (1ull << n_vars)) return ZK_ERR_EVAL_LEN;   // evaluation_form.rs:19-21
    ZKCHK(use_device(c));
    zk_mle *t = nullptr;
    ZKCHK(mle_alloc(c, n_vars, &t));
    hipError_t e = hipMemcpyAsync(t->d, evals, (size_t)len * 32, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        g_hip_err = std::string("upload: ") + hipGetErrorString(e);
        mle_release(t);
        return ZK_ERR_HIP;
    }
    *out = t;
    return ZK_OK;
}
extern "C" int32_t zk_mle_fill_random(zk_ctx *c, zk_mle *t, uint64_t seed, uint64_t first) {
    if (!c || !t) return ZK_ERR_BAD_ARG;
    if (t->ctx != c) return ZK_ERR_CONTEXT_MISMATCH;
    ZKCHK(use_device(c));
    const uint64_t n = 1ull << t->n_vars;
    k_fill_random<<<grid_for(n), kBlock, 0, c->stream>>>(t->d, n, seed, first, c->fi->P);
    HIPCHK(hipGetLastError());
    return ZK_OK;
}
extern "C" int32_t zk_mle_clone(zk_ctx *c, const zk_mle *t, zk_mle **out) {
    if (!c || !t || !out) return ZK_ERR_BAD_ARG;
    if (t->ctx != c) return ZK_ERR_CONTEXT_MISMATCH;
    ZKCHK(use_device(c));
    zk_mle *o = nullptr;
    ZKCHK(mle_alloc(c, t->n_vars, &o));
    HIPCHK(hipMemcpyAsync(o->d, t->d, (size_t)32 << t->n_vars, hipMemcpyDeviceToDevice, c->stream));
    *out = o;
    return ZK_OK;
}
extern "C" int32_t zk_mle_free(zk_ctx *c, zk_mle *t) {
    if (!t) return ZK_OK;
    if (!c || t->ctx != c) return ZK_ERR_CONTEXT_MISMATCH;
    mle_release(t);   // back to the context's pool; reuse is stream-ordered
    return ZK_OK;
}
extern "C" int32_t zk_mle_n_vars(const zk_mle *t, uint64_t *out) {
    if (!t || !out) return ZK_ERR_BAD_ARG;
    *out = t->n_vars;
    return ZK_OK;
}
extern "C" int32_t zk_mle_download(zk_ctx *c, const zk_mle *t, uint64_t *out) {
    if (!c || !t || !out) return ZK_ERR_BAD_ARG;
    if (t->ctx != c) return ZK_ERR_CONTEXT_MISMATCH;
    ZKCHK(use_device(c));
    HIPCHK(hipMemcpyAsync(out, t->d, (size_t)32 << t->n_vars, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return ZK_OK;
}
// #[derive(PartialEq)] (evaluation_form.rs:4): same n_vars and the same evaluations.  Elements are canonical (< p), so
// equality of representatives is equality in F_p.
extern "C" int32_t zk_mle_equal(zk_ctx *c, const zk_mle *a, const zk_mle *b, int32_t *out) {
    if (!c || !a || !b || !out) return ZK_ERR_BAD_ARG;
    if (a->ctx != c || b->ctx != c) return ZK_ERR_CONTEXT_MISMATCH;
    if (a->n_vars != b->n_vars) {
        *out = 0;
        return ZK_OK;
    }
    if (a->d == b->d) {
        *out = 1;
        return ZK_OK;
    }
    ZKCHK(use_device(c));
    uint32_t *d_flag = reinterpret_cast<uint32_t *>(c->d_sums);
    HIPCHK(hipMemsetAsync(d_flag, 0, 4, c->stream));
    const uint64_t n16 = (uint64_t)2 << a->n_vars;
    k_compare<<<grid_for(n16), kBlock, 0, c->stream>>>(reinterpret_cast<const uint4 *>(a->d), reinterpret_cast<const uint4 *>(b->d), n16, d_flag);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(c->h_pinned, d_flag, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    *out = *reinterpret_cast<uint32_t *>(c->h_pinned) == 0 ? 1 : 0;
    return ZK_OK;
}
extern "C" int32_t zk_mle_device_ptr(const zk_mle *t, void **out) {
    if (!t || !out) return ZK_ERR_BAD_ARG;
    *out = t->d;
    return ZK_OK;
}

// one assignment of partial_evaluate (evaluation_form.rs:54-72) as a kernel launch
static int32_t launch_fold(zk_ctx *c, const uint64_t *in, uint64_t *out, uint64_t m, uint64_t initial_var, const Fe &r) {
    const uint64_t pairs = 1ull << (m - 1);
    const uint32_t pos = (uint32_t)(m - 1 - initial_var);
    if (initial_var == 0 && pairs >= 64) {   // the sumcheck fold: lane-pair coalesced, nontemporal streaming kernel
        uint64_t g = pairs / kBlock;           // one 64-element run per wave
        if (g > 2 * kMaxGridStream) g = 2 * kMaxGridStream;
        k_fold_msb<<<(uint32_t)(g ? g : 1), kBlock, 0, c->stream>>>(in, out, pairs, c->fi->P, mul29_prepare(r, c->fi->P));
        HIPCHK(hipGetLastError());
        return ZK_OK;
    }
    if (pos >= 6 && in != out) {   // any other variable whose pair partner is >= 64 elements away: the same run access, block by block
        uint64_t g = pairs / kBlock;
        if (g > 2 * kMaxGridStream) g = 2 * kMaxGridStream;
        k_fold_run<<<(uint32_t)(g ? g : 1), kBlock, 0, c->stream>>>(in, out, pairs, pos, c->fi->P, mul29_prepare(r, c->fi->P));
        HIPCHK(hipGetLastError());
        return ZK_OK;
    }
    if (pos < 6 && pairs >= 64 && in != out) {   // partner inside the wave's 128-element run: one in-wave exchange (k_fold_low)
        uint64_t g = pairs / kBlock;
        if (g > 2 * kMaxGridStream) g = 2 * kMaxGridStream;
        k_fold_low<<<(uint32_t)(g ? g : 1), kBlock, 0, c->stream>>>(in, out, pairs, pos, c->fi->P, mul29_prepare(r, c->fi->P));
        HIPCHK(hipGetLastError());
        return ZK_OK;
    }
    uint64_t g = (pairs + kBlock - 1) / kBlock;
    if (g > kMaxGridStream) g = kMaxGridStream;
    k_fold<<<(uint32_t)(g ? g : 1), kBlock, 0, c->stream>>>(in, out, pairs, pos, c->fi->P, mul29_prepare(r, c->fi->P));
    HIPCHK(hipGetLastError());
    return ZK_OK;
}
// the reference's panics (u8 / usize underflow at evaluation_form.rs:55,75 and pairing_index.rs:3,6) as a status
static int32_t check_partial_args(uint64_t n_vars, uint64_t initial_var, uint64_t n_assign) {
    if (n_assign > n_vars) return ZK_ERR_PANIC_INDEX;
    for (uint64_t i = 0; i < n_assign; ++i) {
        const uint64_t nv = n_vars - i;
        if (nv == 0 || initial_var > nv - 1) return ZK_ERR_PANIC_INDEX;
    }
    return ZK_OK;
}

extern "C" int32_t zk_mle_partial_evaluate(zk_ctx *c, const zk_mle *t, uint64_t initial_var, const uint64_t *assignments,
                                           uint64_t n_assign, zk_mle **out) {
    if (!c || !t || !out || (!assignments && n_assign)) return ZK_ERR_BAD_ARG;
    if (t->ctx != c) return ZK_ERR_CONTEXT_MISMATCH;
    ZKCHK(check_partial_args(t->n_vars, initial_var, n_assign));
    ZKCHK(use_device(c));
    if (n_assign == 0) return zk_mle_clone(c, t, out);
    zk_mle *res = nullptr, *tmp[2] = {nullptr, nullptr};
    ZKCHK(mle_alloc(c, t->n_vars - n_assign, &res));
    int32_t rc = ZK_OK;
    if (n_assign >= 2) rc = mle_alloc(c, t->n_vars - 1, &tmp[0]);
    if (rc == ZK_OK && n_assign >= 3) rc = mle_alloc(c, t->n_vars - 2, &tmp[1]);
    const uint64_t *src = t->d;
    for (uint64_t i = 0; i < n_assign && rc == ZK_OK; ++i) {
        uint64_t *dst = (i == n_assign - 1) ? res->d : tmp[i & 1]->d;
        rc = launch_fold(c, src, dst, t->n_vars - i, initial_var, fe_from_u64limbs(assignments + 4 * i));
        src = dst;
    }
    for (int i = 0; i < 2; ++i) mle_release(tmp[i]);
    if (rc != ZK_OK) {
        mle_release(res);
        return rc;
    }
    *out = res;
    return ZK_OK;
}

extern "C" int32_t zk_mle_fold_into(zk_ctx *c, const zk_mle *t, const uint64_t r[4], zk_mle *out) {
    if (!c || !t || !r || !out) return ZK_ERR_BAD_ARG;
    if (t->ctx != c || out->ctx != c) return ZK_ERR_CONTEXT_MISMATCH;
    if (t->n_vars == 0) return ZK_ERR_PANIC_INDEX;
    if (out->n_vars != t->n_vars - 1) return ZK_ERR_BAD_ARG;
    ZKCHK(use_device(c));
    return launch_fold(c, t->d, out->d, t->n_vars, 0, fe_from_u64limbs(r));
}

// evaluate (evaluation_form.rs:83-89): MSB folds -- the first out of place into scratch, the next ones in place there,
// and the last <= kEvalTailVars variables in one single-workgroup launch (k_evaluate_tail)
// flag_seq != 0: d_out_elem is pinned host memory and the kernel that writes it also stores flag_seq into c->h_flag (host_flag_wait)
static int32_t evaluate_device(zk_ctx *c, const zk_mle *t, const uint64_t *point, uint64_t *d_out_elem, uint32_t flag_seq = 0) {
    const uint64_t n = t->n_vars;
    if (n == 0) {
        HIPCHK(hipMemcpyAsync(d_out_elem, t->d, 32, hipMemcpyDeviceToDevice, c->stream));
        return ZK_OK;
    }
    const FieldParams &P = c->fi->P;
    // bulk: the LOW variables first, 8-12 per launch (k_eval_low, one workgroup per output element).  With 8..12 variables left one
    // workgroup finishes and writes the result; with more, a launch leaves 8 for that last one when 8..12 low variables get
    // there, otherwise it takes 12; fewer than 8 left over go to k_evaluate_tail.
    uint64_t *scratch[2] = {nullptr, nullptr};
    size_t scratch_bytes[2] = {0, 0};
    int32_t rc = ZK_OK;
    const uint64_t *src = t->d;
    uint64_t cur = n;   // variables left
    static const bool bulk_low = !env_flag("ZK_EVAL_FOLDS");   // ZK_EVAL_FOLDS=1: the variable-by-variable path (A/B, tests)
    // tables of at least this many variables take the streaming kernel (k_eval_stream: up to 15 variables per launch, half an
    // element per lane, carry-free column sums); smaller ones are launch latency and keep k_eval_low.  ZK_EVAL_STREAM_MIN overrides.
    static const uint64_t stream_min = env_u64("ZK_EVAL_STREAM_MIN", 21, 0, 1000);   // above kMaxVars: never
    for (int pass = 0; bulk_low && cur >= 8 && rc == ZK_OK; ++pass) {
        // the streaming launch leaves 9 variables (512 workgroups), 8 at 21 variables: every workgroup spends ~2600 instructions per
        // wave on its weight tables before the first product, so at 2^21 elements half as many workgroups with twice the rows win
        // (device time 31.2 -> 29.1 us; 10 left: 54 us -- profiles/r04_evaluate_stream_grid_ab.log).  ZK_EVAL_STREAM_LEAVE overrides.
        static const uint64_t leave_env = env_u64("ZK_EVAL_STREAM_LEAVE", 0, 7, kEvalHighMax);   // 0 = not set
        const uint64_t stream_leave = leave_env ? leave_env : (cur <= 21 ? (uint64_t)8 : (uint64_t)9);
        const bool stream = cur >= stream_min && cur >= (uint64_t)kEvalStreamMin + stream_leave;
        uint64_t L = cur <= (uint64_t)kEvalLowMax ? cur : cur - 8;
        if (L > (uint64_t)kEvalLowMax) L = kEvalLowMax;
        if (L < (uint64_t)kEvalLowMin) L = kEvalLowMin;   // cur >= 13 here
        if (stream) L = cur - stream_leave < (uint64_t)kEvalStreamMax ? cur - stream_leave : (uint64_t)kEvalStreamMax;
        const uint64_t n_out = 1ull << (cur - L);
        uint64_t *dst = d_out_elem;
        if (L != cur) {
            const int b = pass & 1;
            if (scratch_bytes[b] < n_out * 32) {
                if (scratch[b]) pool_free(c, scratch[b], scratch_bytes[b]);
                scratch[b] = nullptr;
                scratch_bytes[b] = (size_t)n_out * 32;
                rc = pool_alloc(c, scratch_bytes[b], (void **)&scratch[b]);
                if (rc != ZK_OK) {
                    scratch_bytes[b] = 0;
                    break;
                }
            }
            dst = scratch[b];
        }
        // when at most kEvalHighMax variables remain after this launch its workgroups weight their outputs with eq(point_high, g)
        // (common.cuh eval_high_weight) and what is left is a plain sum of the 2^H outputs (k_eval_sum) instead of another bulk launch
        const uint64_t H = cur - L;
        EvalHighPoint ph = {};
        static const int weight_mode = (int)env_u64("ZK_EVAL_WEIGHT", 3, 0, 3);   // A/B: bit 0 = k_eval_low, bit 1 = k_eval_stream
        if (H >= 1 && H <= (uint64_t)kEvalHighMax && (weight_mode & (stream ? 2 : 1))) {
            ph.n = (uint32_t)H;
            for (uint64_t p = 0; p < H; ++p) {
                const Fe r = fe_from_u64limbs(point + 4 * (H - 1 - p));   // bit p of the output index <-> variable H-1-p of the point
                for (int i = 0; i < 8; ++i) ph.r[p][i] = r.v[i];
            }
        }
        if (stream) {
            EvalStreamPoint sp = {};
            for (uint64_t p = 0; p < L; ++p) {
                const Fe r = fe_from_u64limbs(point + 4 * (cur - 1 - p));   // index bit p <-> variable cur-1-p (variable 0 is the MSB)
                for (int i = 0; i < 8; ++i) sp.r[p][i] = r.v[i];
            }
            const Fe two128 = {{0, 0, 0, 0, 1, 0, 0, 0}};
            const Fe c128 = fe_from_canonical(two128, P);
            for (int i = 0; i < 8; ++i) sp.c128[i] = c128.v[i];
            k_eval_stream<<<(uint32_t)n_out, kEvalStreamThreads, 0, c->stream>>>(src, dst, (uint32_t)L, sp, P, ph);
        } else {
            EvalLowPoint pt = {};
            for (uint64_t p = 0; p < L; ++p) {
                const Fe r = fe_from_u64limbs(point + 4 * (cur - 1 - p));   // index bit p <-> variable cur-1-p (variable 0 is the MSB)
                for (int i = 0; i < 8; ++i) pt.r[p][i] = r.v[i];
            }
            k_eval_low<<<(uint32_t)n_out, kBlock, 0, c->stream>>>(src, dst, (uint32_t)L, pt, P, (flag_seq && L == cur) ? c->h_flag : nullptr, flag_seq, ph);
        }
        if (hipGetLastError() != hipSuccess) rc = ZK_ERR_HIP;
        src = dst;
        cur -= L;
        if (ph.n && rc == ZK_OK) {   // the outputs carry their weights: the result is their sum
            k_eval_sum<<<1, kBlock, 0, c->stream>>>(src, (uint32_t)n_out, P, d_out_elem, flag_seq ? c->h_flag : nullptr, flag_seq);
            if (hipGetLastError() != hipSuccess) rc = ZK_ERR_HIP;
            cur = 0;
            break;
        }
    }
    auto release = [&]() {
        for (int b = 0; b < 2; ++b)
            if (scratch[b]) pool_free(c, scratch[b], scratch_bytes[b]);
    };
    if (rc != ZK_OK || (bulk_low && cur == 0)) {   // failed, or the last k_eval_low has written the result
        release();
        return rc;
    }
    const uint64_t tail_vars = cur < (uint64_t)kEvalTailVars ? cur : (uint64_t)kEvalTailVars;
    const uint64_t big = cur - tail_vars;                    // folds done as full launches (ZK_EVAL_FOLDS only)
    uint64_t *fold_scratch = nullptr;
    size_t fold_scratch_bytes = 0;
    if (big) {
        fold_scratch_bytes = (size_t)32 << (cur - 1);
        rc = pool_alloc(c, fold_scratch_bytes, (void **)&fold_scratch);
        if (rc != ZK_OK) {
            release();
            return rc;
        }
    }
    for (uint64_t i = 0; i < big && rc == ZK_OK;) {
        const uint64_t left = big - i, m = cur - i;
        if (left >= 2) {   // three (or two) variables per launch
            const int v = left >= 3 ? 3 : 2;
            FoldChallenges ch = {};
            for (int u = 0; u < v; ++u) ch.r[u] = mul29_prepare(fe_from_u64limbs(point + 4 * (i + u)), P);
            const uint64_t n_out = 1ull << (m - v);
            if (v == 3) k_fold_multi<3><<<grid_for(n_out), kBlock, 0, c->stream>>>(src, fold_scratch, n_out, P, ch);
            else k_fold_multi<2><<<grid_for(n_out), kBlock, 0, c->stream>>>(src, fold_scratch, n_out, P, ch);
            if (hipGetLastError() != hipSuccess) rc = ZK_ERR_HIP;
            i += v;
        } else {
            rc = launch_fold(c, src, fold_scratch, m, 0, fe_from_u64limbs(point + 4 * (i)));
            i += 1;
        }
        src = fold_scratch;
    }
    if (rc == ZK_OK) {
        // the remaining assignments travel in the kernel arguments: no staging buffer, no synchronisation
        EvalTailChallenges chs = {};
        for (uint64_t v = 0; v < tail_vars; ++v) {
            const Mul29 r = mul29_prepare(fe_from_u64limbs(point + 4 * (big + v)), P);
            for (int i = 0; i < 9; ++i) chs.w[v][i] = r.l[i];
        }
        const size_t lds = (size_t)32 << (tail_vars - 1);
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_evaluate_tail), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
            rc = ZK_ERR_HIP;
        if (rc == ZK_OK) {
            k_evaluate_tail<<<1, kEvalTailThreads, lds, c->stream>>>(src, (uint32_t)tail_vars, chs, P, d_out_elem, flag_seq ? c->h_flag : nullptr,
                                                                     flag_seq);
            if (hipGetLastError() != hipSuccess) rc = ZK_ERR_HIP;
        }
    }
    release();
    if (fold_scratch) pool_free(c, fold_scratch, fold_scratch_bytes);
    return rc;
}
extern "C" int32_t zk_mle_evaluate(zk_ctx *c, const zk_mle *t, const uint64_t *point, uint64_t n_point, uint64_t out[4]) {
    if (!c || !t || !out || (!point && n_point)) return ZK_ERR_BAD_ARG;
    if (t->ctx != c) return ZK_ERR_CONTEXT_MISMATCH;
    if (n_point != t->n_vars) return ZK_ERR_EVAL_ARITY;   // evaluation_form.rs:84-86
    ZKCHK(use_device(c));
    static const bool host_dbg = env_flag("ZK_HOST_DEBUG");
    const auto t_enter = std::chrono::steady_clock::now();
    if (t->n_vars == 0) {
        ZKCHK(evaluate_device(c, t, point, c->d_sums));
        HIPCHK(hipMemcpyAsync(c->h_pinned, c->d_sums, 32, hipMemcpyDeviceToHost, c->stream));
    } else {
        // the last kernel stores the 32-byte result straight into the pinned host buffer (mapped into the device's address space):
        // a device-to-host copy of 32 bytes is a blit launch of its own (4-5 us on the stream)
        const uint32_t seq = next_flag_seq(c);
        ZKCHK(evaluate_device(c, t, point, c->h_pinned, seq));
        const auto t_enq0 = std::chrono::steady_clock::now();
        ZKCHK(host_flag_wait(c, seq));   // the completion word the last kernel stores next to the result
        if (host_dbg) {
            const auto t_done = std::chrono::steady_clock::now();
            fprintf(stderr, "[host] evaluate n=%llu: enqueue %.1f us, wait %.1f us\n", (unsigned long long)t->n_vars,
                    std::chrono::duration<double, std::micro>(t_enq0 - t_enter).count(),
                    std::chrono::duration<double, std::micro>(t_done - t_enq0).count());
        }
        memcpy(out, c->h_pinned, 32);
        return ZK_OK;
    }
    const auto t_enq = std::chrono::steady_clock::now();
    HIPCHK(stream_wait(c->stream));
    if (host_dbg) {
        const auto t_done = std::chrono::steady_clock::now();
        fprintf(stderr, "[host] evaluate n=%llu: enqueue %.1f us, wait %.1f us\n", (unsigned long long)t->n_vars,
                std::chrono::duration<double, std::micro>(t_enq - t_enter).count(),
                std::chrono::duration<double, std::micro>(t_done - t_enq).count());
    }
    memcpy(out, c->h_pinned, 32);
    return ZK_OK;
}

static int32_t absorb_tables(zk_ctx *c, Sponge &sp, zk_mle *const *f, uint64_t k);
template <class Consume>
static int32_t stream_table_bytes(zk_ctx *c, const zk_mle *const *f, uint64_t k, Consume &&consume);
// chunk -> caller's buffer on a few host threads: a fresh destination (a new Vec<u8>) is page-fault bound, and faults parallelise.
// The helpers live for ONE zk_mle_to_bytes call (started once, handed every chunk, joined at its end), never more than three of
// them; their number follows the CPUs this process may run on (sched_getaffinity, so cgroup / taskset limits count), and
// ZK_TO_BYTES_THREADS (1..4; 1 = the caller's thread only) overrides it.
class CopyHelpers {
  public:
    explicit CopyHelpers(size_t total_bytes) {
        static const unsigned from_env = (unsigned)env_u64("ZK_TO_BYTES_THREADS", 0, 1, 4);   // 0: not set
        unsigned nt = from_env;
        if (!nt) {
            cpu_set_t set;
            CPU_ZERO(&set);
            nt = sched_getaffinity(0, sizeof set, &set) == 0 ? (unsigned)CPU_COUNT(&set) : 1u;
            if (nt > 4) nt = 4;
        }
        if (nt < 2 || total_bytes < 2 * kMinPerThread) return;
        for (unsigned i = 1; i < nt; ++i) {
            try {
                th_.emplace_back([this, i] { work(i); });
            } catch (...) {
                break;   // no thread to be had: the ones that started (perhaps none) share the work
            }
        }
    }
    ~CopyHelpers() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    CopyHelpers(const CopyHelpers &) = delete;
    CopyHelpers &operator=(const CopyHelpers &) = delete;
    void copy(uint8_t *dst, const uint8_t *src, size_t bytes) {
        const unsigned parts = (unsigned)th_.size() + 1;
        if (parts < 2 || bytes < 2 * kMinPerThread) {
            memcpy(dst, src, bytes);
            return;
        }
        const size_t per = (bytes / parts + 4095) & ~(size_t)4095;
        {
            std::lock_guard<std::mutex> lk(mu_);
            dst_ = dst, src_ = src, bytes_ = bytes, per_ = per;
            pending_ = parts - 1;
            ++generation_;
        }
        cv_.notify_all();
        memcpy(dst, src, per < bytes ? per : bytes);
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [this] { return pending_ == 0; });
    }

  private:
    static constexpr size_t kMinPerThread = (size_t)2 << 20;
    void work(unsigned part) {
        uint64_t seen = 0;
        for (;;) {
            uint8_t *dst;
            const uint8_t *src;
            size_t bytes, per;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || generation_ != seen; });
                if (stop_) return;
                seen = generation_;
                dst = dst_, src = src_, bytes = bytes_, per = per_;
            }
            const size_t off = per * part;
            if (off < bytes) memcpy(dst + off, src + off, bytes - off < per ? bytes - off : per);
            std::lock_guard<std::mutex> lk(mu_);
            if (--pending_ == 0) done_.notify_one();
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    uint8_t *dst_ = nullptr;
    const uint8_t *src_ = nullptr;
    size_t bytes_ = 0, per_ = 0;
    unsigned pending_ = 0;
    uint64_t generation_ = 0;
    bool stop_ = false;
};
extern "C" int32_t zk_mle_to_bytes(zk_ctx *c, const zk_mle *t, uint8_t *out) {
    if (!c || !t || !out) return ZK_ERR_BAD_ARG;
    if (t->ctx != c) return ZK_ERR_CONTEXT_MISMATCH;
    ZKCHK(use_device(c));
    // device serialiser -> pinned staging (two buffers, the copy of chunk i + 1 under the host's copy-out of chunk i) -> caller
    size_t off = 0;
    const zk_mle *one[1] = {t};
    CopyHelpers helpers((size_t)32 << t->n_vars);
    return stream_table_bytes(c, one, 1, [&](const uint8_t *p, size_t bytes) {
        helpers.copy(out + off, p, bytes);
        off += bytes;
    });
}

extern "C" int32_t zk_mle_partial_evaluate_host(zk_ctx *c, uint64_t n_vars, const uint64_t *evals, uint64_t len,
                                                uint64_t initial_var, const uint64_t *assignments, uint64_t n_assign,
                                                uint64_t *out_evals) {
    if (!out_evals) return ZK_ERR_BAD_ARG;
    zk_mle *t = nullptr, *o = nullptr;
    ZKCHK(zk_mle_upload(c, n_vars, evals, len, &t));
    int32_t rc = zk_mle_partial_evaluate(c, t, initial_var, assignments, n_assign, &o);
    if (rc == ZK_OK) rc = zk_mle_download(c, o, out_evals);
    (void)zk_mle_free(c, t);
    (void)zk_mle_free(c, o);
    return rc;
}

static int32_t results_staging(zk_ctx *c, size_t bytes, uint8_t **out);
static uint64_t bit_reverse64(uint64_t x) {
    x = ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0f0f0f0f0f0f0f0full) | ((x & 0x0f0f0f0f0f0f0f0full) << 4);
    return __builtin_bswap64(x);
}
// the zeta tile kernels use 64 KiB + 128 B of dynamic LDS (two per CU): above the 64-KiB default, so opt in once per device
static int32_t zeta_lds_opt_in(zk_ctx *c) {
    static std::mutex mu;
    static std::set<int> done;
    std::lock_guard<std::mutex> lk(mu);
    if (done.count(c->device)) return ZK_OK;
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_zeta_first), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kZetaLdsBytes));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_zeta_tile), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kZetaLdsBytes));
    done.insert(c->device);
    return ZK_OK;
}
// CoeffMultilinearPolynomial::to_evaluation_form (coefficient_form.rs:340-347): scatter + zeta transform on the device
static int32_t coeff_to_evaluation_impl(zk_ctx *c, uint64_t n_vars, const uint64_t *keys, const uint64_t *coeffs, uint64_t n_terms,
                                        zk_mle **out);
// the term list is merged in host containers sized by the caller's n_terms: an allocation failure is a status, not an exception
extern "C" int32_t zk_coeff_to_evaluation(zk_ctx *c, uint64_t n_vars, const uint64_t *keys, const uint64_t *coeffs, uint64_t n_terms,
                                          zk_mle **out) {
    try {
        return coeff_to_evaluation_impl(c, n_vars, keys, coeffs, n_terms, out);
    } catch (const std::bad_alloc &) {
        return ZK_ERR_ALLOC;
    } catch (...) {
        return ZK_ERR_BAD_ARG;
    }
}
// lists longer than this are ordered on the device (zeta_sort.hip): the host's sort + merge + upload of 2^16 terms took 3.5 ms, six
// times the transform; ZK_ZETA_DEVICE_SORT_MIN overrides (0 = always on the device, tests)
static uint64_t zeta_device_sort_min() {
    static const uint64_t v = env_u64("ZK_ZETA_DEVICE_SORT_MIN", 4096, 0, (uint64_t)1 << 40);
    return v;
}
namespace zk {
int zeta_sort_terms(hipStream_t stream, const uint64_t *d_keys, uint64_t n, uint32_t n_vars, uint64_t *d_idx_unsorted, uint32_t *d_pos_unsorted,
                    uint64_t *d_idx_sorted, uint32_t *d_perm, void *temp, size_t *temp_bytes);
}
// the tiled passes over a table whose first pass has been given its term list
static int32_t zeta_tiled_passes(zk_ctx *c, zk_mle *t, uint64_t n_vars, const uint64_t *d_keys, const uint64_t *d_coeffs, uint64_t m, const uint32_t *d_perm) {
    ZKCHK(zeta_lds_opt_in(c));
    const uint32_t tile_log = n_vars < kZetaTileLog ? (uint32_t)n_vars : kZetaTileLog;
    k_zeta_first<<<(uint32_t)(1ull << (n_vars - tile_log)), kBlock, kZetaLdsBytes, c->stream>>>(t->d, d_keys, d_coeffs, m, tile_log, c->fi->P, d_perm);
    HIPCHK(hipGetLastError());
    const uint32_t rem = (uint32_t)n_vars - tile_log, n_pass = (rem + 7) / 8;
    uint32_t pos = tile_log;
    for (uint32_t p = 0; p < n_pass; ++p) {
        const uint32_t L = rem / n_pass + (p < rem % n_pass ? 1u : 0u);
        k_zeta_tile<<<(uint32_t)(1ull << (n_vars - kZetaTileLog)), kBlock, kZetaLdsBytes, c->stream>>>(t->d, pos, L, c->fi->P);
        HIPCHK(hipGetLastError());
        pos += L;
    }
    return ZK_OK;
}
// long term lists: upload as given, order on the device, sum duplicate keys inside the first pass
static int32_t coeff_to_evaluation_device_sort(zk_ctx *c, uint64_t n_vars, const uint64_t *keys, const uint64_t *coeffs, uint64_t n_terms, zk_mle **out) {
    zk_mle *t = nullptr;
    ZKCHK(mle_alloc(c, n_vars, &t));
    // one device block: [keys | idx unsorted | idx sorted | coeffs | pos unsorted | perm | sort scratch]
    size_t temp_bytes = 0;
    if (zeta_sort_terms(c->stream, nullptr, n_terms, (uint32_t)n_vars, nullptr, nullptr, nullptr, nullptr, nullptr, &temp_bytes) != 0) {
        mle_release(t);
        return ZK_ERR_HIP;
    }
    const size_t w8 = (size_t)n_terms * 8, w4 = ((size_t)n_terms * 4 + 7) & ~(size_t)7;
    const size_t total = 3 * w8 + 4 * w8 + 2 * w4 + ((temp_bytes + 255) & ~(size_t)255) + 256;
    uint8_t *blk = nullptr;
    int32_t rc = pool_alloc(c, total, (void **)&blk);
    if (rc == ZK_OK) {
        uint64_t *d_keys = reinterpret_cast<uint64_t *>(blk), *d_idx_u = d_keys + n_terms, *d_idx_s = d_idx_u + n_terms, *d_coeffs = d_idx_s + n_terms;
        uint32_t *d_pos_u = reinterpret_cast<uint32_t *>(blk + 7 * w8), *d_perm = reinterpret_cast<uint32_t *>(blk + 7 * w8 + w4);
        void *temp = blk + 7 * w8 + 2 * w4;
        if (hipMemcpyAsync(d_keys, keys, w8, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
            hipMemcpyAsync(d_coeffs, coeffs, 4 * w8, hipMemcpyHostToDevice, c->stream) != hipSuccess)
            rc = ZK_ERR_HIP;
        if (rc == ZK_OK && zeta_sort_terms(c->stream, d_keys, n_terms, (uint32_t)n_vars, d_idx_u, d_pos_u, d_idx_s, d_perm, temp, &temp_bytes) != 0) rc = ZK_ERR_HIP;
        if (rc == ZK_OK) rc = zeta_tiled_passes(c, t, n_vars, d_idx_s, d_coeffs, n_terms, d_perm);
    }
    if (hipStreamSynchronize(c->stream) != hipSuccess && rc == ZK_OK) rc = ZK_ERR_HIP;   // the caller's arrays were read by the copies
    if (blk) pool_free(c, blk, total);
    if (rc != ZK_OK) {
        mle_release(t);
        return rc;
    }
    *out = t;
    return ZK_OK;
}
static int32_t coeff_to_evaluation_impl(zk_ctx *c, uint64_t n_vars, const uint64_t *keys, const uint64_t *coeffs, uint64_t n_terms,
                                        zk_mle **out) {
    if (!c || !out || (n_terms && (!keys || !coeffs))) return ZK_ERR_BAD_ARG;
    if (n_vars == 0 || n_vars > kMaxVars) return ZK_ERR_EVAL_LEN;
    for (uint64_t t = 0; t < n_terms; ++t)
        if (keys[t] >> n_vars) return ZK_ERR_COEFF_RANGE;                            // coefficient_form.rs:183-186
    ZKCHK(use_device(c));
    static const bool global_passes_flag = env_flag("ZK_ZETA_GLOBAL");
    if (!global_passes_flag && n_terms && n_terms >= zeta_device_sort_min() && n_terms < ((uint64_t)1 << 32))
        return coeff_to_evaluation_device_sort(c, n_vars, keys, coeffs, n_terms, out);
    // BTreeMap semantics: one entry per key, duplicate terms summed (coefficient_form.rs:164-171).  Keyed by the TABLE INDEX of the
    // term (key bit v <-> variable v <-> index bit n-1-v: the bit-reversed key), so the list comes out sorted by index, which is
    // what k_zeta_first's per-tile binary search needs.
    std::vector<std::pair<uint64_t, uint64_t>> order(n_terms);   // (table index, position in the caller's list)
    for (uint64_t t = 0; t < n_terms; ++t) order[t] = {bit_reverse64(keys[t]) >> (64 - n_vars), t};
    std::sort(order.begin(), order.end());
    // one staging block: m indices, then m coefficients (4 words each); pinned (the context's results block) while it is small
    const size_t stage_bytes = (size_t)n_terms * 40;
    std::vector<uint64_t> pageable;
    uint64_t *hs = nullptr;
    if (stage_bytes <= ((size_t)1 << 20)) {
        uint8_t *blk = nullptr;
        ZKCHK(results_staging(c, stage_bytes ? stage_bytes : 8, &blk));
        hs = reinterpret_cast<uint64_t *>(blk);
    } else {
        pageable.resize((size_t)n_terms * 5);
        hs = pageable.data();
    }
    uint64_t m = 0;
    {
        uint64_t *hk = hs, *hc = hs + n_terms;   // indices packed at the front; coefficients start at word n_terms (>= m)
        for (uint64_t i = 0; i < n_terms;) {
            Fe acc = fe_from_u64limbs(coeffs + 4 * order[i].second);
            uint64_t j = i + 1;
            for (; j < n_terms && order[j].first == order[i].first; ++j)
                acc = fe_add(acc, fe_from_u64limbs(coeffs + 4 * order[j].second), c->fi->P);
            hk[m] = order[i].first;
            fe_to_u64limbs(acc, hc + 4 * m);
            ++m;
            i = j;
        }
    }
    zk_mle *t = nullptr;
    ZKCHK(mle_alloc(c, n_vars, &t));
    uint64_t *d_terms = nullptr, *d_keys = nullptr, *d_coeffs = nullptr;
    int32_t rc = ZK_OK;
    if (m) {
        rc = pool_alloc(c, (size_t)n_terms * 40, (void **)&d_terms);
        if (rc == ZK_OK && hipMemcpyAsync(d_terms, hs, (size_t)(n_terms + 4 * m) * 8, hipMemcpyHostToDevice, c->stream) != hipSuccess)
            rc = ZK_ERR_HIP;
        d_keys = d_terms;
        d_coeffs = d_terms + n_terms;
    }
    static const bool global_passes = env_flag("ZK_ZETA_GLOBAL");   // round 4's path (memset + scatter + three bits per launch): A/B only
    if (rc == ZK_OK && !global_passes) {
        // LDS-tiled passes (zeta_kernels.cuh): the low min(n, 11) index bits from the term list without reading the table, then
        // the remaining bits in passes of at most 8, evenly split
        rc = zeta_lds_opt_in(c);
        const uint32_t tile_log = n_vars < kZetaTileLog ? (uint32_t)n_vars : kZetaTileLog;
        if (rc == ZK_OK) {
            k_zeta_first<<<(uint32_t)(1ull << (n_vars - tile_log)), kBlock, kZetaLdsBytes, c->stream>>>(t->d, d_keys, d_coeffs, m, tile_log,
                                                                                                      c->fi->P);
            if (hipGetLastError() != hipSuccess) rc = ZK_ERR_HIP;
        }
        const uint32_t rem = (uint32_t)n_vars - tile_log, n_pass = (rem + 7) / 8;
        uint32_t pos = tile_log;
        for (uint32_t p = 0; p < n_pass && rc == ZK_OK; ++p) {
            const uint32_t L = rem / n_pass + (p < rem % n_pass ? 1u : 0u);
            k_zeta_tile<<<(uint32_t)(1ull << (n_vars - kZetaTileLog)), kBlock, kZetaLdsBytes, c->stream>>>(t->d, pos, L, c->fi->P);
            if (hipGetLastError() != hipSuccess) rc = ZK_ERR_HIP;
            pos += L;
        }
    } else if (rc == ZK_OK) {
        if (hipMemsetAsync(t->d, 0, (size_t)32 << n_vars, c->stream) != hipSuccess) rc = ZK_ERR_HIP;
        if (rc == ZK_OK && m) {
            k_scatter_terms<<<grid_for(m), kBlock, 0, c->stream>>>(d_keys, d_coeffs, m, t->d);
            if (hipGetLastError() != hipSuccess) rc = ZK_ERR_HIP;
        }
        for (uint32_t b = 0; b < n_vars && rc == ZK_OK;) {   // the subset-sum butterfly over every index bit, three bits per pass
            const uint32_t v = n_vars - b >= 3 ? 3u : (uint32_t)(n_vars - b);
            const uint64_t groups = 1ull << (n_vars - v);
            uint64_t g = (groups + kBlock - 1) / kBlock;
            if (g > kMaxGridStream) g = kMaxGridStream;
            if (v == 3) k_zeta_multi<3><<<(uint32_t)g, kBlock, 0, c->stream>>>(t->d, groups, b, c->fi->P);
            else if (v == 2) k_zeta_multi<2><<<(uint32_t)g, kBlock, 0, c->stream>>>(t->d, groups, b, c->fi->P);
            else k_zeta_pass<<<(uint32_t)g, kBlock, 0, c->stream>>>(t->d, groups, b, c->fi->P);
            if (hipGetLastError() != hipSuccess) rc = ZK_ERR_HIP;
            b += v;
        }
    }
    if (hipStreamSynchronize(c->stream) != hipSuccess && rc == ZK_OK) rc = ZK_ERR_HIP;   // host staging vectors go out of scope
    if (d_terms) pool_free(c, d_terms, (size_t)n_terms * 40);
    if (rc != ZK_OK) {
        mle_release(t);
        return rc;
    }
    *out = t;
    return ZK_OK;
}

// ------------------------------------------------------------------------------------------------------------
// ProductPoly
// ------------------------------------------------------------------------------------------------------------
extern "C" int32_t zk_product_check(const zk_mle *const *f, uint64_t k) {
    if (k == 0) return ZK_ERR_EMPTY_PRODUCT;   // product_poly.rs:15-17
    if (!f) return ZK_ERR_BAD_ARG;
    for (uint64_t i = 0; i < k; ++i) {
        if (!f[i]) return ZK_ERR_BAD_ARG;
        if (f[i]->n_vars != f[0]->n_vars) return ZK_ERR_ARITY_MISMATCH;   // product_poly.rs:20-26
        if (f[i]->ctx != f[0]->ctx) return ZK_ERR_CONTEXT_MISMATCH;
    }
    return ZK_OK;
}
static int32_t product_args(zk_ctx *c, const zk_mle *const *f, uint64_t k) {
    if (!c) return ZK_ERR_BAD_ARG;
    ZKCHK(zk_product_check(f, k));
    if (f[0]->ctx != c) return ZK_ERR_CONTEXT_MISMATCH;
    if (k > (uint64_t)kMaxFactors) return ZK_ERR_UNSUPPORTED;
    return use_device(c);
}

extern "C" int32_t zk_prod_reduce(zk_ctx *c, const zk_mle *const *f, uint64_t k, zk_mle **out) {
    if (!out) return ZK_ERR_BAD_ARG;
    ZKCHK(product_args(c, f, k));
    zk_mle *o = nullptr;
    ZKCHK(mle_alloc(c, f[0]->n_vars, &o));
    FactorPtrs fp = {};
    for (uint64_t i = 0; i < k; ++i) fp.in[i] = f[i]->d;
    const uint64_t n = 1ull << f[0]->n_vars;
    if (n >= 64) {
        uint64_t g = n / kBlock;   // one 64-element run per wave and pass
        if (g > 2 * kMaxGridStream) g = 2 * kMaxGridStream;
        k_prod_reduce_run<<<(uint32_t)(g ? g : 1), kBlock, 0, c->stream>>>(fp, (int)k, n, o->d, c->fi->P);
    } else {
        k_prod_reduce<<<grid_for(n), kBlock, 0, c->stream>>>(fp, (int)k, n, o->d, c->fi->P);
    }
    HIPCHK(hipGetLastError());
    *out = o;
    return ZK_OK;
}

extern "C" int32_t zk_product_evaluate(zk_ctx *c, const zk_mle *const *f, uint64_t k, const uint64_t *point,
                                       uint64_t n_point, uint64_t out[4]) {
    if (!out || (!point && n_point)) return ZK_ERR_BAD_ARG;
    ZKCHK(product_args(c, f, k));
    if (n_point != f[0]->n_vars) return ZK_ERR_EVAL_ARITY;   // product_poly.rs:37-39
    Fe prod = fe_one(c->fi->P);
    for (uint64_t i = 0; i < k; ++i) {                       // product_poly.rs:41-43
        uint64_t v[4];
        ZKCHK(zk_mle_evaluate(c, f[i], point, n_point, v));
        prod = fe_mul(prod, fe_from_u64limbs(v), c->fi->P);
    }
    fe_to_u64limbs(prod, out);
    return ZK_OK;
}

// ---- round machinery ------------------------------------------------------------------------------------------
// Per-prover device scratch: the word sponge, the current challenge, and the proof being assembled.  Nothing in the
// round loop waits on the host: k_round (+fold) -> k_round_tail (reduce + transcript) -> k_round (+fold) -> ...
constexpr size_t kChalWords = kChallengeBytes / 8;                                   // one challenge record, in u64
constexpr size_t kEpartBytes = (size_t)(kPipeMaxWorkBlocks + 2) * 16 * 32;         // E partials of one pipelined round (+ total + counter)
constexpr size_t kChalBlockBytes = 2 * kChallengeBytes + 32;   // two challenge records + the claim element
struct ProverScratch {
    WordSponge *d_sponge;
    uint64_t *d_challenge;   // TWO challenge records (round s uses slot s & 1): a pipelined launch reads r_{s-2} while r_{s-1} is written
    uint64_t *d_claim;       // one element behind them: the claim S_prev(r_prev) a SKIP1 round kernel parks for its tail (ClaimJob).  Per
                             // PROVER, not per context: the sharded prover keeps it live from round_begin to round_finish, across API
                             // calls in which other provers of the same context may run their own SKIP1 rounds
    uint64_t *d_epart;       // two E-partial buffers (pipelined rounds), same alternation
    uint64_t *d_rp;          // rounds * (D+1) elements
    uint64_t *d_ch;          // rounds elements
    uint64_t *d_final;       // kMaxFactors elements (same block as d_rp, d_ch)
    size_t rp_bytes, ch_bytes;
    bool external;           // sponge and the three outputs belong to a DeviceChain (not allocated / freed here)
};
// A caller that keeps ONE transcript on the device across several sumchecks (the GKR driver): the sponge already holds
// everything absorbed so far INCLUDING this sumcheck's claimed sum; round polynomials, challenges and the factor values at
// the point are written straight to the caller's device buffers; nothing is copied to the host and nothing waits.
struct DeviceChain {
    WordSponge *d_sponge;
    uint64_t *d_rp, *d_ch, *d_final;
    uint64_t *d_epart;   // E-partial buffers + counters shared by the chain's sumchecks (counters zero between launches)
};
// The two last-block-done counters (8 bytes each) sit behind the two E-partial buffers.  A pipelined launch leaves its counter
// at zero, so they are cleared once per proof -- by the launch that stores the initial sponge (sponge_to_device), not by a
// memset of their own.
static inline uint64_t *epart_counters(uint64_t *d_epart) { return d_epart + 2 * (kEpartBytes / 8); }
// The proof being assembled is ONE device block [round polys | challenges | factor values at the point] so that it comes
// back in one copy (through pinned memory: a device-to-pageable copy blocks the host once per call).
static int32_t scratch_alloc(zk_ctx *c, ProverScratch &ps, uint64_t rounds, uint32_t D, const DeviceChain *chain = nullptr) {
    ps = {};
    ps.rp_bytes = (size_t)(rounds ? rounds : 1) * (D + 1) * 32;
    ps.ch_bytes = (size_t)(rounds ? rounds : 1) * 32;
    if (chain) {
        ps.external = true;
        ps.d_sponge = chain->d_sponge;
        ps.d_rp = chain->d_rp;
        ps.d_ch = chain->d_ch;
        ps.d_final = chain->d_final;
        ps.d_epart = chain->d_epart;
        ZKCHK(pool_alloc(c, kChalBlockBytes, (void **)&ps.d_challenge));
        ps.d_claim = ps.d_challenge + 2 * kChalWords;
        return ZK_OK;
    }
    ZKCHK(pool_alloc(c, sizeof(WordSponge), (void **)&ps.d_sponge));
    ZKCHK(pool_alloc(c, kChalBlockBytes, (void **)&ps.d_challenge));
    ps.d_claim = ps.d_challenge + 2 * kChalWords;
    ZKCHK(pool_alloc(c, 2 * kEpartBytes + 16, (void **)&ps.d_epart));   // counters: cleared by sponge_to_device
    ZKCHK(pool_alloc(c, ps.rp_bytes + ps.ch_bytes + kMaxFactors * 32, (void **)&ps.d_rp));
    ps.d_ch = ps.d_rp + ps.rp_bytes / 8;
    ps.d_final = ps.d_ch + ps.ch_bytes / 8;
    return ZK_OK;
}
static void scratch_free(zk_ctx *c, ProverScratch &ps) {
    if (ps.external) {
        pool_free(c, ps.d_challenge, kChalBlockBytes);
        ps = {};
        return;
    }
    pool_free(c, ps.d_sponge, sizeof(WordSponge));
    pool_free(c, ps.d_challenge, kChalBlockBytes);
    pool_free(c, ps.d_epart, 2 * kEpartBytes + 16);
    pool_free(c, ps.d_rp, ps.rp_bytes + ps.ch_bytes + kMaxFactors * 32);
    ps = {};
}
// pinned staging for results, grown on demand
static int32_t results_staging(zk_ctx *c, size_t bytes, uint8_t **out) {
    if (c->h_results_bytes < bytes) {
        if (c->h_results) (void)hipHostFree(c->h_results);
        c->h_results = nullptr;
        c->h_results_bytes = 0;
        size_t cap = 1 << 16;
        while (cap < bytes) cap <<= 1;
        HIPCHK(hipHostMalloc((void **)&c->h_results, cap, kPolledHostFlags));
        c->h_results_bytes = cap;
    }
    *out = c->h_results;
    return ZK_OK;
}

// the block partials of a round: the context's buffer -- inside a batch (launch.hpp BatchRecorder) each proof has its own eighth of it
static inline uint64_t partials_capacity() { return g_batch ? (uint64_t)kMaxGrid * kMaxSums / kMaxBatch : (uint64_t)kMaxGrid * kMaxSums; }
static inline uint64_t *partials_of(zk_ctx *c) { return g_batch ? c->d_partials + (size_t)g_batch->cur * partials_capacity() * 4 : c->d_partials; }
static inline RoundLaunchCtx launch_ctx(zk_ctx *c) {
    RoundLaunchCtx lc = {c->stream, &c->fi->P, partials_of(c), partials_capacity(), {}};
    return lc;
}
// k_round_tail on the current proof's partials: launched, or recorded for the batch's merged launch (one transcript block per proof)
// the initial sponge of a single proof, not stored yet: the first classic tail takes it as an argument (k_round_tail_init); any other
// first consumer stores it first (flush_pending_sponge)
struct PendingSponge {
    WordSponge w;
    WordSponge *dst;
    uint64_t *zero2;
    bool valid;
};
static int32_t launch_tail(zk_ctx *c, uint32_t nblocks, uint32_t ns, WordSponge *sponge, uint64_t *out_rp, uint64_t *out_ch, uint64_t *d_challenge,
                           uint64_t *lanes, const TailDerive &dv, const PendingSponge *init = nullptr) {
    const uint64_t *part = partials_of(c);
    hipStream_t st = c->stream;
    const FieldParams *P = &c->fi->P;
    if (init && init->valid && sponge == init->dst && !lanes && !g_batch) {
        k_round_tail_init<<<1, kBlock, 0, st>>>(part, nblocks, ns, sponge, out_rp, out_ch, d_challenge, *P, dv, init->w, init->zero2);
        HIPCHK(hipGetLastError());
        return ZK_OK;
    }
    if (init && init->valid) {   // (cannot happen with the callers as they are: the state must reach the device before anything reads it)
        k_store_sponge<<<1, 64, 0, st>>>(init->w, init->dst, init->zero2);
        HIPCHK(hipGetLastError());
    }
    auto single = [=]() {
        k_round_tail<<<1, kBlock, 0, st>>>(part, nblocks, ns, sponge, out_rp, out_ch, d_challenge, lanes, *P, dv);
        return hipGetLastError();
    };
    if (!lanes && batch_record(BK_TAIL, 0, 1, kBlock, 0, nblocks, ns, 0, 0, TailSlot{part, sponge, out_rp, out_ch, d_challenge, dv}, single)) return ZK_OK;
    HIPCHK(single());
    return ZK_OK;
}
// ZK_CLAIM_IN_ROUND=0: the tails evaluate the SKIP1 claim themselves (round 4's behaviour; A/B)
static bool claim_in_round() {
    static const bool v = env_u64("ZK_CLAIM_IN_ROUND", 1, 0, 1) != 0;
    return v;
}
static inline bool fast_degree(uint32_t D) { return D >= 1 && D <= 4; }

// Where a round's sums go after the second-stage reduction.
struct TailTargets {
    WordSponge *sponge;     // run the transcript step (single-GPU prover)
    uint64_t *out_rp;       // round polynomial, D+1 elements (device)
    uint64_t *out_ch;       // challenge record (device), may be null
    uint64_t *d_challenge;  // challenge for the next fused fold
    uint64_t *lanes;        // 32-bit digit lanes for the cross-GPU all-reduce, may be null
    const PendingSponge *init;   // non-null: the proof's initial sponge has not been stored -- this round's tail takes it as an argument
    uint64_t *claim_park;   // one element owned by the prover (ProverScratch::d_claim); null: no SKIP1 round is possible (no previous round)
    TailDerive *lanes_dv;   // with lanes (host memory, may be null): out -- what k_lanes_transcript has to derive after the all-reduce
                            // (S(1) from the claim, S(D) from the leading coefficient); null: the round kernels compute every sum
};
static std::vector<Fe> interp_weights(uint32_t D, const FieldParams &P);   // defined with the verifier
// Rounds with at least this many pairs leave out the t = 1 sums (k_round_kd SKIP1 + TailDerive): below it the extra
// D + 1 dependent multiplies in the tail cost more than the products they save.  ZK_SKIP1_MIN_PAIRS overrides (tests).
static uint64_t skip1_min_pairs() {
    static const uint64_t v = env_u64("ZK_SKIP1_MIN_PAIRS", (uint64_t)1 << 16, 1, (uint64_t)1 << 40);
    return v;
}
static inline TermSpec single_term(int k) {
    TermSpec ts = {1, {k, 0, 0, 0}};
    return ts;
}
// sums of the current tables (already folded) -> targets.  Handles every degree.  With several terms each term's round
// kernel writes its own range of block partials and the one tail reduction adds them all (the sum over terms is free).
// dv (optional): previous round polynomial + Lagrange weights; allows the big fused rounds to skip the t = 1 sums.
// defer (optional): when the sums take the one-launch fast path, do NOT launch k_round_tail; report what it would have
// reduced (the caller merges the tail into the next pipelined launch).  defer->blocks stays 0 when the tail was launched.
struct DeferredTail {
    uint32_t blocks;
    bool skip1;
    bool lead;    // slot D of the partials holds the leading coefficient (k_round_kd LEAD)
    const uint64_t *claim;   // skip1: where the round kernel's claim workgroup left S_prev(r_prev) (null: the tail evaluates it)
};
// Rounds with at least this many pairs accumulate the leading coefficient instead of S(D) (k_round_kd LEAD): one modular
// addition per factor and pair index fewer against three to six more in the tail.  ZK_LEAD_MIN_PAIRS overrides (tests).
static uint64_t lead_min_pairs() {
    static const uint64_t v = env_u64("ZK_LEAD_MIN_PAIRS", (uint64_t)1 << 16, 1, (uint64_t)1 << 40);
    return v;
}
// ZK_SHARD_SKIP1=0: the sharded prover's round kernels compute every sum themselves (round 4's behaviour; A/B)
static bool shard_skip_on() {
    static const bool v = env_u64("ZK_SHARD_SKIP1", 1, 0, 1) != 0;
    return v;
}
static int32_t launch_sums(zk_ctx *c, const FactorPtrs &fp, const TermSpec &ts, uint64_t q, uint32_t D, bool fused,
                           const uint64_t *d_r, const TailTargets &tt, const TailDerive *dv = nullptr, DeferredTail *defer = nullptr) {
    if (defer) defer->blocks = 0;
    const FieldParams &P = c->fi->P;
    if (fast_degree(D)) {
        uint32_t total = 0;
        int first = 0;
        // (sharded: both variants accumulate quantities that are linear in the shards, so they survive the all-reduce; the
        // derivations move behind it -- tt.lanes_dv tells zk_shard_prover_round_finish what to derive)
        const bool may_derive = !tt.lanes || (tt.lanes_dv && shard_skip_on());
        bool skip1 = dv && dv->prev_rp && tt.claim_park && fused && may_derive && D <= (uint32_t)kMaxSkipDegree && q >= skip1_min_pairs();
        bool lead = may_derive && q >= lead_min_pairs();   // (the launchers clear it for shapes without the variant)
        // a SKIP1 kernel, if one is launched below, evaluates the claim its tail needs beside its work blocks (ClaimJob)
        ClaimJob cjob = {};
        if (skip1 && claim_in_round()) cjob = ClaimJob{dv->prev_rp, dv->prev_chal, dv->w, tt.claim_park, D + 1};
        auto with_claim = [&](RoundLaunchCtx lc) {
            lc.claim = cjob;
            return lc;
        };
        auto tail_dv = [&]() {
            TailDerive t = skip1 && dv ? *dv : TailDerive{};
            t.lead = lead ? D : 0;
            t.claim = (t.prev_rp && cjob.out) ? cjob.out : nullptr;
            if (tt.lanes) {
                if (tt.lanes_dv) *tt.lanes_dv = t;
                t.local_only = 1;
            }
            return t;
        };
        if (tt.lanes_dv) *tt.lanes_dv = TailDerive{};
        if (ts.n_terms == 2 && ts.term_k[1] == 1) {   // product + one single-factor term (a GKR layer): one pass
            uint32_t g = 0;
            const int lrc = launch_round_plus1(with_claim(launch_ctx(c)), fp, ts.term_k[0], q, D, fused, d_r, &g, &skip1, &lead);
            if (lrc == kLaunchHipError) {
                g_hip_err = "round kernel launch failed";
                return ZK_ERR_HIP;
            }
            if (lrc == kLaunchOk) {
                if (defer) {
                    defer->blocks = g;
                    defer->skip1 = skip1;
                    defer->claim = skip1 ? cjob.out : nullptr;
                    defer->lead = lead;
                    return ZK_OK;
                }
                return launch_tail(c, g, D + 1, tt.sponge, tt.out_rp, tt.out_ch, tt.d_challenge, tt.lanes, tail_dv(), tt.init);
            }
        }
        if (ts.n_terms != 1) {
            skip1 = lead = false;
            cjob = {};   // no SKIP1 kernel below: no claim workgroup either
        }
        for (int i = 0; i < ts.n_terms; ++i) {
            FactorPtrs sub = {};
            for (int f = 0; f < ts.term_k[i]; ++f) {
                sub.in[f] = fp.in[first + f];
                sub.out[f] = fp.out[first + f];
            }
            RoundLaunchCtx lc = with_claim(launch_ctx(c));
            lc.d_partials += (size_t)total * (D + 1) * 4;
            lc.capacity_elems -= (uint64_t)total * (D + 1);
            uint32_t g = 0;
            const int lrc = launch_round(lc, sub, ts.term_k[i], q, D, fused, d_r, &g, &skip1, &lead);
            if (lrc == kLaunchUnsupported) return ZK_ERR_UNSUPPORTED;
            if (lrc != kLaunchOk) {
                g_hip_err = "round kernel launch failed";
                return ZK_ERR_HIP;
            }
            total += g;
            first += ts.term_k[i];
        }
        if (defer) {
            defer->blocks = total;
            defer->skip1 = skip1;
            defer->claim = skip1 ? cjob.out : nullptr;
            defer->lead = lead;
            return ZK_OK;
        }
        return launch_tail(c, total, D + 1, tt.sponge, tt.out_rp, tt.out_ch, tt.d_challenge, tt.lanes, tail_dv(), tt.init);
    }
    if (ts.n_terms != 1) return ZK_ERR_UNSUPPORTED;
    const int k = ts.term_k[0];
    // any other degree (0, or > 4): one pass per evaluation point over tables that are already folded
    for (uint32_t t = 0; t <= D; ++t) {
        uint32_t g = 0;
        if (launch_round_single_t(launch_ctx(c), fp, k, q, fe_from_u32(t, P), &g) != kLaunchOk) {
            g_hip_err = "round kernel launch failed";
            return ZK_ERR_HIP;
        }
        k_round_tail<<<1, kBlock, 0, c->stream>>>(c->d_partials, g, 1, nullptr, c->d_sums + 4 * t, nullptr, nullptr, nullptr, P);
        HIPCHK(hipGetLastError());
    }
    k_round_tail<<<1, kBlock, 0, c->stream>>>(c->d_sums, 1, D + 1, tt.sponge, tt.out_rp, tt.out_ch, tt.d_challenge, tt.lanes, P);
    HIPCHK(hipGetLastError());
    return ZK_OK;
}

extern "C" int32_t zk_round_sums(zk_ctx *c, const zk_mle *const *f, uint64_t k, uint32_t D, uint64_t *out) {
    if (!out) return ZK_ERR_BAD_ARG;
    ZKCHK(product_args(c, f, k));
    if (D >= kMaxSums) return ZK_ERR_UNSUPPORTED;
    if (f[0]->n_vars == 0) return ZK_ERR_PANIC_INDEX;   // partial_evaluate(0, [..]) on a 0-variable poly panics
    FactorPtrs fp = {};
    for (uint64_t i = 0; i < k; ++i) fp.in[i] = f[i]->d;
    const uint64_t q = 1ull << (f[0]->n_vars - 1);
    uint64_t *d_out = c->d_sums + 4 * kMaxSums;   // second third of d_sums
    TailTargets tt = {nullptr, d_out, nullptr, nullptr, nullptr, nullptr, nullptr};
    ZKCHK(launch_sums(c, fp, single_term((int)k), q, D, false, nullptr, tt));
    HIPCHK(hipMemcpyAsync(out, d_out, (size_t)(D + 1) * 32, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return ZK_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Transcript (host)
// ------------------------------------------------------------------------------------------------------------
extern "C" int32_t zk_transcript_new(zk_transcript **out) {
    if (!out) return ZK_ERR_BAD_ARG;
    zk_transcript *t = new (std::nothrow) zk_transcript();
    if (!t) return ZK_ERR_ALLOC;
    t->sp.init();
    *out = t;
    return ZK_OK;
}
extern "C" int32_t zk_transcript_free(zk_transcript *t) {
    delete t;
    return ZK_OK;
}
extern "C" int32_t zk_transcript_append(zk_transcript *t, const uint8_t *data, size_t len) {
    if (!t || (!data && len)) return ZK_ERR_BAD_ARG;
    t->sp.update(data, len);
    return ZK_OK;
}
extern "C" int32_t zk_transcript_sample_challenge(zk_transcript *t, uint8_t out[32]) {
    if (!t || !out) return ZK_ERR_BAD_ARG;
    t->sp.sample_challenge(out);
    return ZK_OK;
}
extern "C" int32_t zk_transcript_sample_field_element(zk_transcript *t, int32_t field, uint64_t out[4]) {
    if (!t || !out) return ZK_ERR_BAD_ARG;
    const FieldInfo *fi = field_info(field);
    if (!fi) return ZK_ERR_BAD_FIELD;
    uint8_t h[32];
    t->sp.sample_challenge(h);                                           // transcript/src/lib.rs:28
    fe_to_u64limbs(fe_from_be_bytes_mod_order(h, 32, fi->P), out);       // :29
    return ZK_OK;
}
extern "C" int32_t zk_transcript_sample_n_field_elements(zk_transcript *t, int32_t field, uint64_t n, uint64_t *out) {
    if (!t || (!out && n)) return ZK_ERR_BAD_ARG;
    if (!field_info(field)) return ZK_ERR_BAD_FIELD;
    for (uint64_t i = 0; i < n; ++i) ZKCHK(zk_transcript_sample_field_element(t, field, out + 4 * i));   // transcript/src/lib.rs:32-34
    return ZK_OK;
}
extern "C" int32_t zk_keccak256(const uint8_t *data, size_t len, uint8_t out[32]) {
    if ((!data && len) || !out) return ZK_ERR_BAD_ARG;
    Sponge s;
    s.init();
    s.update(data, len);
    s.finalize_reset(out);
    return ZK_OK;
}

static void absorb_elements(Sponge &sp, const uint64_t *elems, uint64_t n, const FieldParams &P) {   // sumcheck/src/lib.rs:23-29
    uint8_t b[32];
    for (uint64_t i = 0; i < n; ++i) {
        fe_to_bytes_be(fe_from_u64limbs(elems + 4 * i), P, b);
        sp.update(b, 32);
    }
}
static Fe squeeze_field_element(Sponge &sp, const FieldParams &P) {   // transcript/src/lib.rs:27-30
    uint8_t h[32];
    sp.sample_challenge(h);
    return fe_from_be_bytes_mod_order(h, 32, P);
}
// absorb poly.to_bytes() (product_poly.rs:77-83) -- device serialiser, chunked D2H, host sponge.  The Keccak sponge is
// serial by construction and runs on the host (one GPU wave permutes 136 bytes in ~3 us = 45 MB/s; a host core does
// 0.4-0.75 GB/s with keccak_host::rounds), so it bounds `prove`; the serialiser kernel and the copy of chunk i+1 run while the host absorbs
// chunk i (two device + two pinned buffers, one event each; the pinned buffers stay with the context).
// consume(host_ptr, bytes) is called once per 16-MiB chunk, in order, on the calling thread, while the next chunk is serialised and copied
template <class Consume>
static int32_t stream_table_bytes(zk_ctx *c, const zk_mle *const *f, uint64_t k, Consume &&consume) {
    const uint64_t n = 1ull << f[0]->n_vars;
    const uint64_t chunk = n < (1ull << 19) ? n : (1ull << 19);   // 16 MiB of bytes per chunk
    const size_t cb = (size_t)chunk * 32;
    const uint64_t per_table = n / chunk, total = per_table * k;
    if (c->h_absorb_bytes < cb) {
        for (int b = 0; b < 2; ++b) {
            if (c->h_absorb[b]) (void)hipHostFree(c->h_absorb[b]);
            c->h_absorb[b] = nullptr;
        }
        c->h_absorb_bytes = 0;
        for (int b = 0; b < 2; ++b) HIPCHK(hipHostMalloc((void **)&c->h_absorb[b], cb, hipHostMallocDefault));
        c->h_absorb_bytes = cb;
    }
    if (!c->ev_absorb[0])
        for (int b = 0; b < 2; ++b) HIPCHK(hipEventCreateWithFlags(&c->ev_absorb[b], hipEventDisableTiming));
    uint8_t *d_bytes[2] = {nullptr, nullptr};
    ZKCHK(pool_alloc(c, cb, (void **)&d_bytes[0]));
    int32_t rc = pool_alloc(c, cb, (void **)&d_bytes[1]);
    auto enqueue = [&](uint64_t i) -> int32_t {
        const int b = (int)(i & 1);
        const uint64_t *src = f[i / per_table]->d + 4 * (i % per_table) * chunk;
        k_to_bytes<<<grid_for(chunk), kBlock, 0, c->stream>>>(src, d_bytes[b], chunk, c->fi->P);
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(c->h_absorb[b], d_bytes[b], cb, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipEventRecord(c->ev_absorb[b], c->stream) != hipSuccess)
            return ZK_ERR_HIP;
        return ZK_OK;
    };
    if (rc == ZK_OK) rc = enqueue(0);
    for (uint64_t i = 0; i < total && rc == ZK_OK; ++i) {
        if (i + 1 < total) rc = enqueue(i + 1);   // its buffers were released when chunk i-1 was consumed
        if (rc == ZK_OK && hipEventSynchronize(c->ev_absorb[i & 1]) != hipSuccess) rc = ZK_ERR_HIP;
        if (rc == ZK_OK) consume(c->h_absorb[i & 1], cb);
    }
    if (rc != ZK_OK) (void)hipStreamSynchronize(c->stream);
    pool_free(c, d_bytes[0], cb);
    if (d_bytes[1]) pool_free(c, d_bytes[1], cb);
    return rc;
}
static int32_t absorb_tables(zk_ctx *c, Sponge &sp, zk_mle *const *f, uint64_t k) {
    return stream_table_bytes(c, (const zk_mle *const *)f, k, [&](const uint8_t *p, size_t bytes) { sp.update(p, bytes); });
}

// ------------------------------------------------------------------------------------------------------------
// SumcheckProver -- device-resident round loop (prover.rs:33-73)
// ------------------------------------------------------------------------------------------------------------
struct RoundState {
    zk_ctx *c;
    uint64_t k;
    uint64_t vars_left;               // variables of the tables in `cur` (before any pending fold)
    uint64_t round;                   // rounds completed
    uint32_t D;
    bool pending_fold;                // the last challenge has not been applied to `cur` yet (it is fused into the next round)
    bool first_out_of_place;          // next fold must leave `cur` intact (caller keeps the inputs): write to scratch
    uint64_t *cur[kMaxFactors];       // current tables (device)
    uint64_t *scratch[kMaxFactors];   // owned tables
    size_t scratch_bytes[kMaxFactors];
    ProverScratch ps;
    TermSpec terms;                   // how the k flat factors group into products (one term = ProductPoly)
    uint64_t *d_final;                // optional (= ps.d_final when requested): the factors at the challenge point
    TailDerive dv;                    // Lagrange weights on 0..D (prev_rp is set per round)
    // pipelined rounds (pipe_kernels.cuh): the E partials of round `round` already exist (computed from `cur`, the table of
    // round - 1, before its challenge was known); `cur` still awaits that fold (pending_fold is true)
    bool pipe_active;
    uint32_t pipe_blocks;             // work blocks that wrote them
    bool pipe_total;                  // slot 0 of their buffer holds the total (k_round_pipe: the block that finishes last adds them up);
                                      // false: the next launch's transcript block (or the finisher) adds the pipe_blocks partials up
    PendingSponge init;               // valid: the initial sponge is still on the host side of the launch queue (prove_core)
    FinishPublish pub;                // flag != null: the pipelined finisher, being the call's last launch, publishes the proof block itself
    bool published;                   // ... and has been enqueued with that job
};
// challenge records alternate between two slots: round s publishes into slot s & 1
static inline uint64_t *chal_of_round(const RoundState &st, uint64_t round) { return st.ps.d_challenge + (round & 1) * kChalWords; }
static inline uint64_t *chal_cur(const RoundState &st) { return chal_of_round(st, st.round); }        // this round's (to be written)
static inline uint64_t *chal_prev(const RoundState &st) { return chal_of_round(st, st.round + 1); }   // round - 1 (same parity as round + 1)
static inline uint64_t *epart_of_round(const RoundState &st, uint64_t round) { return st.ps.d_epart + (round & 1) * (kEpartBytes / 8); }
static inline uint32_t *epart_counter(const RoundState &st, uint64_t round) {
    return reinterpret_cast<uint32_t *>(st.ps.d_epart + 2 * (kEpartBytes / 8) + (round & 1));
}
static void round_state_release(RoundState &st) {
    for (uint64_t i = 0; i < (uint64_t)kMaxFactors; ++i)
        if (st.scratch[i]) {
            pool_free(st.c, st.scratch[i], st.scratch_bytes[i]);
            st.scratch[i] = nullptr;
        }
    st.d_final = nullptr;
    scratch_free(st.c, st.ps);
}
static int32_t round_state_init(RoundState &st, zk_ctx *c, zk_mle *const *f, uint64_t k, uint32_t D, bool consume,
                                uint64_t total_rounds, const DeviceChain *chain = nullptr) {
    st.c = c;
    st.k = k;
    st.vars_left = f[0]->n_vars;
    st.round = 0;
    st.D = D;
    st.pending_fold = false;
    st.first_out_of_place = !consume;
    st.ps = {};
    st.terms = single_term((int)k);
    st.d_final = nullptr;
    st.dv = {};
    st.pipe_active = false;
    st.pipe_blocks = 0;
    st.pipe_total = true;
    st.pub = {};
    st.published = false;
    st.init.valid = false;
    for (uint64_t i = 0; i < (uint64_t)kMaxFactors; ++i) {
        st.cur[i] = i < k ? f[i]->d : nullptr;
        st.scratch[i] = nullptr;
        st.scratch_bytes[i] = 0;
    }
    int32_t rc = scratch_alloc(c, st.ps, total_rounds, D, chain);
    if (rc == ZK_OK && D >= 1 && D <= (uint32_t)kMaxSkipDegree) {
        auto it = c->lagrange_w.find(D);
        if (it == c->lagrange_w.end()) {
            const std::vector<Fe> w = interp_weights(D, c->fi->P);
            std::vector<uint64_t> limbs(4 * (size_t)(D + 1));
            for (uint32_t t = 0; t <= D; ++t) fe_to_u64limbs(w[t], limbs.data() + 4 * t);
            uint64_t *d_w = nullptr;
            rc = raw_alloc(c, limbs.size() * 8, (void **)&d_w);
            if (rc == ZK_OK && hipMemcpy(d_w, limbs.data(), limbs.size() * 8, hipMemcpyHostToDevice) != hipSuccess) {
                (void)hipFree(d_w);
                d_w = nullptr;
                rc = ZK_ERR_HIP;
            }
            if (rc == ZK_OK) it = c->lagrange_w.emplace(D, d_w).first;
        }
        if (rc == ZK_OK) st.dv.w = it->second;
    }
    if (rc == ZK_OK && !consume && st.vars_left >= 2)
        for (uint64_t i = 0; i < k && rc == ZK_OK; ++i) {
            st.scratch_bytes[i] = (size_t)32 << (st.vars_left - 1);
            rc = pool_alloc(c, st.scratch_bytes[i], (void **)&st.scratch[i]);
        }
    if (rc != ZK_OK) round_state_release(st);
    return rc;
}
// Enqueue the next round: apply the pending fold (prover.rs:64 of the previous round, fused) and compute this round's
// sums (prover.rs:49-56).  `lanes` selects the sharded form (sums -> digit lanes, transcript deferred).
// store the proof's initial sponge with a launch of its own (what every proof did until round 6): for first transcript steps that are
// not a classic tail (a pipelined launch, a finisher) and for the batch / sharded provers
static int32_t flush_pending_sponge(RoundState &st) {
    if (!st.init.valid) return ZK_OK;
    st.init.valid = false;
    k_store_sponge<<<1, 64, 0, st.c->stream>>>(st.init.w, st.init.dst, st.init.zero2);
    HIPCHK(hipGetLastError());
    return ZK_OK;
}
static int32_t round_enqueue(RoundState &st, uint64_t *lanes, DeferredTail *defer = nullptr, TailDerive *lanes_dv = nullptr) {
    zk_ctx *c = st.c;
    if (st.pending_fold) st.vars_left -= 1;           // tables shrink by the fold fused into this launch
    const uint64_t m = st.vars_left;                  // variables of this round's table
    if (m == 0) return ZK_ERR_BAD_ARG;
    const uint64_t q = 1ull << (m - 1);
    FactorPtrs fp = {};
    for (uint64_t i = 0; i < st.k; ++i) {
        fp.in[i] = st.cur[i];
        fp.out[i] = (st.pending_fold && st.first_out_of_place && st.scratch[i]) ? st.scratch[i] : st.cur[i];   // else in place
    }
    TailTargets tt;
    tt.sponge = lanes ? nullptr : st.ps.d_sponge;
    tt.out_rp = st.ps.d_rp + st.round * (st.D + 1) * 4;
    tt.out_ch = st.ps.d_ch + st.round * 4;
    tt.d_challenge = chal_cur(st);
    tt.lanes = lanes;
    tt.claim_park = st.ps.d_claim;
    tt.lanes_dv = lanes_dv;
    // the initial sponge, if it has not been stored yet: this round's tail takes it as an argument when it is a classic tail launched right
    // here (launch_sums on a supported degree, not deferred into a pipelined launch, not sharded); else it is stored now
    tt.init = nullptr;
    if (st.init.valid) {
        if (!lanes && !defer && fast_degree(st.D) && !st.pending_fold) tt.init = &st.init;
        else ZKCHK(flush_pending_sponge(st));
    }
    int32_t rc;
    if (st.pending_fold && !fast_degree(st.D)) {
        // generic degree: fold as separate launches, then the per-point passes
        Fe r;
        (void)r;
        rc = ZK_OK;
        for (uint64_t i = 0; i < st.k && rc == ZK_OK; ++i) {
            k_fold_dev<<<grid_for(q * 2), kBlock, 0, c->stream>>>(fp.in[i], fp.out[i], q * 2, (uint32_t)m, c->fi->P, chal_prev(st));
            if (hipGetLastError() != hipSuccess) rc = ZK_ERR_HIP;
        }
        FactorPtrs g = {};
        for (uint64_t i = 0; i < st.k; ++i) g.in[i] = fp.out[i];
        if (rc == ZK_OK) rc = launch_sums(c, g, st.terms, q, st.D, false, nullptr, tt);
    } else {
        st.dv.prev_rp = st.round ? tt.out_rp - (size_t)(st.D + 1) * 4 : nullptr;
        st.dv.prev_chal = chal_prev(st);
        rc = launch_sums(c, fp, st.terms, q, st.D, st.pending_fold, chal_prev(st), tt, &st.dv, defer);
        if (tt.init) st.init.valid = false;   // (consumed by the tail; on an error the proof is abandoned anyway)
    }
    if (st.pending_fold) {
        for (uint64_t i = 0; i < st.k; ++i) st.cur[i] = fp.out[i];
        st.first_out_of_place = false;
    }
    st.pending_fold = true;    // this round's challenge gets applied by the next launch
    return rc;
}

// ---- pipelined rounds (pipe_kernels.cuh): host schedule -------------------------------------------------------------------
// Rounds with at most this many pairs have their sums prepared before their challenge exists (0 switches the pipeline
// off; ZK_PIPE_MAX_PAIRS overrides; the accumulators hold at most kMaxLazy products per lane, hence the cap).
static uint64_t pipe_max_pairs() {
    static const uint64_t v = [] {
        uint64_t x = env_u64("ZK_PIPE_MAX_PAIRS", (uint64_t)1 << 12, 0, (uint64_t)1 << 40);   // 0 = no pipelined rounds; capped below
        const uint64_t cap = (uint64_t)kPipeMaxWorkBlocks * 16 * kMaxLazy;   // products a lane may accumulate unreduced (16 rows per block)
        return x > cap ? cap : x;
    }();
    return v;
}
static inline uint64_t pipe_limit_pairs() { return pipe_max_pairs(); }
static bool pipe_shape(const RoundState &st, int *k, int *extra) {
    if (st.terms.n_terms == 1) {
        *k = st.terms.term_k[0];
        *extra = 0;
    } else if (st.terms.n_terms == 2 && st.terms.term_k[1] == 1) {
        *k = st.terms.term_k[0];
        *extra = 1;
    } else {
        return false;
    }
    return pipe_shape_ok(*k, st.D, *extra);
}
static constexpr size_t kDbgWords = 64 * 32 + 64 * 16;
static uint64_t *pipe_dbg_slot(zk_ctx *c, bool finisher) {
    static const bool on = env_flag("ZK_PIPE_DEBUG");
    if (!on) return nullptr;
    if (!c->d_dbg) {
        if (hipMalloc(&c->d_dbg, kDbgWords * 8) != hipSuccess) return nullptr;
        (void)hipMemset(c->d_dbg, 0, kDbgWords * 8);
    }
    if (finisher) return c->d_dbg + 64 * 32;
    const uint32_t i = c->dbg_launch++ % 64;
    return c->d_dbg + (size_t)i * 32;
}
static void pipe_dbg_dump(zk_ctx *c) {
    if (!c->d_dbg) return;
    std::vector<uint64_t> h(kDbgWords);
    if (hipMemcpy(h.data(), c->d_dbg, kDbgWords * 8, hipMemcpyDeviceToHost) != hipSuccess) return;
    uint64_t t0 = 0;
    for (uint32_t i = 0; i < c->dbg_launch && i < 64; ++i) {
        const uint64_t *d = h.data() + (size_t)i * 32;
        if (!t0) t0 = d[0];
        fprintf(stderr, "[pipe %2u] tail: start %7.2f red %5.2f eval %5.2f transcript %5.2f publish %5.2f | work: start %7.2f loop %5.2f close %5.2f store %5.2f\n", i,
                (d[0] - t0) / 100.0, (d[1] - d[0]) / 100.0, (d[2] - d[1]) / 100.0, (d[3] - d[2]) / 100.0, (d[4] - d[3]) / 100.0,
                d[8] ? (d[8] - t0) / 100.0 : 0.0, (d[9] - d[8]) / 100.0, (d[10] - d[9]) / 100.0, (d[11] - d[10]) / 100.0);
    }
    const uint64_t *f = h.data() + 64 * 32;
    for (uint32_t r = 0; r < 24 && f[16 * r]; ++r) {
        const uint64_t *d = f + 16 * r;
        fprintf(stderr, "[fin %2u] start %7.2f gather %5.2f eval %5.2f transcript %5.2f barrier %5.2f | work %5.2f (from %5.2f)\n", r, (d[0] - t0) / 100.0,
                (d[1] - d[0]) / 100.0, (d[2] - d[1]) / 100.0, (d[3] - d[2]) / 100.0, (d[4] - d[3]) / 100.0, d[8] ? (d[9] - d[8]) / 100.0 : 0.0,
                d[8] ? (d[8] - d[0]) / 100.0 : 0.0);
    }
    (void)hipMemset(c->d_dbg, 0, kDbgWords * 8);
    c->dbg_launch = 0;
}
// the pipelined finisher takes over from every state when the tables it would hold fit LDS (ZK_FINISH_PIPE=0: classic one)
static bool finish_pipe_on() {
    static const bool v = env_u64("ZK_FINISH_PIPE", 1, 0, 1) != 0;
    return v;
}
static bool finish_pipe_applies(const RoundState &st) {
    int k, extra;
    if (!finish_pipe_on() || !pipe_shape(st, &k, &extra)) return false;
    return st.vars_left >= 3 && st.vars_left - 1 <= (uint64_t)finish_pipe_max_vars(k + extra);
}
// May round (st.round + 1) be prepared by the pipeline, given that round st.round's table has m_s variables?
static bool pipe_wants_next(const RoundState &st, uint64_t m_s) {
    int k, extra;
    if (!pipe_limit_pairs() || !pipe_shape(st, &k, &extra)) return false;
    if (m_s < 3) return false;                                                       // the launches after it need 8 elements
    if (!finish_pipe_on() && m_s - 1 <= (uint64_t)kFinishVars) return false;          // the classic finisher takes round + 1
    return ((uint64_t)1 << (m_s - 2)) <= pipe_limit_pairs();
}
// every remaining round in one launch of the pipelined finisher, from whatever state the loop is in
static int32_t finish_pipe_enqueue(RoundState &st) {
    zk_ctx *c = st.c;
    int k = 0, extra = 0;
    (void)pipe_shape(st, &k, &extra);
    FinishPipeLaunch fl = {};
    fl.k = k;
    fl.extra = extra;
    fl.D = st.D;
    fl.m_in = (uint32_t)st.vars_left;
    fl.entry = st.pipe_active ? 2 : (st.pending_fold ? 1 : 0);
    fl.e_partials = epart_of_round(st, st.round);
    fl.e_blocks = 1;   // slot 0 of the buffer holds the total ...
    if (st.pipe_active && !st.pipe_total) {   // ... unless the launch before left its block partials only (slots 1 .. pipe_blocks)
        fl.e_partials += (size_t)pipe_values_per_block(k, st.D) * 4;
        fl.e_blocks = st.pipe_blocks;
    }
    fl.pc = c->pipe_consts;
    fl.chal_in = chal_prev(st);
    const uint64_t remaining = st.pending_fold ? st.vars_left - 1 : st.vars_left;
    fl.chal_last = chal_of_round(st, st.round + remaining - 1);
    fl.sponge = st.ps.d_sponge;
    fl.out_rp = st.ps.d_rp + st.round * (st.D + 1) * 4;
    fl.out_ch = st.ps.d_ch + st.round * 4;
    fl.out_final = st.d_final;
    fl.dbg = pipe_dbg_slot(c, true);
    fl.pub = st.pub;
    FactorPtrs fp = {};
    for (uint64_t i = 0; i < st.k; ++i) fp.in[i] = st.cur[i];
    const int lrc = launch_finish_pipe(launch_ctx(c), fp, fl);
    if (lrc != kLaunchOk) {
        g_hip_err = "pipelined finisher launch failed";
        return lrc == kLaunchUnsupported ? ZK_ERR_UNSUPPORTED : ZK_ERR_HIP;
    }
    st.round += remaining;
    st.vars_left = 0;
    st.pending_fold = false;
    st.pipe_active = false;
    st.published = st.pub.flag != nullptr;
    return ZK_OK;
}
static PipeTailArgs pipe_tail_args(const RoundState &st, int mode, const uint64_t *partials, uint32_t nblocks, uint32_t n_in) {
    PipeTailArgs ta = {};
    ta.dbg = pipe_dbg_slot(st.c, false);
    ta.partials = partials;
    ta.nblocks = nblocks;
    ta.n_in = n_in;
    ta.mode = mode;
    ta.chal_in = chal_prev(st);
    ta.chal_out = chal_cur(st);
    ta.sponge = st.ps.d_sponge;
    ta.out_rp = st.ps.d_rp + st.round * (st.D + 1) * 4;
    ta.out_ch = st.ps.d_ch + st.round * 4;
    ta.pc = st.c->pipe_consts;
    return ta;
}
// Classic round st.round whose sums kernel has just been launched with its tail deferred (dt): ONE launch closes it
// (transcript block) and prepares round + 1 from the table of this round (already folded: `cur`).
static int32_t pipe_enter(RoundState &st, const DeferredTail &dt) {
    zk_ctx *c = st.c;
    int k = 0, extra = 0;
    (void)pipe_shape(st, &k, &extra);
    PipeLaunch pl = {};
    pl.k = k;
    pl.extra = extra;
    pl.D = st.D;
    pl.fold = false;
    pl.emit = 1;
    pl.q = (uint64_t)1 << (st.vars_left - 2);   // vars_left = variables of this round's table (after its fold)
    pl.chal_fold = nullptr;
    pl.e_partials = epart_of_round(st, st.round + 1);
    pl.done_counter = epart_counter(st, st.round + 1);
    pl.tail = pipe_tail_args(st, 0, partials_of(c), dt.blocks, st.D + 1);
    if (dt.skip1) {
        pl.tail.dv = st.dv;
        pl.tail.dv.prev_rp = pl.tail.out_rp - (size_t)(st.D + 1) * 4;
        pl.tail.dv.prev_chal = chal_prev(st);
        pl.tail.dv.claim = dt.claim;
    }
    pl.tail.dv.lead = dt.lead ? st.D : 0;
    FactorPtrs fp = {};
    for (uint64_t i = 0; i < st.k; ++i) fp.in[i] = fp.out[i] = st.cur[i];
    uint32_t g = 0;
    const int lrc = launch_round_pipe(launch_ctx(c), fp, pl, &g);
    if (lrc != kLaunchOk) {
        g_hip_err = "pipelined round launch failed";
        return lrc == kLaunchUnsupported ? ZK_ERR_UNSUPPORTED : ZK_ERR_HIP;
    }
    st.pipe_active = true;
    st.pipe_blocks = g;
    st.pipe_total = true;
    return ZK_OK;
}
// Pipelined state at round s = st.round: cur = table of round s-1 (vars_left variables), its challenge r_{s-1} pending, E_s
// partials ready.  ONE launch: transcript block closes round s; work blocks fold cur at r_{s-1} and, when round s+1 stays in
// the pipeline, prepare E_{s+1}.
static int32_t pipe_step(RoundState &st) {
    zk_ctx *c = st.c;
    int k = 0, extra = 0;
    (void)pipe_shape(st, &k, &extra);
    const uint64_t m = st.vars_left;             // variables of cur = table of round s - 1
    const bool stay = pipe_wants_next(st, m - 1);
    PipeLaunch pl = {};
    pl.k = k;
    pl.extra = extra;
    pl.D = st.D;
    pl.fold = true;
    pl.emit = stay ? 1 : 0;
    pl.q = m >= 3 ? (uint64_t)1 << (m - 3) : 0;
    if (pl.q == 0) return ZK_ERR_BAD_ARG;        // cannot happen: the finisher takes tables this small
    pl.chal_fold = chal_prev(st);
    pl.e_partials = epart_of_round(st, st.round + 1);
    pl.done_counter = epart_counter(st, st.round + 1);
    pl.tail = pipe_tail_args(st, 1, epart_of_round(st, st.round), 1, pipe_values_per_block(k, st.D));   // slot 0: the total
    if (!st.pipe_total) {   // the launch before left its block partials only: this transcript block adds them up
        pl.tail.partials = epart_of_round(st, st.round) + (size_t)pipe_values_per_block(k, st.D) * 4;
        pl.tail.nblocks = st.pipe_blocks;
    }
    FactorPtrs fp = {};
    for (uint64_t i = 0; i < st.k; ++i) {
        fp.in[i] = st.cur[i];
        fp.out[i] = (st.first_out_of_place && st.scratch[i]) ? st.scratch[i] : st.cur[i];
    }
    uint32_t g = 0;
    const int lrc = launch_round_pipe(launch_ctx(c), fp, pl, &g);
    if (lrc != kLaunchOk) {
        g_hip_err = "pipelined round launch failed";
        return lrc == kLaunchUnsupported ? ZK_ERR_UNSUPPORTED : ZK_ERR_HIP;
    }
    for (uint64_t i = 0; i < st.k; ++i) st.cur[i] = fp.out[i];
    st.first_out_of_place = false;
    st.vars_left = m - 1;                        // cur = table of round s, r_s pending
    st.pipe_active = stay;
    st.pipe_blocks = g;
    st.pipe_total = true;
    ++st.round;
    return ZK_OK;
}
// One step of the single-GPU round loop (prover.rs:44-68): everything that can be enqueued for round st.round.
static int32_t prover_step(RoundState &st, bool *finished_in_kernel);

// ---- finisher: every remaining round in one single-workgroup launch (k_finish) ----
template <int K, int D>
static int32_t launch_finish(zk_ctx *c, const FactorPtrs &fp, uint32_t m_in, int pending, uint64_t *d_challenge, WordSponge *sp,
                             uint64_t *out_rp, uint64_t *out_ch, uint64_t *out_final) {
    const uint32_t m = pending ? m_in - 1 : m_in;
    const size_t lds = (size_t)K * ((size_t)32 << m) + (kBlock / 64) * (D + 1) * 32 + (D + 1) * 32 + 48;
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_finish<K, D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipStream_t st = c->stream;
    const FieldParams *P = &c->fi->P;
    auto single = [=]() {
        k_finish<K, D><<<1, kBlock, lds, st>>>(fp, m_in, pending, *P, d_challenge, sp, out_rp, out_ch, out_final);
        return hipGetLastError();
    };
    if (batch_record_other(single)) return ZK_OK;   // (no batched twin: a batch replays the classic finisher proof by proof)
    HIPCHK(single());
    return ZK_OK;
}
static bool finish_shape_ok(uint64_t k, uint32_t D) {
    const int shape = (int)k * 10 + (int)D;
    return shape == 11 || shape == 12 || shape == 21 || shape == 22 || shape == 23 || shape == 32 || shape == 33;
}
// Runs rounds st.round .. (st.round + remaining - 1) where remaining = variables left after the pending fold.
static int32_t finish_enqueue(RoundState &st) {
    zk_ctx *c = st.c;
    FactorPtrs fp = {};
    for (uint64_t i = 0; i < st.k; ++i) fp.in[i] = st.cur[i];
    const int pending = st.pending_fold ? 1 : 0;
    const uint32_t m_in = (uint32_t)st.vars_left;
    const uint64_t remaining = pending ? st.vars_left - 1 : st.vars_left;
    uint64_t *out_rp = st.ps.d_rp + st.round * (st.D + 1) * 4, *out_ch = st.ps.d_ch + st.round * 4;
    int32_t rc = ZK_ERR_UNSUPPORTED;
    if (st.terms.n_terms > 1) {
        const uint32_t m = pending ? m_in - 1 : m_in;
        const size_t lds = (size_t)st.k * ((size_t)32 << m) + (kBlock / 64) * (st.D + 1) * 32 + (st.D + 1) * 32 + 48;
        const FieldParams &P = c->fi->P;
#define ZK_FINISH_TERMS(DD)                                                                                                     \
    case DD:                                                                                                                    \
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_finish_terms<DD>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds) != hipSuccess)                                                                        \
            return ZK_ERR_HIP;                                                                                                  \
        k_finish_terms<DD><<<1, kBlock, lds, c->stream>>>(fp, st.terms, m_in, pending, P, chal_prev(st), st.ps.d_sponge,     \
                                                          out_rp, out_ch, st.d_final);                                          \
        break;
        switch (st.D) {
            ZK_FINISH_TERMS(1)
            ZK_FINISH_TERMS(2)
            ZK_FINISH_TERMS(3)
            ZK_FINISH_TERMS(4)
            default: return ZK_ERR_UNSUPPORTED;
        }
#undef ZK_FINISH_TERMS
        rc = hipGetLastError() == hipSuccess ? ZK_OK : ZK_ERR_HIP;
    } else {
        switch ((int)st.k * 10 + (int)st.D) {
            case 11: rc = launch_finish<1, 1>(c, fp, m_in, pending, chal_prev(st), st.ps.d_sponge, out_rp, out_ch, st.d_final); break;
            case 12: rc = launch_finish<1, 2>(c, fp, m_in, pending, chal_prev(st), st.ps.d_sponge, out_rp, out_ch, st.d_final); break;
            case 21: rc = launch_finish<2, 1>(c, fp, m_in, pending, chal_prev(st), st.ps.d_sponge, out_rp, out_ch, st.d_final); break;
            case 22: rc = launch_finish<2, 2>(c, fp, m_in, pending, chal_prev(st), st.ps.d_sponge, out_rp, out_ch, st.d_final); break;
            case 23: rc = launch_finish<2, 3>(c, fp, m_in, pending, chal_prev(st), st.ps.d_sponge, out_rp, out_ch, st.d_final); break;
            case 32: rc = launch_finish<3, 2>(c, fp, m_in, pending, chal_prev(st), st.ps.d_sponge, out_rp, out_ch, st.d_final); break;
            case 33: rc = launch_finish<3, 3>(c, fp, m_in, pending, chal_prev(st), st.ps.d_sponge, out_rp, out_ch, st.d_final); break;
            default: break;
        }
    }
    if (rc == ZK_OK) {
        st.round += remaining;
        st.vars_left = 0;
        st.pending_fold = false;
    }
    return rc;
}
static inline bool finish_applies(const RoundState &st) {
    if (st.terms.n_terms == 1 ? !finish_shape_ok(st.k, st.D) : !fast_degree(st.D)) return false;
    const uint64_t after = st.pending_fold ? st.vars_left - 1 : st.vars_left;
    return after >= 1 && after <= (uint64_t)kFinishVars;
}

static int32_t prover_step(RoundState &st, bool *finished_in_kernel) {
    if (st.init.valid && (finish_pipe_applies(st) || st.pipe_active || finish_applies(st))) ZKCHK(flush_pending_sponge(st));
    if (finish_pipe_applies(st)) {                                       // (covers the pipelined state as well)
        *finished_in_kernel = true;
        return finish_pipe_enqueue(st);
    }
    if (st.pipe_active) return pipe_step(st);
    if (finish_applies(st)) {
        *finished_in_kernel = true;
        return finish_enqueue(st);                                       // all remaining rounds in one launch
    }
    // the table of this round has m_s variables; if the NEXT round belongs to the pipeline, this round's tail is merged
    // into the launch that prepares it
    const uint64_t m_s = st.pending_fold ? st.vars_left - 1 : st.vars_left;
    DeferredTail dt = {0, false, false, nullptr};
    const bool enter = fast_degree(st.D) && pipe_wants_next(st, m_s);
    ZKCHK(round_enqueue(st, nullptr, enter ? &dt : nullptr));
    // dt.blocks == 0: round_enqueue's contract for "the tail was launched after all" (a sums path that cannot defer it) --
    // the round is closed and its challenge published, so the next round simply takes the classic path
    if (enter && dt.blocks) ZKCHK(pipe_enter(st, dt));
    ++st.round;
    return ZK_OK;
}

// host byte sponge (table + claimed sum absorbed) -> device word sponge.  The 208-byte state travels as a kernel argument:
// no staging buffer, no copy engine, no host synchronisation in front of the first round.
// d_epart: the E-partial block whose two counters the same launch clears (every proof starts with this launch)
static int32_t sponge_to_device(zk_ctx *c, const Sponge &host, WordSponge *d_sponge, uint64_t *d_epart) {
    WordSponge w;
    if (!w.from_byte_sponge(host)) return ZK_ERR_BAD_ARG;
    uint64_t *zero2 = d_epart ? epart_counters(d_epart) : nullptr;
    hipStream_t st = c->stream;
    auto single = [=]() {
        k_store_sponge<<<1, 64, 0, st>>>(w, d_sponge, zero2);
        return hipGetLastError();
    };
    if (batch_record(BK_STORE_SPONGE, 0, 1, 64, 0, 0, 0, 0, 0, SpongeSlot{w, d_sponge, zero2}, single)) return ZK_OK;
    HIPCHK(single());
    return ZK_OK;
}

static bool has_duplicate_handles(zk_mle *const *f, uint64_t k) {
    for (uint64_t i = 0; i < k; ++i)
        for (uint64_t j = i + 1; j < k; ++j)
            if (f[i] == f[j]) return true;
    return false;
}
// The prover for sum_i prod_{f in term i} T_f (one term = the reference's ProductPoly).  out_final (optional): the k
// factors evaluated at the challenge point.
// d_keep_ch / d_keep_final (optional, DEVICE buffers of n / k elements): device-resident copies of the challenges and of the
// factor values, for a caller that chains further device work on them without a host round trip (the GKR driver).
static int32_t prove_core(zk_ctx *c, zk_mle *const *f, uint64_t k, const TermSpec &ts, uint32_t D, const uint64_t sum[4],
                          int32_t absorb_table, int32_t consume, uint64_t *out_rp, uint64_t *out_ch, uint64_t *out_final,
                          uint64_t *d_keep_ch = nullptr, uint64_t *d_keep_final = nullptr, const Sponge *init = nullptr,
                          const DeviceChain *chain = nullptr) {
    if (!sum && !chain) return ZK_ERR_BAD_ARG;
    ZKCHK(product_args(c, (const zk_mle *const *)f, k));
    if (chain) {   // device-resident transcript: enqueue the rounds and return (no host data, no synchronisation)
        const uint64_t n = f[0]->n_vars;
        if (n == 0 || D >= kMaxSums) return ZK_ERR_UNSUPPORTED;
        if (has_duplicate_handles(f, k)) consume = 0;
        RoundState st;
        ZKCHK(round_state_init(st, c, f, k, D, consume != 0, n, chain));
        st.terms = ts;
        st.d_final = st.ps.d_final;
        int32_t rc = ZK_OK;
        bool fin = false;
        while (st.round < n && rc == ZK_OK) rc = prover_step(st, &fin);
        if (rc == ZK_OK && !fin) {
            FactorPtrs fp = {};
            for (uint64_t i = 0; i < k; ++i) fp.in[i] = st.cur[i];
            k_final_evals<<<1, 64, 0, c->stream>>>(fp, (uint32_t)k, chal_prev(st), st.d_final, c->fi->P);
            if (hipGetLastError() != hipSuccess) rc = ZK_ERR_HIP;
        }
        round_state_release(st);   // stream-ordered reuse of the scratch
        return rc;
    }
    if (f[0]->n_vars && (!out_rp || !out_ch)) return ZK_ERR_BAD_ARG;
    if (D >= kMaxSums) return ZK_ERR_UNSUPPORTED;
    const FieldParams &P = c->fi->P;
    const uint64_t n = f[0]->n_vars;
    Sponge sp;
    if (init) sp = *init;                                                // a driver chaining its sumchecks onto its own transcript (GKR)
    else sp.init();                                                      // Transcript::new (prover.rs:16,28)
    if (absorb_table) ZKCHK(absorb_tables(c, sp, f, k));                 // prover.rs:17
    absorb_elements(sp, sum, 1, P);                                      // prover.rs:42
    if (n == 0) {                                                        // no rounds (prover.rs:44)
        if (out_final)
            for (uint64_t i = 0; i < k; ++i) ZKCHK(zk_mle_download(c, f[i], out_final + 4 * i));
        return ZK_OK;
    }
    // the reference takes the polynomial by value and clones per fold, so a table may appear twice (A * A): in-place folds
    // would then fold the shared buffer once per listing -- such a product is proved out of place (own scratch per factor)
    if (has_duplicate_handles(f, k)) consume = 0;
    static const bool host_dbg = env_flag("ZK_HOST_DEBUG");
    const auto t_enter = std::chrono::steady_clock::now();
    RoundState st;
    ZKCHK(round_state_init(st, c, f, k, D, consume != 0, n));
    st.terms = ts;
    int32_t rc = ZK_OK;
    if (out_final) st.d_final = st.ps.d_final;
    // ZK_SPONGE_IN_TAIL=0: the initial sponge goes to the device with a launch of its own in front of round 0 (until round 6; A/B)
    static const bool sponge_in_tail = env_u64("ZK_SPONGE_IN_TAIL", 1, 0, 1) != 0;
    if (rc == ZK_OK) {
        if (sponge_in_tail && !g_batch) {
            if (!st.init.w.from_byte_sponge(sp)) rc = ZK_ERR_BAD_ARG;
            st.init.dst = st.ps.d_sponge;
            st.init.zero2 = st.ps.d_epart ? epart_counters(st.ps.d_epart) : nullptr;
            st.init.valid = rc == ZK_OK;
        } else {
            rc = sponge_to_device(c, sp, st.ps.d_sponge, st.ps.d_epart);
        }
    }
    // one copy of [round polys | challenges | finals] into pinned memory behind a completion word: by the pipelined finisher itself
    // when it is the call's last launch (ZK_PUBLISH_IN_FINISHER=0: always by k_publish_host), else by k_publish_host below
    uint8_t *stage = nullptr;
    const size_t block = st.ps.rp_bytes + st.ps.ch_bytes + kMaxFactors * 32;
    if (rc == ZK_OK) rc = results_staging(c, block, &stage);
    uint32_t seq = 0;
    static const bool publish_in_finisher = env_u64("ZK_PUBLISH_IN_FINISHER", 1, 0, 1) != 0;
    if (rc == ZK_OK) {
        seq = next_flag_seq(c);
        if (publish_in_finisher && !d_keep_ch && !d_keep_final)
            st.pub = FinishPublish{st.ps.d_rp, reinterpret_cast<uint64_t *>(stage), (uint32_t)(block / 8), c->h_flag, seq};
    }
    bool finished_in_kernel = false;
    while (st.round < n && rc == ZK_OK) rc = prover_step(st, &finished_in_kernel);   // prover.rs:44-68, all on device
    // prover.rs:64 after the LAST round folds to a 0-variable polynomial the reference drops: computed only on request.
    if (rc == ZK_OK && out_final) {
        if (!finished_in_kernel) {
            FactorPtrs fp = {};
            for (uint64_t i = 0; i < k; ++i) fp.in[i] = st.cur[i];
            k_final_evals<<<1, 64, 0, c->stream>>>(fp, (uint32_t)k, chal_prev(st), st.d_final, P);
            if (hipGetLastError() != hipSuccess) rc = ZK_ERR_HIP;
        }
    }
    if (rc == ZK_OK && d_keep_ch && hipMemcpyAsync(d_keep_ch, st.ps.d_ch, (size_t)n * 32, hipMemcpyDeviceToDevice, c->stream) != hipSuccess)
        rc = ZK_ERR_HIP;
    if (rc == ZK_OK && d_keep_final && out_final &&
        hipMemcpyAsync(d_keep_final, st.ps.d_final, (size_t)k * 32, hipMemcpyDeviceToDevice, c->stream) != hipSuccess)
        rc = ZK_ERR_HIP;
    if (rc == ZK_OK && !st.published) {   // the proof block goes to pinned memory by a kernel that also stores the completion word
        k_publish_host<<<1, 64, 0, c->stream>>>(st.ps.d_rp, reinterpret_cast<uint64_t *>(stage), (uint32_t)(block / 8), c->h_flag, seq);
        if (hipGetLastError() != hipSuccess) rc = ZK_ERR_HIP;
    }
    const auto t_enq = std::chrono::steady_clock::now();
    if (rc == ZK_OK) rc = host_flag_wait(c, seq);
    else (void)stream_wait(c->stream);
    if (host_dbg) {
        const auto t_done = std::chrono::steady_clock::now();
        fprintf(stderr, "[host] n=%llu: enqueue %.1f us (everything up to the last launch), wait %.1f us\n", (unsigned long long)n,
                std::chrono::duration<double, std::micro>(t_enq - t_enter).count(), std::chrono::duration<double, std::micro>(t_done - t_enq).count());
    }
    if (c->d_dbg) pipe_dbg_dump(c);
    if (rc == ZK_OK) {
        memcpy(out_rp, stage, (size_t)n * (D + 1) * 32);
        memcpy(out_ch, stage + st.ps.rp_bytes, (size_t)n * 32);
        if (out_final) memcpy(out_final, stage + st.ps.rp_bytes + st.ps.ch_bytes, (size_t)k * 32);
    }
    round_state_release(st);
    return rc;
}

extern "C" int32_t zk_sumcheck_prove(zk_ctx *c, zk_mle *const *f, uint64_t k, uint32_t D, const uint64_t sum[4],
                                     int32_t absorb_table, int32_t consume, uint64_t *out_rp, uint64_t *out_ch) {
    if (k == 0 || k > (uint64_t)kMaxFactors) return product_args(c, (const zk_mle *const *)f, k);
    return prove_core(c, f, k, single_term((int)k), D, sum, absorb_table, consume, out_rp, out_ch, nullptr);
}

// prove_partial for a SUM of products (what a GKR layer needs): factors flat, term_k[i] factors per term.
// Tables must not be shared between terms (each is folded by the term that lists it).
extern "C" int32_t zk_sumcheck_prove_terms(zk_ctx *c, zk_mle *const *f, const uint64_t *term_k, uint64_t n_terms, uint32_t D,
                                           const uint64_t sum[4], int32_t consume, uint64_t *out_rp, uint64_t *out_ch,
                                           uint64_t *out_final) {
    if (!c || !f || !term_k) return ZK_ERR_BAD_ARG;
    if (n_terms == 0) return ZK_ERR_EMPTY_PRODUCT;
    if (n_terms > (uint64_t)kMaxTerms) return ZK_ERR_UNSUPPORTED;
    TermSpec ts = {};
    ts.n_terms = (int)n_terms;
    uint64_t k = 0;
    for (uint64_t i = 0; i < n_terms; ++i) {
        if (term_k[i] == 0) return ZK_ERR_EMPTY_PRODUCT;
        if (term_k[i] > (uint64_t)kMaxFactors || term_k[i] > D) return ZK_ERR_UNSUPPORTED;   // a term of k factors has degree k
        ts.term_k[i] = (int)term_k[i];
        k += term_k[i];
    }
    if (k > (uint64_t)kMaxFactors) return ZK_ERR_UNSUPPORTED;
    for (uint64_t i = 0; i < k; ++i)
        for (uint64_t j = i + 1; j < k; ++j)
            if (f[i] && f[i] == f[j]) return ZK_ERR_BAD_ARG;
    if (n_terms > 1 && !fast_degree(D)) return ZK_ERR_UNSUPPORTED;
    return prove_core(c, f, k, ts, D, sum, 0, consume, out_rp, out_ch, out_final);
}

extern "C" int32_t zk_sumcheck_prove_host(zk_ctx *c, const uint64_t *const *tables, uint64_t k, uint64_t n_vars, uint32_t D,
                                          const uint64_t sum[4], int32_t absorb_table, uint64_t *out_rp, uint64_t *out_ch) {
    if (!c) return ZK_ERR_BAD_ARG;
    if (k == 0) return ZK_ERR_EMPTY_PRODUCT;
    if (!tables || k > (uint64_t)kMaxFactors) return tables ? ZK_ERR_UNSUPPORTED : ZK_ERR_BAD_ARG;
    zk_mle *h[kMaxFactors] = {};
    int32_t rc = ZK_OK;
    for (uint64_t i = 0; i < k && rc == ZK_OK; ++i) rc = zk_mle_upload(c, n_vars, tables[i], 1ull << n_vars, &h[i]);
    if (rc == ZK_OK) rc = zk_sumcheck_prove(c, h, k, D, sum, absorb_table, 1, out_rp, out_ch);
    for (uint64_t i = 0; i < k; ++i) (void)zk_mle_free(c, h[i]);
    return rc;
}

// ------------------------------------------------------------------------------------------------------------
// B independent proofs in ONE launch sequence (config 4 as SURVEY 8d words it: "8 independent layers x (k = 3, n = 20)")
// ------------------------------------------------------------------------------------------------------------
// n_proofs separate prove_partial calls (prover.rs:24-30: one per ProductPoly; nothing in the reference orders independent calls) of
// one shape (k, D) and one size, each with its own transcript, proved side by side: the host schedule of a proof runs once per proof
// with a recorder installed (launch.hpp), and launch i of all proofs is issued as one launch of the kernel's batched twin -- the B
// transcript steps of a round run next to each other instead of one after the other, and the number of launches does not grow with B.
// Every proof is bit-identical to the one zk_sumcheck_prove returns for the same inputs (same kernels' bodies, same arithmetic).
static thread_local uint64_t g_batch_merged = 0, g_batch_replayed = 0;
static int32_t batch_flush(BatchRecorder &r) {
    int32_t rc = ZK_OK;
    const size_t len = r.recs[0].size();
    bool same_len = true;
    for (int b = 1; b < r.n; ++b) same_len = same_len && r.recs[b].size() == len;
    auto replay = [&](int b, size_t i) {
        if (rc == ZK_OK && r.recs[b][i].single() != hipSuccess) {
            g_hip_err = "batched prover: a replayed launch failed";
            rc = ZK_ERR_HIP;
        }
        ++r.replayed;
    };
    if (!same_len) {   // cannot happen for proofs of one shape and size; proofs are independent, so proof-by-proof order is valid too
        for (int b = 0; b < r.n; ++b)
            for (size_t i = 0; i < r.recs[b].size(); ++i) replay(b, i);
    } else {
        for (size_t i = 0; i < len && rc == ZK_OK; ++i) {
            const BatchRecord &r0 = r.recs[0][i];
            bool same = r0.kernel != BK_OTHER;
            for (int b = 1; b < r.n && same; ++b) {
                const BatchRecord &x = r.recs[b][i];
                same = x.kernel == r0.kernel && x.shape == r0.shape && x.grid == r0.grid && x.block == r0.block && x.lds == r0.lds &&
                       memcmp(x.s, r0.s, sizeof r0.s) == 0;
            }
            int lrc = kLaunchUnsupported;
            if (same) {
                const dim3 grid(r0.grid, (uint32_t)r.n);
                if (r0.kernel == BK_STORE_SPONGE) {
                    BatchOf<SpongeSlot> slots;
                    batch_gather(r, i, slots);
                    k_store_sponge_b<<<grid, r0.block, 0, r.stream>>>(slots);
                    lrc = hipGetLastError() == hipSuccess ? kLaunchOk : kLaunchHipError;
                } else if (r0.kernel == BK_TAIL) {
                    BatchOf<TailSlot> slots;
                    batch_gather(r, i, slots);
                    k_round_tail_b<<<grid, r0.block, 0, r.stream>>>(slots, (uint32_t)r0.s[0], (uint32_t)r0.s[1], *r.P);
                    lrc = hipGetLastError() == hipSuccess ? kLaunchOk : kLaunchHipError;
                } else if (r0.kernel == BK_PIPE || r0.kernel == BK_FINISH_PIPE) {
                    lrc = batch_launch_pipe(r, i);
                } else {
                    lrc = batch_launch_rounds(r, i);
                }
            }
            if (lrc == kLaunchOk) {
                ++r.merged;
            } else if (lrc == kLaunchHipError) {
                g_hip_err = "batched prover: a merged launch failed";
                rc = ZK_ERR_HIP;
            } else {
                static const bool dbg = env_flag("ZK_BATCH_DEBUG");
                if (dbg)
                    fprintf(stderr, "[batch] step %zu replayed: kernel %d shape 0x%x grid %u block %u lds %zu s = %llu %llu %llu %llu (%s)\n", i, r0.kernel,
                            r0.shape, r0.grid, r0.block, r0.lds, (unsigned long long)r0.s[0], (unsigned long long)r0.s[1], (unsigned long long)r0.s[2],
                            (unsigned long long)r0.s[3], same ? "no batched twin" : "shared arguments differ between the proofs");
                for (int b = 0; b < r.n; ++b) replay(b, i);
            }
        }
    }
    for (int b = 0; b < r.n; ++b) r.recs[b].clear();
    return rc;
}
// up to kMaxBatch proofs; f = B * k handles (proof-major), sums = B * 4 words, outputs proof-major
static int32_t prove_batch_group(zk_ctx *c, int B, zk_mle *const *f, uint64_t k, uint32_t D, const uint64_t *sums, int32_t consume, uint64_t *out_rp,
                                 uint64_t *out_ch) {
    const uint64_t n = f[0]->n_vars;
    const size_t rp_words = (size_t)n * (D + 1) * 4, ch_words = (size_t)n * 4;
    if (B == 1 || n == 0 || !fast_degree(D)) {   // nothing to merge, or a degree whose rounds go through context-wide scratch: one by one
        for (int b = 0; b < B; ++b)
            ZKCHK(prove_core(c, f + (size_t)b * k, k, single_term((int)k), D, sums + 4 * b, 0, consume, out_rp + b * rp_words, out_ch + b * ch_words, nullptr));
        return ZK_OK;
    }
    const FieldParams &P = c->fi->P;
    BatchRecorder rec;
    rec.n = B;
    rec.stream = c->stream;
    rec.P = &P;
    struct Guard {   // the recorder is installed for this thread only while the schedule runs
        explicit Guard(BatchRecorder *r) { g_batch = r; }
        ~Guard() { g_batch = nullptr; }
    };
    std::vector<RoundState> st((size_t)B);
    int inited = 0;
    int32_t rc = ZK_OK;
    uint32_t seq[kMaxBatch] = {};
    uint8_t *stage = nullptr;
    size_t block = 0;
    static const bool publish_in_finisher = env_u64("ZK_PUBLISH_IN_FINISHER", 1, 0, 1) != 0;
    {
        Guard guard(&rec);
        for (int b = 0; b < B && rc == ZK_OK; ++b) {
            rec.cur = b;
            rc = round_state_init(st[(size_t)b], c, f + (size_t)b * k, k, D, consume != 0, n);
            if (rc == ZK_OK) ++inited;
        }
        if (rc == ZK_OK) {
            block = (st[0].ps.rp_bytes + st[0].ps.ch_bytes + kMaxFactors * 32 + 63) & ~(size_t)63;
            rc = results_staging(c, block * (size_t)B, &stage);
        }
        for (int b = 0; b < B && rc == ZK_OK; ++b) {
            rec.cur = b;
            RoundState &s = st[(size_t)b];
            s.terms = single_term((int)k);
            Sponge sp;
            sp.init();                                        // Transcript::new (prover.rs:28)
            absorb_elements(sp, sums + 4 * b, 1, P);          // prover.rs:42
            rc = sponge_to_device(c, sp, s.ps.d_sponge, s.ps.d_epart);
            seq[b] = next_flag_seq(c);
            if (publish_in_finisher)
                s.pub = FinishPublish{s.ps.d_rp, reinterpret_cast<uint64_t *>(stage + block * (size_t)b), (uint32_t)(block / 8), c->h_flag + 16 * b, seq[b]};
        }
        if (rc == ZK_OK) rc = batch_flush(rec);
        while (rc == ZK_OK && st[0].round < n) {              // prover.rs:44-68: one step of every proof, then the merged launches
            for (int b = 0; b < B && rc == ZK_OK; ++b) {
                rec.cur = b;
                bool fin = false;
                rc = prover_step(st[(size_t)b], &fin);
            }
            if (rc == ZK_OK) rc = batch_flush(rec);
        }
        for (int b = 0; b < B && rc == ZK_OK; ++b) {
            rec.cur = b;
            RoundState &s = st[(size_t)b];
            if (s.round != n) rc = ZK_ERR_BAD_ARG;            // (the proofs must have advanced in lockstep)
            if (rc == ZK_OK && !s.published) {
                const uint64_t *src = s.ps.d_rp;
                uint64_t *dst = reinterpret_cast<uint64_t *>(stage + block * (size_t)b);
                const uint32_t words = (uint32_t)((s.ps.rp_bytes + s.ps.ch_bytes + kMaxFactors * 32) / 8);
                volatile uint32_t *flag = c->h_flag + 16 * b;
                const uint32_t sq = seq[b];
                hipStream_t stream = c->stream;
                (void)batch_record_other([=]() {
                    k_publish_host<<<1, 64, 0, stream>>>(src, dst, words, flag, sq);
                    return hipGetLastError();
                });
            }
        }
        if (rc == ZK_OK) rc = batch_flush(rec);
    }
    g_batch_merged = rec.merged;
    g_batch_replayed = rec.replayed;
    for (int b = 0; b < B; ++b) {
        if (rc == ZK_OK) rc = host_flag_wait(c, seq[b], (uint32_t)b);
    }
    if (rc != ZK_OK) (void)stream_wait(c->stream);
    if (rc == ZK_OK) {
        for (int b = 0; b < B; ++b) {
            const uint8_t *src = stage + block * (size_t)b;
            memcpy(out_rp + b * rp_words, src, rp_words * 8);
            memcpy(out_ch + b * ch_words, src + st[(size_t)b].ps.rp_bytes, ch_words * 8);
        }
    }
    for (int b = 0; b < inited; ++b) round_state_release(st[(size_t)b]);
    return rc;
}
static int32_t prove_batch_impl(zk_ctx *c, uint64_t n_proofs, zk_mle *const *f, uint64_t k, uint32_t D, const uint64_t *sums, int32_t consume,
                                uint64_t *out_rp, uint64_t *out_ch);
// the recorder keeps its launches in host containers: an allocation failure is a status, never an exception across the C ABI
extern "C" int32_t zk_sumcheck_prove_batch(zk_ctx *c, uint64_t n_proofs, zk_mle *const *f, uint64_t k, uint32_t D, const uint64_t *sums, int32_t consume,
                                           uint64_t *out_rp, uint64_t *out_ch) {
    try {
        return prove_batch_impl(c, n_proofs, f, k, D, sums, consume, out_rp, out_ch);
    } catch (const std::bad_alloc &) {
        g_batch = nullptr;
        if (c) (void)stream_wait(c->stream);
        return ZK_ERR_ALLOC;
    } catch (...) {
        g_batch = nullptr;
        if (c) (void)stream_wait(c->stream);
        return ZK_ERR_BAD_ARG;
    }
}
static int32_t prove_batch_impl(zk_ctx *c, uint64_t n_proofs, zk_mle *const *f, uint64_t k, uint32_t D, const uint64_t *sums, int32_t consume,
                                uint64_t *out_rp, uint64_t *out_ch) {
    if (!c || !f || !sums) return ZK_ERR_BAD_ARG;
    if (n_proofs == 0) return ZK_OK;
    if (k == 0) return ZK_ERR_EMPTY_PRODUCT;
    if (k > (uint64_t)kMaxFactors || D >= kMaxSums) return ZK_ERR_UNSUPPORTED;
    for (uint64_t p = 0; p < n_proofs; ++p) {
        ZKCHK(product_args(c, (const zk_mle *const *)(f + p * k), k));          // each proof: ProductPoly::new's checks (product_poly.rs:14-32)
        if (f[p * k]->n_vars != f[0]->n_vars) return ZK_ERR_ARITY_MISMATCH;      // a batch is one size
    }
    const uint64_t n = f[0]->n_vars;
    if (n && (!out_rp || !out_ch)) return ZK_ERR_BAD_ARG;
    // in-place folds need every table to be its proof's own: a handle listed twice (inside a proof, or by two proofs) -> out of place
    if (consume) {
        std::set<const zk_mle *> seen;
        for (uint64_t i = 0; i < n_proofs * k && consume; ++i)
            if (!seen.insert(f[i]).second) consume = 0;
    }
    g_batch_merged = g_batch_replayed = 0;
    uint64_t merged = 0, replayed = 0;
    const size_t rp_words = (size_t)n * (D + 1) * 4, ch_words = (size_t)n * 4;
    for (uint64_t p0 = 0; p0 < n_proofs; p0 += kMaxBatch) {
        const int B = (int)std::min<uint64_t>(kMaxBatch, n_proofs - p0);
        ZKCHK(prove_batch_group(c, B, f + p0 * k, k, D, sums + 4 * p0, consume, out_rp ? out_rp + p0 * rp_words : nullptr,
                                out_ch ? out_ch + p0 * ch_words : nullptr));
        merged += g_batch_merged;
        replayed += g_batch_replayed;
    }
    g_batch_merged = merged;
    g_batch_replayed = replayed;
    return ZK_OK;
}
// what the last zk_sumcheck_prove_batch of this thread did: launches issued for all proofs at once / replayed proof by proof
extern "C" int32_t zk_batch_last_stats(uint64_t *out_merged, uint64_t *out_replayed) {
    if (!out_merged || !out_replayed) return ZK_ERR_BAD_ARG;
    *out_merged = g_batch_merged;
    *out_replayed = g_batch_replayed;
    return ZK_OK;
}

// ------------------------------------------------------------------------------------------------------------
// sharded prover: the same loop with one exchange point per round
// ------------------------------------------------------------------------------------------------------------
struct zk_shard_prover {
    RoundState st;
    uint32_t world;
    uint64_t local_rounds, total_rounds;
    uint64_t *d_lanes;    // (D+1)*8 u64 lanes
    TailDerive lanes_dv;  // what round_finish derives from the all-reduced lanes (set by round_begin)
    uint64_t *d_tail;     // k * 2^tail_s elements: this rank's shard tables at the moment of the gather
    size_t tail_bytes;
    uint32_t tail_s;      // variables left in the local tables when gathered
    bool tail_done;
    uint64_t *d_gathered; // zk_shard_prover_run: the all-gathered tails [world][k][2^tail_s]
    size_t gathered_bytes;
};
extern "C" int32_t zk_shard_prover_create(zk_ctx *c, zk_mle *const *f, uint64_t k, uint32_t D, const uint64_t sum[4],
                                          uint32_t world, zk_shard_prover **out) {
    if (!out || !sum) return ZK_ERR_BAD_ARG;
    ZKCHK(product_args(c, (const zk_mle *const *)f, k));
    if (D >= kMaxSums) return ZK_ERR_UNSUPPORTED;
    if (world == 0 || (world & (world - 1)) || world > 65536) return ZK_ERR_BAD_ARG;
    uint32_t lw = 0;
    while ((1u << lw) < world) ++lw;
    zk_shard_prover *sp = new (std::nothrow) zk_shard_prover();
    if (!sp) return ZK_ERR_ALLOC;
    sp->world = world;
    sp->local_rounds = f[0]->n_vars;
    sp->total_rounds = f[0]->n_vars + lw;
    sp->d_lanes = sp->d_tail = nullptr;
    sp->tail_bytes = 0;
    sp->tail_s = 0;
    sp->tail_done = false;
    sp->d_gathered = nullptr;
    sp->gathered_bytes = 0;
    // the shard tables are consumed (folded in place) unless one is listed twice (see prove_core)
    int32_t rc = round_state_init(sp->st, c, f, k, D, /*consume=*/!has_duplicate_handles(f, k), sp->total_rounds);
    if (rc == ZK_OK) rc = pool_alloc(c, (size_t)kMaxSums * 8 * sizeof(uint64_t), (void **)&sp->d_lanes);
    if (rc == ZK_OK) {
        Sponge host;
        host.init();
        absorb_elements(host, sum, 1, c->fi->P);   // prover.rs:42 -- the GLOBAL claimed sum, identical on every rank
        rc = sponge_to_device(c, host, sp->st.ps.d_sponge, sp->st.ps.d_epart);
    }
    if (rc != ZK_OK) {
        (void)zk_shard_prover_destroy(sp);
        return rc;
    }
    *out = sp;
    return ZK_OK;
}
extern "C" int32_t zk_shard_prover_destroy(zk_shard_prover *sp) {
    if (!sp) return ZK_OK;
    zk_ctx *c = sp->st.c;
    (void)hipSetDevice(c->device);
    round_state_release(sp->st);
    pool_free(c, sp->d_lanes, (size_t)kMaxSums * 8 * sizeof(uint64_t));
    pool_free(c, sp->d_tail, sp->tail_bytes);
    pool_free(c, sp->d_gathered, sp->gathered_bytes);
    delete sp;
    return ZK_OK;
}
extern "C" int32_t zk_shard_prover_lanes_ptr(zk_shard_prover *sp, void **out_ptr, uint64_t *out_n) {
    if (!sp || !out_ptr || !out_n) return ZK_ERR_BAD_ARG;
    *out_ptr = sp->d_lanes;
    *out_n = (uint64_t)(sp->st.D + 1) * 8;
    return ZK_OK;
}
extern "C" int32_t zk_shard_prover_rounds(zk_shard_prover *sp, uint64_t *out_local, uint64_t *out_total, uint64_t *out_done) {
    if (!sp) return ZK_ERR_BAD_ARG;
    if (out_local) *out_local = sp->local_rounds;
    if (out_total) *out_total = sp->total_rounds;
    if (out_done) *out_done = sp->st.round;
    return ZK_OK;
}
// local part of a round: (fold +) sums -> digit lanes.  Asynchronous.
extern "C" int32_t zk_shard_prover_round_begin(zk_shard_prover *sp) {
    if (!sp) return ZK_ERR_BAD_ARG;
    RoundState &st = sp->st;
    if (st.round >= sp->local_rounds) return ZK_ERR_BAD_ARG;
    ZKCHK(use_device(st.c));
    sp->lanes_dv = TailDerive{};
    return round_enqueue(st, sp->d_lanes, nullptr, &sp->lanes_dv);
}
// after the caller's all-reduce of the lanes: reduce mod p, absorb, squeeze.  Asynchronous.
extern "C" int32_t zk_shard_prover_round_finish(zk_shard_prover *sp) {
    if (!sp) return ZK_ERR_BAD_ARG;
    RoundState &st = sp->st;
    zk_ctx *c = st.c;
    if (st.round >= sp->local_rounds || !st.pending_fold) return ZK_ERR_BAD_ARG;
    ZKCHK(use_device(c));
    {
        uint32_t lw = 0;
        while ((1u << lw) < sp->world) ++lw;
        sp->lanes_dv.log_world = lw;
    }
    k_lanes_transcript<<<1, 128, 0, c->stream>>>(sp->d_lanes, st.D + 1, st.ps.d_sponge, st.ps.d_rp + st.round * (st.D + 1) * 4,
                                                st.ps.d_ch + st.round * 4, chal_cur(st), c->fi->P, sp->lanes_dv);
    HIPCHK(hipGetLastError());
    ++st.round;
    return ZK_OK;
}
// Stop exchanging per round: apply the pending challenge and expose this rank's k shard tables (2^s elements each,
// s = local variables still unfolded; s = 0 after the last local round) as one device buffer [k][2^s] for the all-gather.
// May be called after any number of local rounds: once shards are tiny a collective per round costs more than finishing
// redundantly on every rank.
extern "C" int32_t zk_shard_prover_tail_ptr(zk_shard_prover *sp, void **out_ptr, uint64_t *out_elems) {
    if (!sp || !out_ptr || !out_elems) return ZK_ERR_BAD_ARG;
    RoundState &st = sp->st;
    zk_ctx *c = st.c;
    if (st.round > sp->local_rounds) return ZK_ERR_BAD_ARG;
    ZKCHK(use_device(c));
    if (!sp->tail_done) {
        const uint64_t s = st.pending_fold ? st.vars_left - 1 : st.vars_left;
        sp->tail_s = (uint32_t)s;
        sp->tail_bytes = (size_t)st.k * ((size_t)32 << s);
        ZKCHK(pool_alloc(c, sp->tail_bytes, (void **)&sp->d_tail));
        for (uint64_t i = 0; i < st.k; ++i) {
            uint64_t *dst = sp->d_tail + ((uint64_t)i << s) * 4;
            if (st.pending_fold) {
                const uint64_t pairs = 1ull << s;
                k_fold_dev<<<grid_for(pairs), kBlock, 0, c->stream>>>(st.cur[i], dst, pairs, (uint32_t)(s + 1), c->fi->P, chal_prev(st));
                HIPCHK(hipGetLastError());
            } else {
                HIPCHK(hipMemcpyAsync(dst, st.cur[i], (size_t)32 << s, hipMemcpyDeviceToDevice, c->stream));
            }
        }
        st.pending_fold = false;
        st.vars_left = s;
        sp->tail_done = true;
    }
    *out_ptr = sp->d_tail;
    *out_elems = (uint64_t)st.k << sp->tail_s;
    return ZK_OK;
}
// gathered: device array [world][k][2^s] elements (rank-major, the all-gather of every rank's tail buffer), identical
// on every rank.  Builds the k tables of s + log2(world) variables -- global index = local * world + rank, i.e. the
// reference's own index order -- and runs ALL remaining rounds locally (no further collective).  Asynchronous.
extern "C" int32_t zk_shard_prover_tail_rounds(zk_shard_prover *sp, const void *gathered) {
    if (!sp || !gathered) return ZK_ERR_BAD_ARG;
    RoundState &st = sp->st;
    zk_ctx *c = st.c;
    if (!sp->tail_done) return ZK_ERR_BAD_ARG;
    ZKCHK(use_device(c));
    uint32_t lw = 0;
    while ((1u << lw) < sp->world) ++lw;
    const uint64_t vars = (uint64_t)sp->tail_s + lw;
    if (st.round + vars != sp->total_rounds) return ZK_ERR_BAD_ARG;
    if (vars == 0) return ZK_OK;
    FactorPtrs fp = {};
    for (uint64_t i = 0; i < st.k; ++i) {
        if (st.scratch[i]) pool_free(c, st.scratch[i], st.scratch_bytes[i]);
        st.scratch_bytes[i] = (size_t)32 << vars;
        st.scratch[i] = nullptr;
        ZKCHK(pool_alloc(c, st.scratch_bytes[i], (void **)&st.scratch[i]));
        fp.out[i] = st.scratch[i];
        st.cur[i] = st.scratch[i];
    }
    const uint64_t items = ((uint64_t)st.k << sp->tail_s) * sp->world;
    k_gather_to_tables<<<grid_for(items), kBlock, 0, c->stream>>>((const uint64_t *)gathered, fp, (uint32_t)st.k, sp->world, sp->tail_s);
    HIPCHK(hipGetLastError());
    st.pending_fold = false;
    st.first_out_of_place = false;
    st.vars_left = vars;
    int32_t rc = ZK_OK;
    bool fin = false;
    while (st.round < sp->total_rounds && rc == ZK_OK) rc = prover_step(st, &fin);
    return rc;
}
// download what has been proven so far (synchronises): total_rounds*(D+1) and total_rounds elements
extern "C" int32_t zk_shard_prover_results(zk_shard_prover *sp, uint64_t *out_rp, uint64_t *out_ch) {
    if (!sp || !out_rp || !out_ch) return ZK_ERR_BAD_ARG;
    RoundState &st = sp->st;
    zk_ctx *c = st.c;
    ZKCHK(use_device(c));
    if (st.round) {
        HIPCHK(hipMemcpyAsync(out_rp, st.ps.d_rp, (size_t)st.round * (st.D + 1) * 32, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(out_ch, st.ps.d_ch, (size_t)st.round * 32, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return ZK_OK;
}
// plain device <-> host copies on the context's stream (synchronous): lets a host without a HIP binding (tests, the
// single-process rehearsal of the sharded prover) move the lanes / tail buffers
extern "C" int32_t zk_ctx_memcpy_dtoh(zk_ctx *c, void *dst_host, const void *src_dev, uint64_t bytes) {
    if (!c || !dst_host || !src_dev) return ZK_ERR_BAD_ARG;
    ZKCHK(use_device(c));
    HIPCHK(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return ZK_OK;
}
extern "C" int32_t zk_ctx_memcpy_htod(zk_ctx *c, void *dst_dev, const void *src_host, uint64_t bytes) {
    if (!c || !dst_dev || !src_host) return ZK_ERR_BAD_ARG;
    ZKCHK(use_device(c));
    HIPCHK(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return ZK_OK;
}
extern "C" int32_t zk_ctx_device_alloc(zk_ctx *c, uint64_t bytes, void **out) {
    if (!c || !out) return ZK_ERR_BAD_ARG;
    ZKCHK(use_device(c));
    return pool_alloc(c, bytes, out);
}
extern "C" int32_t zk_ctx_device_free(zk_ctx *c, void *ptr, uint64_t bytes) {
    if (!c) return ZK_ERR_BAD_ARG;
    pool_free(c, ptr, bytes);
    return ZK_OK;
}

// ------------------------------------------------------------------------------------------------------------
// SumcheckVerifier (host protocol logic; verifier.rs:15-78, univariate_poly.rs:29-80)
// ------------------------------------------------------------------------------------------------------------
// value at x of the unique polynomial of degree <= D through (i, ys[i]), i = 0..D: what
// UnivariatePolynomial::interpolate(ys).evaluate(x) returns (univariate_poly.rs:43-49, :29-40); exact in F_p.
// Lagrange basis on the nodes 0..D: w_i = 1 / prod_{j != i} (i - j), computed once per proof (one field inversion each)
// Barycentric weights of the nodes 0..D: one field inversion per node (~0.1 ms of host time for D = 2), so they are computed
// once per (modulus, D) and kept (a GKR verification interpolates in 2 x depth sumchecks).
static std::vector<Fe> interp_weights_compute(uint32_t D, const FieldParams &P);
static std::vector<Fe> interp_weights(uint32_t D, const FieldParams &P) {
    static std::mutex mu;
    static std::map<std::pair<std::array<uint32_t, 8>, uint32_t>, std::vector<Fe>> cache;
    std::array<uint32_t, 8> key;
    for (int i = 0; i < 8; ++i) key[i] = P.p[i];
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find({key, D});
    if (it == cache.end()) it = cache.emplace(std::make_pair(key, D), interp_weights_compute(D, P)).first;
    return it->second;
}
static std::vector<Fe> interp_weights_compute(uint32_t D, const FieldParams &P) {
    std::vector<Fe> w(D + 1);
    for (uint32_t i = 0; i <= D; ++i) {
        Fe den = fe_one(P);
        const Fe xi = fe_from_u32(i, P);
        for (uint32_t j = 0; j <= D; ++j)
            if (j != i) den = fe_mul(den, fe_sub(xi, fe_from_u32(j, P), P), P);
        w[i] = fe_inverse(den, P);
    }
    return w;
}
static Fe interp_eval(const std::vector<Fe> &ys, const std::vector<Fe> &w, const Fe &x, const FieldParams &P) {
    const size_t n = ys.size();
    Fe acc = fe_zero();
    for (size_t i = 0; i < n; ++i) {
        Fe num = fe_one(P);
        for (size_t j = 0; j < n; ++j)
            if (j != i) num = fe_mul(num, fe_sub(x, fe_from_u32((uint32_t)j, P), P), P);
        acc = fe_add(acc, fe_mul(ys[i], fe_mul(num, w[i], P), P), P);
    }
    return acc;
}
// Each round polynomial is interpolated at ITS OWN length (proof.round_polys is a Vec<Vec<F>>; verifier.rs:55-58 hands
// whatever the round carries to UnivariatePolynomial::interpolate): lens[r] evaluations for round r, stored back to back.
// 0 evaluations -> the zero polynomial (interpolate of no points, evaluate of no coefficients = 0), 1 -> a constant.
// lens == nullptr: every round carries `uniform_len` evaluations (the D + 1 of verify / verify_partial) -- no per-round array is
// built from a caller-supplied round count, so a hostile count cannot make the library allocate (or throw) before it is checked.
static int32_t verify_internal(const FieldParams &P, Sponge &sp, uint64_t n_rounds, const uint32_t *lens, uint32_t uniform_len,
                               const uint64_t sum[4], const uint64_t *rps, Fe &claimed, uint64_t *out_ch) {   // verifier.rs:44-78
    absorb_elements(sp, sum, 1, P);                                      // :50
    claimed = fe_from_u64limbs(sum);
    std::vector<Fe> w;
    uint32_t w_len = ~0u;
    const uint64_t *rp = rps;
    for (uint64_t r = 0; r < n_rounds; ++r) {
        const uint32_t len = lens ? lens[r] : uniform_len;
        if (len) absorb_elements(sp, rp, len, P);                        // :56 (no bytes for an empty round polynomial)
        std::vector<Fe> ys(len);
        for (uint32_t t = 0; t < len; ++t) ys[t] = fe_from_u64limbs(rp + 4 * t);
        if (len && len != w_len) w = interp_weights(len - 1, P), w_len = len;
        // :61-62 p(0), p(1): the interpolant through (i, ys[i]) takes exactly ys[0], ys[1] at the nodes 0 and 1
        const Fe p0 = len ? ys[0] : fe_zero(), p1 = len >= 2 ? ys[1] : (len ? ys[0] : fe_zero());
        if (!fe_eq(claimed, fe_add(p0, p1, P))) return ZK_ERR_VERIFY_SUM;                  // :64
        const Fe ch = squeeze_field_element(sp, P);                      // :69
        claimed = len ? interp_eval(ys, w, ch, P) : fe_zero();           // :70
        fe_to_u64limbs(ch, out_ch + 4 * r);
        rp += (size_t)len * 4;
    }
    return ZK_OK;
}
static int32_t verify_internal(const FieldParams &P, Sponge &sp, uint64_t n_rounds, uint32_t D, const uint64_t sum[4],
                               const uint64_t *rps, Fe &claimed, uint64_t *out_ch) {   // every round D + 1 evaluations
    return verify_internal(P, sp, n_rounds, nullptr, D + 1, sum, rps, claimed, out_ch);
}
static int32_t check_lens(uint64_t n_rounds, const uint32_t *lens) {
    for (uint64_t r = 0; r < n_rounds; ++r)
        if (lens[r] > kMaxSums) return ZK_ERR_BAD_ARG;
    return ZK_OK;
}
// shared bodies of the four verifier entry points: lens == nullptr means `uniform_len` evaluations in every round
static int32_t verify_partial_common(int32_t field, uint64_t n_rounds, const uint32_t *lens, uint32_t uniform_len,
                                     const uint64_t sum[4], const uint64_t *rps, uint64_t out_sum[4], uint64_t *out_ch) {
    const FieldInfo *fi = field_info(field);
    if (!fi) return ZK_ERR_BAD_FIELD;
    if (!sum || !out_sum || (n_rounds && (!rps || !out_ch))) return ZK_ERR_BAD_ARG;
    if (lens) ZKCHK(check_lens(n_rounds, lens));
    Sponge sp;
    sp.init();
    Fe claimed;
    ZKCHK(verify_internal(fi->P, sp, n_rounds, lens, uniform_len, sum, rps, claimed, out_ch));
    fe_to_u64limbs(claimed, out_sum);
    return ZK_OK;
}
static int32_t verify_common(zk_ctx *c, const zk_mle *const *f, uint64_t k, uint64_t n_rps, const uint32_t *lens,
                             uint32_t uniform_len, const uint64_t sum[4], const uint64_t *rps, int32_t *out_ok) {
    if (!sum || !out_ok || (n_rps && !rps)) return ZK_ERR_BAD_ARG;
    ZKCHK(product_args(c, f, k));
    if (n_rps != f[0]->n_vars) return ZK_ERR_VERIFY_ROUNDS;              // verifier.rs:17-19 -- BEFORE anything is sized by n_rps
    if (lens) ZKCHK(check_lens(n_rps, lens));
    Sponge sp;
    sp.init();
    ZKCHK(absorb_tables(c, sp, (zk_mle *const *)f, k));                  // :22
    uint64_t ch[4 * (kMaxVars + 1)];                                     // n_rps == n_vars <= kMaxVars
    Fe claimed;
    ZKCHK(verify_internal(c->fi->P, sp, n_rps, lens, uniform_len, sum, rps, claimed, ch));
    uint64_t ev[4];
    ZKCHK(zk_product_evaluate(c, f, k, ch, n_rps, ev));                  // :27-29
    *out_ok = fe_eq(fe_from_u64limbs(ev), claimed) ? 1 : 0;              // :31
    return ZK_OK;
}
extern "C" int32_t zk_sumcheck_verify_partial_lengths(int32_t field, uint64_t n_rounds, const uint32_t *lens,
                                                      const uint64_t sum[4], const uint64_t *rps, uint64_t out_sum[4],
                                                      uint64_t *out_ch) {
    if (n_rounds && !lens) return ZK_ERR_BAD_ARG;
    static const uint32_t none = 0;
    return verify_partial_common(field, n_rounds, lens ? lens : &none, 0, sum, rps, out_sum, out_ch);
}
extern "C" int32_t zk_sumcheck_verify_partial(int32_t field, uint64_t n_rounds, uint32_t D, const uint64_t sum[4],
                                              const uint64_t *rps, uint64_t out_sum[4], uint64_t *out_ch) {
    if (D >= kMaxSums) return ZK_ERR_BAD_ARG;
    return verify_partial_common(field, n_rounds, nullptr, D + 1, sum, rps, out_sum, out_ch);
}
extern "C" int32_t zk_sumcheck_verify_lengths(zk_ctx *c, const zk_mle *const *f, uint64_t k, uint64_t n_rps,
                                              const uint32_t *lens, const uint64_t sum[4], const uint64_t *rps,
                                              int32_t *out_ok) {
    if (n_rps && !lens) return ZK_ERR_BAD_ARG;
    static const uint32_t none = 0;
    return verify_common(c, f, k, n_rps, lens ? lens : &none, 0, sum, rps, out_ok);
}
extern "C" int32_t zk_sumcheck_verify(zk_ctx *c, const zk_mle *const *f, uint64_t k, uint64_t n_rps, uint32_t D,
                                      const uint64_t sum[4], const uint64_t *rps, int32_t *out_ok) {
    if (D >= kMaxSums) return ZK_ERR_BAD_ARG;
    return verify_common(c, f, k, n_rps, nullptr, D + 1, sum, rps, out_ok);
}

// ------------------------------------------------------------------------------------------------------------
// fft crate
// ------------------------------------------------------------------------------------------------------------
static int32_t make_twiddles(zk_ctx *c, uint32_t log_n, const Fe &omega, uint64_t **out, bool full = false) {
    const uint64_t count = full ? (1ull << log_n) : (log_n ? (1ull << (log_n - 1)) : 1);
    uint64_t *tw = nullptr;
    ZKCHK(raw_alloc(c, (size_t)count * 32, (void **)&tw));
    k_twiddle_table<<<grid_for((count + 63) / 64), kBlock, 0, c->stream>>>(tw, count, omega, c->fi->P);
    if (hipGetLastError() != hipSuccess) {
        (void)hipFree(tw);
        return ZK_ERR_HIP;
    }
    *out = tw;
    return ZK_OK;
}
static int32_t ntt_with_table(zk_ctx *c, const uint64_t *in, uint64_t *out, uint32_t log_n, const uint64_t *tw) {
    const uint64_t n = 1ull << log_n;
    k_bitrev_copy<<<grid_for(n), kBlock, 0, c->stream>>>(in, out, log_n);
    HIPCHK(hipGetLastError());
    for (uint32_t s = 0; s < log_n; ++s) {
        k_ntt_stage<<<grid_for(n / 2), kBlock, 0, c->stream>>>(out, tw, log_n, s, c->fi->P);
        HIPCHK(hipGetLastError());
    }
    return ZK_OK;
}
// ---- LDS-staged multi-pass NTT (ntt_kernels.cuh) for n >= 2^8 ----
static void ntt_make_plan(uint32_t log_n, NttPlan &pl) {
    pl.log_n = log_n;
    pl.n_pass = (log_n + kNttMaxLog - 1) / kNttMaxLog;
    if (pl.n_pass < 2) pl.n_pass = 2;
    const uint32_t base = log_n / pl.n_pass, rem = log_n % pl.n_pass;
    for (uint32_t p = 0; p < 4; ++p) pl.l[p] = p < pl.n_pass ? base + (p < rem ? 1 : 0) : 0;
    pl.lo_bits = log_n < 12 ? log_n : 12;
    pl.w_lo = nullptr;
    pl.w_hi = nullptr;
    for (int p = 0; p < 4; ++p) pl.w_full[p] = nullptr;
}
static int32_t ntt_build_tables(zk_ctx *c, NttPlan &pl, const Fe &omega) {
    const uint32_t hi_bits = pl.log_n - pl.lo_bits;
    uint32_t *lo = nullptr;
    uint64_t *hi = nullptr;
    ZKCHK(raw_alloc(c, ((size_t)kTw29Words * 4) << pl.lo_bits, (void **)&lo));
    if (raw_alloc(c, (size_t)32 << hi_bits, (void **)&hi) != ZK_OK) {
        (void)hipFree(lo);
        return ZK_ERR_ALLOC;
    }
    k_ntt_tables<<<grid_for((1ull << pl.lo_bits) + (1ull << hi_bits)), kBlock, 0, c->stream>>>(lo, hi, pl.lo_bits, hi_bits, omega, c->fi->P);
    if (hipGetLastError() != hipSuccess) {
        (void)hipFree(lo);
        (void)hipFree(hi);
        return ZK_ERR_HIP;
    }
    pl.w_lo = lo;
    pl.w_hi = hi;
    // full inter-pass tables for the middle passes while they stay <= 2^24 entries (512 MiB): a 32-byte read per element instead
    // of the multiplication that composes the twiddle from the two-level table -- the passes are bound by VALU issue, not by HBM
    // (ZK_NTT_FULL_TABLE_MAX_LOG: largest table built, log2 entries; 0 = compose everything.  A/B: profiles/r05_ntt_table_ab.log)
    static const uint32_t full_max_log = (uint32_t)env_u64("ZK_NTT_FULL_TABLE_MAX_LOG", 24, 0, 24);
    uint32_t lo_sum = 0;
    for (uint32_t p = 0; p + 1 < pl.n_pass; ++p) {
        const uint32_t log_entries = pl.log_n - lo_sum;   // R_p * I_p = n / O_p
        if (log_entries <= full_max_log) {
            uint64_t *t = nullptr;
            if (raw_alloc(c, (size_t)32 << log_entries, (void **)&t) == ZK_OK) {
                k_ntt_full_table<<<grid_for(1ull << log_entries), kBlock, 0, c->stream>>>(t, pl, log_entries - pl.l[p], pl.l[p], lo_sum, c->fi->P);
                if (hipGetLastError() == hipSuccess) pl.w_full[p] = t;
                else (void)hipFree(t);
            }
        }
        lo_sum += pl.l[p];
    }
    return ZK_OK;
}
static void ntt_free_tables(NttPlan &pl) {
    if (pl.w_lo) (void)hipFree((void *)pl.w_lo);
    if (pl.w_hi) (void)hipFree((void *)pl.w_hi);
    for (int p = 0; p < 4; ++p)
        if (pl.w_full[p]) (void)hipFree((void *)pl.w_full[p]);
    pl.w_lo = nullptr;
    pl.w_hi = nullptr;
}
template <int L>
static hipError_t ntt_launch_l(const NttPlan &pl, uint32_t p, bool last, uint32_t tiles, size_t lds, hipStream_t st, const uint64_t *src,
                               uint64_t *dst, const FieldParams &P, const Mul29 &scale, int do_scale) {
    hipError_t e;
    if (!last) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ntt_pass<L, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        k_ntt_pass<L, false><<<tiles, kNttThreads, lds, st>>>(src, dst, pl, p, P, scale, 0);
    } else {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ntt_pass<L, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        k_ntt_pass<L, true><<<tiles, kNttThreads, lds, st>>>(src, dst, pl, p, P, scale, do_scale);
    }
    return hipGetLastError();
}
static hipError_t ntt_launch_pass(const NttPlan &pl, uint32_t p, bool last, uint32_t tiles, size_t lds, hipStream_t st,
                                  const uint64_t *src, uint64_t *dst, const FieldParams &P, const Mul29 &scale, int do_scale) {
    switch (pl.l[p]) {
        case 4: return ntt_launch_l<4>(pl, p, last, tiles, lds, st, src, dst, P, scale, do_scale);
        case 5: return ntt_launch_l<5>(pl, p, last, tiles, lds, st, src, dst, P, scale, do_scale);
        case 6: return ntt_launch_l<6>(pl, p, last, tiles, lds, st, src, dst, P, scale, do_scale);
        case 7: return ntt_launch_l<7>(pl, p, last, tiles, lds, st, src, dst, P, scale, do_scale);
        case 8: return ntt_launch_l<8>(pl, p, last, tiles, lds, st, src, dst, P, scale, do_scale);
        default: return hipErrorInvalidValue;
    }
}
static int32_t ntt_run_plan(zk_ctx *c, const NttPlan &pl, const uint64_t *in, uint64_t *out, bool inverse) {
    const FieldParams &P = c->fi->P;
    const uint64_t n = 1ull << pl.log_n;
    uint64_t *scratch = nullptr;
    ZKCHK(pool_alloc(c, (size_t)n * 32, (void **)&scratch));
    Mul29 scale = {};
    if (inverse) {                                                               // fft/src/lib.rs:17: * F::from(n).inverse()
        const uint64_t nl[4] = {n, 0, 0, 0};
        scale = mul29_prepare(fe_inverse(fe_from_canonical(fe_from_u64limbs(nl), P), P), P);
    }
    int32_t rc = ZK_OK;
    const uint64_t *src = in;
    for (uint32_t p = 0; p < pl.n_pass && rc == ZK_OK; ++p) {
        const uint32_t R = 1u << pl.l[p];
        const size_t lds = (size_t)R * kNttRowBytes + (size_t)(R / 2) * kTw29Words * 4;   // one plane (halves take turns) + twiddles
        const uint32_t tiles = (uint32_t)(n / ((uint64_t)R * kNttCols));
        const bool last = p + 1 == pl.n_pass;
        hipError_t e = ntt_launch_pass(pl, p, last, tiles, lds, c->stream, src, last ? out : scratch, P, scale, (last && inverse) ? 1 : 0);
        if (!last) src = scratch;   // middle passes keep their addresses: later ones run in place on scratch
        if (e != hipSuccess) {
            g_hip_err = std::string("ntt pass: ") + hipGetErrorString(e);
            rc = ZK_ERR_HIP;
        }
    }
    pool_free(c, scratch, (size_t)n * 32);
    return rc;
}

extern "C" int32_t zk_ntt(zk_ctx *c, const zk_mle *in, int32_t inverse, zk_mle *out) {
    if (!c || !in || !out) return ZK_ERR_BAD_ARG;
    if (in->ctx != c || out->ctx != c) return ZK_ERR_CONTEXT_MISMATCH;
    if (in->n_vars != out->n_vars || in->d == out->d) return ZK_ERR_BAD_ARG;
    const uint32_t log_n = (uint32_t)in->n_vars;
    Fe omega;
    if (!field_root_of_unity(*c->fi, log_n, omega)) return ZK_ERR_FFT_NO_ROOT;   // fft/src/lib.rs:6
    ZKCHK(use_device(c));
    const FieldParams &P = c->fi->P;
    if (inverse) omega = fe_inverse(omega, P);                                   // fft/src/lib.rs:14
    auto key = std::make_pair(log_n, inverse ? 1 : 0);
    if (log_n >= 8) {
        auto pit = c->ntt_plans.find(key);
        if (pit == c->ntt_plans.end()) {
            NttPlan pl;
            ntt_make_plan(log_n, pl);
            ZKCHK(ntt_build_tables(c, pl, omega));
            pit = c->ntt_plans.emplace(key, pl).first;
        }
        return ntt_run_plan(c, pit->second, in->d, out->d, inverse != 0);
    }
    auto it = c->twiddles.find(key);
    if (it == c->twiddles.end()) {
        uint64_t *tw = nullptr;
        ZKCHK(make_twiddles(c, log_n, omega, &tw));
        it = c->twiddles.emplace(key, tw).first;
    }
    ZKCHK(ntt_with_table(c, in->d, out->d, log_n, it->second));
    if (inverse) {                                                               // fft/src/lib.rs:17
        const uint64_t nl[4] = {1ull << log_n, 0, 0, 0};                         // F::from(n).inverse()
        const Fe ninv = fe_inverse(fe_from_canonical(fe_from_u64limbs(nl), P), P);
        k_scale<<<grid_for(1ull << log_n), kBlock, 0, c->stream>>>(out->d, 1ull << log_n, ninv, P);
        HIPCHK(hipGetLastError());
    }
    return ZK_OK;
}
static int32_t fft_host_common(zk_ctx *c, const uint64_t *in, uint64_t n, uint64_t *out, int mode, const uint64_t *omega_user) {
    if (!c || !out || (!in && n)) return ZK_ERR_BAD_ARG;
    if (mode == 2) {                                        // fft_internal: len 1 returns, non power of two panics (:22-30)
        if (n == 0 || (n & (n - 1))) return ZK_ERR_FFT_NOT_POW2;
    } else {                                                // fft / ifft: get_root_of_unity(n) first (:6, :14)
        if (n == 0 || (n & (n - 1))) return ZK_ERR_FFT_NO_ROOT;
    }
    uint32_t log_n = 0;
    while ((1ull << log_n) < n) ++log_n;
    if (mode != 2 && log_n > c->fi->two_adicity) return ZK_ERR_FFT_NO_ROOT;
    if (log_n > kMaxVars) return ZK_ERR_UNSUPPORTED;
    zk_mle *a = nullptr, *b = nullptr;
    ZKCHK(zk_mle_upload(c, log_n, in, n, &a));
    int32_t rc = mle_alloc(c, log_n, &b);
    if (rc == ZK_OK) {
        // every other path uses the (u + t, u - t) butterfly, i.e. assumes omega^(n/2) = -1; fft_internal's caller may pass
        // any omega (fft/src/lib.rs:21), for which the reference's literal omega^(i + n/2) differs: full-table stages
        bool primitive = true;
        if (mode == 2 && log_n >= 1) {
            const FieldParams &P = c->fi->P;
            primitive = fe_eq(fe_pow_u64(fe_from_u64limbs(omega_user), n / 2, P), fe_sub(fe_zero(), fe_one(P), P));
        }
        if (mode == 2 && !primitive) {
            uint64_t *tw = nullptr;
            rc = make_twiddles(c, log_n, fe_from_u64limbs(omega_user), &tw, /*full=*/true);
            if (rc == ZK_OK) {
                k_bitrev_copy<<<grid_for(n), kBlock, 0, c->stream>>>(a->d, b->d, log_n);
                for (uint32_t s = 0; s < log_n; ++s)
                    k_ntt_stage_generic<<<grid_for(n / 2), kBlock, 0, c->stream>>>(b->d, tw, log_n, s, c->fi->P);
                if (hipGetLastError() != hipSuccess) rc = ZK_ERR_HIP;
            }
            if (hipStreamSynchronize(c->stream) != hipSuccess && rc == ZK_OK) rc = ZK_ERR_HIP;
            if (tw) (void)hipFree(tw);
        } else if (mode == 2 && log_n >= 8) {
            NttPlan pl;
            ntt_make_plan(log_n, pl);
            rc = ntt_build_tables(c, pl, fe_from_u64limbs(omega_user));
            if (rc == ZK_OK) rc = ntt_run_plan(c, pl, a->d, b->d, false);
            if (hipStreamSynchronize(c->stream) != hipSuccess && rc == ZK_OK) rc = ZK_ERR_HIP;
            ntt_free_tables(pl);
        } else if (mode == 2) {
            uint64_t *tw = nullptr;
            rc = make_twiddles(c, log_n, fe_from_u64limbs(omega_user), &tw);
            if (rc == ZK_OK) rc = ntt_with_table(c, a->d, b->d, log_n, tw);
            if (hipStreamSynchronize(c->stream) != hipSuccess && rc == ZK_OK) rc = ZK_ERR_HIP;
            if (tw) (void)hipFree(tw);
        } else {
            rc = zk_ntt(c, a, mode, b);
        }
    }
    if (rc == ZK_OK) rc = zk_mle_download(c, b, out);
    (void)zk_mle_free(c, a);
    (void)zk_mle_free(c, b);
    return rc;
}
extern "C" int32_t zk_fft_host(zk_ctx *c, const uint64_t *in, uint64_t n, uint64_t *out) { return fft_host_common(c, in, n, out, 0, nullptr); }
extern "C" int32_t zk_ifft_host(zk_ctx *c, const uint64_t *in, uint64_t n, uint64_t *out) { return fft_host_common(c, in, n, out, 1, nullptr); }
extern "C" int32_t zk_fft_internal_host(zk_ctx *c, const uint64_t *in, uint64_t n, const uint64_t omega[4], uint64_t *out) {
    if (!omega) return ZK_ERR_BAD_ARG;
    return fft_host_common(c, in, n, out, 2, omega);
}

// ------------------------------------------------------------------------------------------------------------
// measurement hooks
// ------------------------------------------------------------------------------------------------------------
extern "C" int32_t zk_bench_fold(zk_ctx *c, const zk_mle *t, const uint64_t r[4], zk_mle *out, int32_t reps, double *out_ms) {
    if (!c || !t || !r || !out || !out_ms || reps <= 0) return ZK_ERR_BAD_ARG;
    if (t->ctx != c || out->ctx != c) return ZK_ERR_CONTEXT_MISMATCH;
    if (t->n_vars == 0 || out->n_vars != t->n_vars - 1) return ZK_ERR_BAD_ARG;
    ZKCHK(use_device(c));
    const Fe rr = fe_from_u64limbs(r);
    HIPCHK(hipEventRecord(c->ev0, c->stream));
    for (int i = 0; i < reps; ++i) ZKCHK(launch_fold(c, t->d, out->d, t->n_vars, 0, rr));
    HIPCHK(hipEventRecord(c->ev1, c->stream));
    HIPCHK(hipEventSynchronize(c->ev1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    *out_ms = (double)ms / reps;
    return ZK_OK;
}
// the same with one event per launch boundary: out_ms_each[i] = duration of launch i (reps values); median / min / mean are
// the caller's to take (report median and min)
extern "C" int32_t zk_bench_fold_samples(zk_ctx *c, const zk_mle *t, const uint64_t r[4], zk_mle *out, int32_t reps, int32_t group,
                                         double *out_ms_each) {
    if (!c || !t || !r || !out || !out_ms_each || reps <= 0 || group <= 0 || reps > (1 << 20)) return ZK_ERR_BAD_ARG;
    if (t->ctx != c || out->ctx != c) return ZK_ERR_CONTEXT_MISMATCH;
    if (t->n_vars == 0 || out->n_vars != t->n_vars - 1) return ZK_ERR_BAD_ARG;
    ZKCHK(use_device(c));
    const Fe rr = fe_from_u64limbs(r);
    // an event record costs the stream ~4 us (r02: 22.5 vs 17.3 us per step on a 2^21 shard), so a sample may span `group` launches
    const int n_samples = (reps + group - 1) / group;
    if (n_samples > 65536) return ZK_ERR_BAD_ARG;
    std::vector<hipEvent_t> ev((size_t)n_samples + 1, nullptr);
    int32_t rc = ZK_OK;
    for (auto &e : ev)
        if (hipEventCreate(&e) != hipSuccess) rc = ZK_ERR_HIP;
    if (rc == ZK_OK && hipEventRecord(ev[0], c->stream) != hipSuccess) rc = ZK_ERR_HIP;
    for (int i = 0; i < reps && rc == ZK_OK; ++i) {
        rc = launch_fold(c, t->d, out->d, t->n_vars, 0, rr);
        const bool closes = (i + 1) % group == 0 || i + 1 == reps;
        if (rc == ZK_OK && closes && hipEventRecord(ev[i / group + 1], c->stream) != hipSuccess) rc = ZK_ERR_HIP;
    }
    if (rc == ZK_OK && hipEventSynchronize(ev[n_samples]) != hipSuccess) rc = ZK_ERR_HIP;
    for (int i = 0; i < n_samples && rc == ZK_OK; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ev[i], ev[i + 1]) != hipSuccess) rc = ZK_ERR_HIP;
        const int launches = i + 1 < n_samples ? group : reps - i * group;
        out_ms_each[i] = (double)ms / launches;
    }
    for (auto &e : ev)
        if (e) (void)hipEventDestroy(e);
    return rc;
}
extern "C" int32_t zk_bench_ntt(zk_ctx *c, const zk_mle *in, int32_t inverse, zk_mle *out, int32_t reps, double *out_ms) {
    if (!c || !in || !out || !out_ms || reps <= 0) return ZK_ERR_BAD_ARG;
    // builds the twiddle tables, then as many untimed transforms as timed ones: the passes are ALU-bound and follow the shader
    // clock, which keeps climbing for ~20 ms after idle (r02 kernel trace: 986 -> 720 us for the same kernel over 12 transforms)
    for (int i = 0; i <= reps; ++i) ZKCHK(zk_ntt(c, in, inverse, out));
    HIPCHK(hipEventRecord(c->ev0, c->stream));
    for (int i = 0; i < reps; ++i) ZKCHK(zk_ntt(c, in, inverse, out));
    HIPCHK(hipEventRecord(c->ev1, c->stream));
    HIPCHK(hipEventSynchronize(c->ev1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    *out_ms = (double)ms / reps;
    return ZK_OK;
}
extern "C" int32_t zk_bench_prove_partial(zk_ctx *c, zk_mle *const *f, uint64_t k, uint32_t D, const uint64_t sum[4], int32_t reps,
                                          double *out_ms_each) {
    if (!c || !f || !sum || !out_ms_each || reps <= 0) return ZK_ERR_BAD_ARG;
    ZKCHK(product_args(c, (const zk_mle *const *)f, k));
    const uint64_t n = f[0]->n_vars;
    std::vector<uint64_t> rp((size_t)(n ? n : 1) * (D + 1) * 4), ch((size_t)(n ? n : 1) * 4);
    for (int i = 0; i < reps; ++i) {
        HIPCHK(hipStreamSynchronize(c->stream));
        const auto t0 = std::chrono::steady_clock::now();
        ZKCHK(zk_sumcheck_prove(c, f, k, D, sum, 0, 0, rp.data(), ch.data()));
        out_ms_each[i] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return ZK_OK;
}
extern "C" int32_t zk_bench_evaluate(zk_ctx *c, const zk_mle *t, const uint64_t *point, uint64_t n_point, int32_t reps, double *out_ms_each) {
    if (!c || !t || !out_ms_each || reps <= 0) return ZK_ERR_BAD_ARG;
    uint64_t out[4];
    for (int i = 0; i < reps; ++i) {
        HIPCHK(hipStreamSynchronize(c->stream));
        const auto t0 = std::chrono::steady_clock::now();
        ZKCHK(zk_mle_evaluate(c, t, point, n_point, out));
        out_ms_each[i] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return ZK_OK;
}
// device time of evaluate: `reps` back-to-back enqueues of its launches (no host wait in between) between two HIP events on the
// context's stream -> average ms per evaluate.  What the roofline of the streaming kernel is priced on (bench.py roofline_evaluate).
extern "C" int32_t zk_bench_evaluate_device(zk_ctx *c, const zk_mle *t, const uint64_t *point, uint64_t n_point, int32_t reps, double *out_ms) {
    if (!c || !t || !out_ms || reps <= 0 || (!point && n_point)) return ZK_ERR_BAD_ARG;
    if (t->ctx != c) return ZK_ERR_CONTEXT_MISMATCH;
    if (n_point != t->n_vars) return ZK_ERR_EVAL_ARITY;
    ZKCHK(use_device(c));
    ZKCHK(evaluate_device(c, t, point, c->d_sums));   // warm: scratch from the pool
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipEventRecord(c->ev0, c->stream));
    for (int i = 0; i < reps; ++i) ZKCHK(evaluate_device(c, t, point, c->d_sums));
    HIPCHK(hipEventRecord(c->ev1, c->stream));
    HIPCHK(hipEventSynchronize(c->ev1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    *out_ms = (double)ms / reps;
    return ZK_OK;
}
extern "C" int32_t zk_bench_modmul(zk_ctx *c, int32_t variant, int32_t iters, double *out) {
    if (!c || !out || iters <= 0) return ZK_ERR_BAD_ARG;
    if (variant != 0 && variant != 1) return ZK_ERR_UNSUPPORTED;
    ZKCHK(use_device(c));
    const uint32_t blocks = 256 * 8;
    Fe seed = c->fi->two_adic_root;
    const Mul29 seed29 = mul29_prepare(seed, c->fi->P);
    k_bench_modmul<<<blocks, kBlock, 0, c->stream>>>(c->d_sums, 8, c->fi->P, seed, seed29, variant);   // warm-up
    HIPCHK(hipEventRecord(c->ev0, c->stream));
    k_bench_modmul<<<blocks, kBlock, 0, c->stream>>>(c->d_sums, iters, c->fi->P, seed, seed29, variant);
    HIPCHK(hipEventRecord(c->ev1, c->stream));
    HIPCHK(hipEventSynchronize(c->ev1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    *out = (double)blocks * kBlock * 2.0 * iters / (ms * 1e-3);
    return ZK_OK;
}
extern "C" int32_t zk_bench_copy(zk_ctx *c, uint64_t bytes, int32_t reps, double *out_gbps) {
    if (!c || !out_gbps || reps <= 0 || bytes < 16) return ZK_ERR_BAD_ARG;
    ZKCHK(use_device(c));
    uint4 *a = nullptr, *b = nullptr;
    HIPCHK(hipMalloc(&a, bytes));
    if (hipMalloc(&b, bytes) != hipSuccess) {
        (void)hipFree(a);
        return ZK_ERR_ALLOC;
    }
    (void)hipMemsetAsync(a, 1, bytes, c->stream);
    const uint64_t n16 = bytes / 16;
    k_bench_copy<<<kMaxGrid, kBlock, 0, c->stream>>>(a, b, n16);
    (void)hipEventRecord(c->ev0, c->stream);
    for (int i = 0; i < reps; ++i) k_bench_copy<<<kMaxGrid, kBlock, 0, c->stream>>>(a, b, n16);
    (void)hipEventRecord(c->ev1, c->stream);
    hipError_t e = hipEventSynchronize(c->ev1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
    (void)hipFree(a);
    (void)hipFree(b);
    if (e != hipSuccess) return ZK_ERR_HIP;
    *out_gbps = 2.0 * (double)n16 * 16.0 * reps / (ms * 1e-3) / 1e9;
    return ZK_OK;
}

#include "comm_host.inc"
#include "gkr_host.inc"
