// zk.hpp -- C++17 host-side mirror of the reference's public API for the hot path, over the C ABI (include/zk_amd.h).
//
// The reference is Rust; the image has no Rust toolchain, so this header is the compiled-language host side: the same
// type and method names, argument meaning and error behaviour as
//   polynomial::multilinear::evaluation_form::MultiLinearPolynomial<F>   evaluation_form.rs:7-103
//   polynomial::product_poly::ProductPoly<F>                             product_poly.rs:7-88
//   sumcheck::prover::SumcheckProver<MAX_VAR_DEGREE, F>                  prover.rs:9-73
//   sumcheck::verifier::SumcheckVerifier<F>                              verifier.rs:9-78
//   transcript::Transcript                                               transcript/src/lib.rs:5-35
//   fft::{fft, ifft}                                                     fft/src/lib.rs:4-19
// `Result<T, &'static str>` is zk::Result<T> (value or the reference's message); `F` is a field tag type.  Tables stay
// resident on the GPU behind the handle; `evaluation_slice()` downloads.  bindings/rust/ is the same thing in Rust.
#pragma once
#include <array>
#include <cstdint>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../../include/zk_amd.h"

namespace zk {

struct Bn254Fr { static constexpr int32_t id = ZK_FIELD_BN254_FR; };
struct Bls12_381Fr { static constexpr int32_t id = ZK_FIELD_BLS12_381_FR; };
struct Bls12_377Fr { static constexpr int32_t id = ZK_FIELD_BLS12_377_FR; };

// a field element exactly as ark-ff stores it: 4 LE u64 limbs, Montgomery form
template <class F>
struct Fe {
    std::array<uint64_t, 4> l{};
    static Fe from(uint64_t v) {   // F::from(v)
        Fe e;
        zk_fe_from_u64(F::id, v, e.l.data());
        return e;
    }
    static Fe from_i64(int64_t v) {   // Fr::from(-2) in the reference's tests
        if (v >= 0) return from((uint64_t)v);
        uint64_t p[4], c[4];
        zk_field_modulus(F::id, p);
        uint64_t borrow = (uint64_t)(-v);   // p - |v|
        for (int i = 0; i < 4; ++i) {
            c[i] = p[i] - borrow;
            borrow = p[i] < borrow ? 1 : 0;
        }
        Fe e;
        zk_fe_from_canonical(F::id, c, e.l.data());
        return e;
    }
    bool operator==(const Fe &o) const { return l == o.l; }
    bool operator!=(const Fe &o) const { return !(*this == o); }
};

// Result<T, &'static str>
template <class T>
class Result {
    std::unique_ptr<T> v_;
    const char *err_ = nullptr;

public:
    Result(T v) : v_(new T(std::move(v))) {}
    Result(int32_t status) : err_(zk_strerror(status)) {}
    bool is_ok() const { return err_ == nullptr; }
    bool is_err() const { return err_ != nullptr; }
    const char *err() const { return err_; }
    T &unwrap() {
        if (err_) throw std::runtime_error(std::string("called `Result::unwrap()` on an `Err` value: ") + err_);
        return *v_;
    }
    T &expect(const char *msg) {
        if (err_) throw std::runtime_error(std::string(msg) + ": " + err_);
        return *v_;
    }
};

// one device context per field tag (the reference has no such notion: arithmetic is ambient)
template <class F>
inline zk_ctx *context() {
    static zk_ctx *ctx = [] {
        zk_ctx *c = nullptr;
        const int32_t rc = zk_ctx_create(F::id, 0, &c);
        if (rc != ZK_OK) throw std::runtime_error(std::string("zk_ctx_create: ") + zk_strerror(rc));
        return c;
    }();
    return ctx;
}

template <class F>
class MultiLinearPolynomial {
    struct Handle {
        zk_mle *h = nullptr;
        ~Handle() { if (h) zk_mle_free(context<F>(), h); }
    };
    std::shared_ptr<Handle> h_;
    explicit MultiLinearPolynomial(zk_mle *h) : h_(std::make_shared<Handle>()) { h_->h = h; }
    template <class> friend class ProductPoly;
    template <uint8_t, class> friend class SumcheckProver;
    template <class> friend class Circuit;

public:
    // evaluation_form.rs:15-27
    static Result<MultiLinearPolynomial> new_(size_t n_vars, const std::vector<Fe<F>> &evaluations) {
        zk_mle *h = nullptr;
        const int32_t rc = zk_mle_upload(context<F>(), n_vars, reinterpret_cast<const uint64_t *>(evaluations.data()),
                                         evaluations.size(), &h);
        if (rc != ZK_OK) return rc;
        return MultiLinearPolynomial(h);
    }
    size_t n_vars() const {   // :30
        uint64_t n = 0;
        zk_mle_n_vars(h_->h, &n);
        return (size_t)n;
    }
    // :40-80
    Result<MultiLinearPolynomial> partial_evaluate(size_t initial_var, const std::vector<Fe<F>> &assignments) const {
        zk_mle *o = nullptr;
        const int32_t rc = zk_mle_partial_evaluate(context<F>(), h_->h, initial_var,
                                                   reinterpret_cast<const uint64_t *>(assignments.data()), assignments.size(), &o);
        if (rc != ZK_OK) return rc;
        return MultiLinearPolynomial(o);
    }
    // :83-89
    Result<Fe<F>> evaluate(const std::vector<Fe<F>> &assignments) const {
        Fe<F> out;
        const int32_t rc = zk_mle_evaluate(context<F>(), h_->h, reinterpret_cast<const uint64_t *>(assignments.data()),
                                           assignments.size(), out.l.data());
        if (rc != ZK_OK) return rc;
        return out;
    }
    // :92-94 (downloads)
    std::vector<Fe<F>> evaluation_slice() const {
        std::vector<Fe<F>> v((size_t)1 << n_vars());
        zk_mle_download(context<F>(), h_->h, reinterpret_cast<uint64_t *>(v.data()));
        return v;
    }
    // :97-103
    std::vector<uint8_t> to_bytes() const {
        std::vector<uint8_t> b((size_t)32 << n_vars());
        zk_mle_to_bytes(context<F>(), h_->h, b.data());
        return b;
    }
    bool operator==(const MultiLinearPolynomial &o) const {   // #[derive(PartialEq)]: compared on the device (zk_mle_equal)
        int32_t eq = 0;
        if (zk_mle_equal(context<F>(), h_->h, o.h_->h, &eq) != ZK_OK) return false;
        return eq != 0;
    }
    zk_mle *raw() const { return h_->h; }
};

template <class F>
class ProductPoly {
    std::vector<MultiLinearPolynomial<F>> polys_;
    explicit ProductPoly(std::vector<MultiLinearPolynomial<F>> p) : polys_(std::move(p)) {}
    std::vector<const zk_mle *> handles() const {
        std::vector<const zk_mle *> h;
        for (auto &p : polys_) h.push_back(p.raw());
        return h;
    }
    template <uint8_t, class> friend class SumcheckProver;
    template <class> friend class SumcheckVerifier;

public:
    // product_poly.rs:14-32
    static Result<ProductPoly> new_(std::vector<MultiLinearPolynomial<F>> polynomials) {
        std::vector<const zk_mle *> h;
        for (auto &p : polynomials) h.push_back(p.raw());
        const int32_t rc = zk_product_check(h.data(), h.size());
        if (rc != ZK_OK) return rc;
        return ProductPoly(std::move(polynomials));
    }
    size_t n_vars() const { return polys_[0].n_vars(); }   // :86
    const std::vector<MultiLinearPolynomial<F>> &polynomials() const { return polys_; }
    Result<Fe<F>> evaluate(const std::vector<Fe<F>> &assignments) const {   // :36-44
        Fe<F> out;
        auto h = handles();
        const int32_t rc = zk_product_evaluate(context<F>(), h.data(), h.size(), reinterpret_cast<const uint64_t *>(assignments.data()),
                                               assignments.size(), out.l.data());
        if (rc != ZK_OK) return rc;
        return out;
    }
    Result<ProductPoly> partial_evaluate(size_t initial_var, const std::vector<Fe<F>> &assignments) const {   // :48-63
        std::vector<MultiLinearPolynomial<F>> out;
        for (auto &p : polys_) {
            auto r = p.partial_evaluate(initial_var, assignments);
            if (r.is_err()) return Result<ProductPoly>(ZK_ERR_PANIC_INDEX);
            out.push_back(r.unwrap());
        }
        return ProductPoly(std::move(out));
    }
    std::vector<Fe<F>> prod_reduce() const {   // :66-74
        auto h = handles();
        zk_mle *o = nullptr;
        std::vector<Fe<F>> v((size_t)1 << n_vars());
        if (zk_prod_reduce(context<F>(), h.data(), h.size(), &o) == ZK_OK) {
            zk_mle_download(context<F>(), o, reinterpret_cast<uint64_t *>(v.data()));
            zk_mle_free(context<F>(), o);
        }
        return v;
    }
    std::vector<uint8_t> to_bytes() const {   // :77-83
        std::vector<uint8_t> out;
        for (auto &p : polys_) {
            auto b = p.to_bytes();
            out.insert(out.end(), b.begin(), b.end());
        }
        return out;
    }
};

template <class F>
struct SumcheckProof {   // sumcheck/src/lib.rs:8-11 (fields private in the reference)
    Fe<F> sum;
    std::vector<std::vector<Fe<F>>> round_polys;
};
template <class F>
struct SubClaim {   // sumcheck/src/lib.rs:17-20
    Fe<F> sum;
    std::vector<Fe<F>> challenges;
};

template <uint8_t MAX_VAR_DEGREE, class F>
class SumcheckProver {
    static Result<std::pair<SumcheckProof<F>, std::vector<Fe<F>>>> run(const ProductPoly<F> &poly, Fe<F> sum, int absorb) {
        const size_t n = poly.n_vars(), ns = (size_t)MAX_VAR_DEGREE + 1;
        std::vector<zk_mle *> h;
        for (auto &p : poly.polys_) h.push_back(p.raw());
        std::vector<uint64_t> rp(4 * n * ns + 4), ch(4 * n + 4);
        const int32_t rc = zk_sumcheck_prove(context<F>(), h.data(), h.size(), MAX_VAR_DEGREE, sum.l.data(), absorb, /*consume=*/0,
                                             rp.data(), ch.data());
        if (rc != ZK_OK) return rc;
        SumcheckProof<F> proof{sum, {}};
        std::vector<Fe<F>> challenges(n);
        for (size_t r = 0; r < n; ++r) {
            std::vector<Fe<F>> row(ns);
            for (size_t t = 0; t < ns; ++t)
                for (int i = 0; i < 4; ++i) row[t].l[i] = rp[4 * (r * ns + t) + i];
            proof.round_polys.push_back(row);
            for (int i = 0; i < 4; ++i) challenges[r].l[i] = ch[4 * r + i];
        }
        return std::make_pair(proof, challenges);
    }

public:
    static Result<SumcheckProof<F>> prove(const ProductPoly<F> &poly, Fe<F> sum) {   // prover.rs:15-20
        auto r = run(poly, sum, 1);
        if (r.is_err()) return Result<SumcheckProof<F>>(ZK_ERR_PANIC_INDEX);
        return r.unwrap().first;
    }
    static Result<std::pair<SumcheckProof<F>, std::vector<Fe<F>>>> prove_partial(const ProductPoly<F> &poly, Fe<F> sum) {   // :24-30
        return run(poly, sum, 0);
    }
    // polys.size() independent prove_partial calls (one ProductPoly each, same factor count and arity) proved side by side:
    // zk_sumcheck_prove_batch, one launch per round for all of them; element i equals prove_partial(polys[i], sums[i])
    static Result<std::vector<std::pair<SumcheckProof<F>, std::vector<Fe<F>>>>> prove_partial_batch(const std::vector<ProductPoly<F>> &polys,
                                                                                                  const std::vector<Fe<F>> &sums) {
        using Out = std::vector<std::pair<SumcheckProof<F>, std::vector<Fe<F>>>>;
        if (polys.size() != sums.size()) return Result<Out>(ZK_ERR_BAD_ARG);
        if (polys.empty()) return Out{};
        const size_t B = polys.size(), k = polys[0].polys_.size(), n = polys[0].n_vars(), ns = (size_t)MAX_VAR_DEGREE + 1;
        std::vector<zk_mle *> h;
        std::vector<uint64_t> s;
        for (size_t b = 0; b < B; ++b) {
            if (polys[b].polys_.size() != k) return Result<Out>(ZK_ERR_ARITY_MISMATCH);
            for (auto &p : polys[b].polys_) h.push_back(p.raw());
            s.insert(s.end(), sums[b].l.begin(), sums[b].l.end());
        }
        std::vector<uint64_t> rp(4 * B * n * ns + 4), ch(4 * B * n + 4);
        const int32_t rc = zk_sumcheck_prove_batch(context<F>(), B, h.data(), k, MAX_VAR_DEGREE, s.data(), /*consume=*/0, rp.data(), ch.data());
        if (rc != ZK_OK) return Result<Out>(rc);
        Out out;
        for (size_t b = 0; b < B; ++b) {
            SumcheckProof<F> proof{sums[b], {}};
            std::vector<Fe<F>> challenges(n);
            for (size_t r = 0; r < n; ++r) {
                std::vector<Fe<F>> row(ns);
                for (size_t t = 0; t < ns; ++t)
                    for (int i = 0; i < 4; ++i) row[t].l[i] = rp[4 * ((b * n + r) * ns + t) + i];
                proof.round_polys.push_back(row);
                for (int i = 0; i < 4; ++i) challenges[r].l[i] = ch[4 * (b * n + r) + i];
            }
            out.emplace_back(proof, challenges);
        }
        return out;
    }
};

template <class F>
class SumcheckVerifier {
    // round_polys is a Vec<Vec<F>>: every round goes to the library at its own length (verifier.rs:55-58)
    static std::vector<uint64_t> flatten(const SumcheckProof<F> &proof, std::vector<uint32_t> &lens) {
        std::vector<uint64_t> rp;
        lens.clear();
        for (auto &row : proof.round_polys) {
            lens.push_back((uint32_t)row.size());
            for (auto &e : row) rp.insert(rp.end(), e.l.begin(), e.l.end());
        }
        lens.push_back(0);
        rp.resize(rp.size() + 4);
        return rp;
    }

public:
    static Result<bool> verify(const ProductPoly<F> &poly, const SumcheckProof<F> &proof) {   // verifier.rs:15-33
        std::vector<uint32_t> lens;
        auto rp = flatten(proof, lens);
        auto h = poly.handles();
        int32_t ok = 0;
        const int32_t rc = zk_sumcheck_verify_lengths(context<F>(), h.data(), h.size(), proof.round_polys.size(), lens.data(),
                                                      proof.sum.l.data(), rp.data(), &ok);
        if (rc != ZK_OK) return rc;
        return ok != 0;
    }
    static Result<SubClaim<F>> verify_partial(const SumcheckProof<F> &proof) {   // verifier.rs:38-41
        std::vector<uint32_t> lens;
        auto rp = flatten(proof, lens);
        const size_t n = proof.round_polys.size();
        std::vector<uint64_t> ch(4 * n + 4);
        SubClaim<F> sub;
        const int32_t rc = zk_sumcheck_verify_partial_lengths(F::id, n, lens.data(), proof.sum.l.data(), rp.data(),
                                                              sub.sum.l.data(), ch.data());
        if (rc != ZK_OK) return rc;
        sub.challenges.resize(n);
        for (size_t r = 0; r < n; ++r)
            for (int i = 0; i < 4; ++i) sub.challenges[r].l[i] = ch[4 * r + i];
        return sub;
    }
};

class Transcript {   // transcript/src/lib.rs:5-35
    zk_transcript *t_ = nullptr;

public:
    Transcript() { zk_transcript_new(&t_); }
    ~Transcript() { zk_transcript_free(t_); }
    Transcript(const Transcript &) = delete;
    void append(const std::vector<uint8_t> &new_data) { zk_transcript_append(t_, new_data.data(), new_data.size()); }
    template <class F>
    Fe<F> sample_field_element() {
        Fe<F> e;
        zk_transcript_sample_field_element(t_, F::id, e.l.data());
        return e;
    }
    template <class F>
    std::vector<Fe<F>> sample_n_field_elements(size_t n) {
        std::vector<Fe<F>> v(n);   // transcript/src/lib.rs:32-34
        zk_transcript_sample_n_field_elements(t_, F::id, n, reinterpret_cast<uint64_t *>(v.data()));
        return v;
    }
};

// fft/src/lib.rs:4-19.  The reference panics on bad lengths; here that is a std::runtime_error carrying the message.
template <class F>
inline std::vector<Fe<F>> fft(const std::vector<Fe<F>> &coefficients) {
    std::vector<Fe<F>> out(coefficients.size());
    const int32_t rc = zk_fft_host(context<F>(), reinterpret_cast<const uint64_t *>(coefficients.data()), coefficients.size(),
                                   reinterpret_cast<uint64_t *>(out.data()));
    if (rc != ZK_OK) throw std::runtime_error(zk_strerror(rc));
    return out;
}
// fft/src/lib.rs:21-46 -- any omega; "values must be a power of 2" as the reference
template <class F>
inline std::vector<Fe<F>> fft_internal(const std::vector<Fe<F>> &values, const Fe<F> &omega) {
    std::vector<Fe<F>> out(values.size());
    const int32_t rc = zk_fft_internal_host(context<F>(), reinterpret_cast<const uint64_t *>(values.data()), values.size(), omega.l.data(),
                                            reinterpret_cast<uint64_t *>(out.data()));
    if (rc != ZK_OK) throw std::runtime_error(zk_strerror(rc));
    return out;
}
// polynomial/src/multilinear/pairing_index.rs:24-26: a bit sequence of n ones (mask(1) -> 1, mask(3) -> 0b111).  The reference
// overflows (panics in debug builds) at n >= usize::BITS; here that is the index panic status.
inline size_t mask(uint8_t n) {
    if (n >= sizeof(size_t) * 8) throw std::runtime_error(zk_strerror(ZK_ERR_PANIC_INDEX));
    return ((size_t)1 << n) - 1;
}
// polynomial/src/multilinear/pairing_index.rs:2-9 (host index arithmetic; the kernels compute the same indices inline)
inline std::vector<std::pair<size_t, size_t>> index_pair(uint8_t n_vars, uint8_t index) {
    if (n_vars == 0 || index > n_vars - 1) throw std::runtime_error(zk_strerror(ZK_ERR_PANIC_INDEX));
    const unsigned pos = n_vars - 1 - index;
    std::vector<std::pair<size_t, size_t>> out;
    for (size_t j = 0; j < ((size_t)1 << (n_vars - 1)); ++j) {
        const size_t left = ((j >> pos) << (pos + 1)) | (j & mask((uint8_t)pos));
        out.emplace_back(left, left | ((size_t)1 << pos));
    }
    return out;
}
template <class F>
inline std::vector<Fe<F>> ifft(const std::vector<Fe<F>> &evaluations) {
    std::vector<Fe<F>> out(evaluations.size());
    const int32_t rc = zk_ifft_host(context<F>(), reinterpret_cast<const uint64_t *>(evaluations.data()), evaluations.size(),
                                    reinterpret_cast<uint64_t *>(out.data()));
    if (rc != ZK_OK) throw std::runtime_error(zk_strerror(rc));
    return out;
}

// ---- GKR-shaped driver (the reference has no gkr crate; formats are this library's) ----------
// prove_partial (prover.rs:24-30) on a sum of products: terms[i] = the factors of product i.
template <class F>
struct TermsProof {
    std::vector<std::vector<Fe<F>>> round_polys;
    std::vector<Fe<F>> challenges, finals;   // finals: every factor at the challenge point, term after term
};
template <uint8_t MAX_VAR_DEGREE, class F>
inline Result<TermsProof<F>> prove_partial_terms(const std::vector<std::vector<MultiLinearPolynomial<F>>> &terms, Fe<F> sum) {
    std::vector<zk_mle *> h;
    std::vector<uint64_t> tk;
    for (auto &t : terms) {
        tk.push_back(t.size());
        for (auto &p : t) h.push_back(p.raw());
    }
    if (h.empty()) return Result<TermsProof<F>>(ZK_ERR_EMPTY_PRODUCT);
    const size_t n = terms[0][0].n_vars(), ns = (size_t)MAX_VAR_DEGREE + 1;
    std::vector<uint64_t> rp(4 * n * ns + 4), ch(4 * n + 4), fin(4 * h.size());
    const int32_t rc = zk_sumcheck_prove_terms(context<F>(), h.data(), tk.data(), tk.size(), MAX_VAR_DEGREE, sum.l.data(), 0,
                                               rp.data(), ch.data(), fin.data());
    if (rc != ZK_OK) return rc;
    TermsProof<F> out;
    out.challenges.resize(n);
    out.finals.resize(h.size());
    for (size_t r = 0; r < n; ++r) {
        std::vector<Fe<F>> row(ns);
        for (size_t t = 0; t < ns; ++t)
            for (int i = 0; i < 4; ++i) row[t].l[i] = rp[4 * (r * ns + t) + i];
        out.round_polys.push_back(row);
        for (int i = 0; i < 4; ++i) out.challenges[r].l[i] = ch[4 * r + i];
    }
    for (size_t f = 0; f < h.size(); ++f)
        for (int i = 0; i < 4; ++i) out.finals[f].l[i] = fin[4 * f + i];
    return out;
}

struct Layer {   // 2^log_out gates over 2^log_in values; op 0 = add, 1 = mul
    size_t log_out, log_in;
    std::vector<uint8_t> op;
    std::vector<uint32_t> left, right;
};
template <class F>
struct GkrProof {
    std::vector<Fe<F>> elements;   // per layer [round polys #1 | round polys #2 | W(u) | W(v)]
};
template <class F>
class Circuit {
    struct Handle {
        zk_circuit *h = nullptr;
        ~Handle() { if (h) zk_circuit_free(h); }
    };
    std::shared_ptr<Handle> h_;
    explicit Circuit(zk_circuit *h) : h_(std::make_shared<Handle>()) { h_->h = h; }

public:
    // layers[0] = output layer
    static Result<Circuit> new_(const std::vector<Layer> &layers) {
        zk_circuit *z = nullptr;
        int32_t rc = zk_circuit_create(context<F>(), &z);
        if (rc != ZK_OK) return rc;
        Circuit c(z);
        for (auto &l : layers) {
            if (l.op.size() != ((size_t)1 << l.log_out) || l.left.size() != l.op.size() || l.right.size() != l.op.size())
                return Result<Circuit>(ZK_ERR_BAD_ARG);
            rc = zk_circuit_add_layer(z, l.log_out, l.log_in, l.op.data(), l.left.data(), l.right.data());
            if (rc != ZK_OK) return rc;
        }
        return c;
    }
    Result<MultiLinearPolynomial<F>> evaluate(const MultiLinearPolynomial<F> &input) const {
        zk_mle *o = nullptr;
        const int32_t rc = zk_gkr_evaluate(h_->h, input.raw(), &o);
        if (rc != ZK_OK) return rc;
        return MultiLinearPolynomial<F>(o);
    }
    // -> (outputs, proof)
    Result<std::pair<MultiLinearPolynomial<F>, GkrProof<F>>> prove(const MultiLinearPolynomial<F> &input,
                                                                   const std::array<uint8_t, 32> &seed) const {
        uint64_t n = 0;
        zk_circuit_proof_elems(h_->h, &n);
        GkrProof<F> proof;
        proof.elements.resize(n);
        zk_mle *o = nullptr;
        const int32_t rc = zk_gkr_prove(h_->h, input.raw(), seed.data(), &o, reinterpret_cast<uint64_t *>(proof.elements.data()));
        if (rc != ZK_OK) return rc;
        return std::make_pair(MultiLinearPolynomial<F>(o), proof);
    }
    // Ok(true) accept, Ok(false) reject, Err on misuse
    Result<bool> verify(const MultiLinearPolynomial<F> &input, const MultiLinearPolynomial<F> &outputs,
                        const std::array<uint8_t, 32> &seed, const GkrProof<F> &proof) const {
        uint64_t n = 0;
        zk_circuit_proof_elems(h_->h, &n);
        if (proof.elements.size() != n) return Result<bool>(ZK_ERR_BAD_ARG);
        const int32_t rc = zk_gkr_verify(h_->h, input.raw(), outputs.raw(), seed.data(),
                                         reinterpret_cast<const uint64_t *>(proof.elements.data()));
        if (rc == ZK_OK) return true;
        if (rc == ZK_ERR_VERIFY_SUM || rc == ZK_ERR_GKR_REJECT) return false;
        return rc;
    }
};

}  // namespace zk
