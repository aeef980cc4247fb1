// The reference's own unit tests for the hot path, restated against the C++ host mirror (zk_amd/host/zk.hpp) so they
// read like the originals.  Runs on the GPU box (needs a gfx950 device); driven by tests/test_gpu_cpp_host.py.
//   polynomial/src/multilinear/evaluation_form.rs:111-202, polynomial/src/product_poly.rs:97-196,
//   sumcheck/src/lib.rs:53-122, fft/src/lib.rs:78-82
#include <cstdio>
#include <stdexcept>
#include <string>

#include "../../zk_amd/host/zk.hpp"

using namespace zk;
using Fr = Fe<Bls12_381Fr>;           // the reference's `use ark_bls12_381::Fr`
using F = Bls12_381Fr;

static int failures = 0;
#define ASSERT(cond) do { if (!(cond)) { std::printf("  ASSERT FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } } while (0)
#define ASSERT_EQ(a, b) ASSERT((a) == (b))
#define TEST(name) static void name(); static void run_##name() { std::printf("test %s\n", #name); name(); } static void name()

static std::vector<Fr> frs(std::initializer_list<long> v) {
    std::vector<Fr> out;
    for (long x : v) out.push_back(Fr::from_i64(x));
    return out;
}

// evaluation_form.rs:111-125
TEST(test_new_multilinear_poly) {
    ASSERT(MultiLinearPolynomial<F>::new_(2, frs({3, 1, 2})).is_err());
    ASSERT(std::string(MultiLinearPolynomial<F>::new_(2, frs({3, 1})).err()) == "evaluation vec len should equal 2^n_vars");
    ASSERT(MultiLinearPolynomial<F>::new_(1, frs({3, 1})).is_ok());
    ASSERT(MultiLinearPolynomial<F>::new_(2, frs({3, 1, 2, 5})).is_ok());
}
// evaluation_form.rs:128-146
TEST(test_partial_evaluate_single_variable) {
    auto poly = MultiLinearPolynomial<F>::new_(2, frs({3, 1, 2, 5})).unwrap();
    ASSERT_EQ(poly.partial_evaluate(0, {Fr::from(5)}).unwrap().evaluation_slice(), frs({-2, 21}));
    ASSERT_EQ(poly.partial_evaluate(0, {Fr::from(0)}).unwrap().evaluation_slice(), frs({3, 1}));
}
// evaluation_form.rs:149-171 : 2ab + 3bc, b = 2, c = 3
TEST(test_partial_evaluate_consecutive_variables) {
    auto poly = MultiLinearPolynomial<F>::new_(3, frs({0, 0, 0, 3, 0, 0, 2, 5})).unwrap();
    ASSERT_EQ(poly.partial_evaluate(1, {Fr::from(2), Fr::from(3)}).unwrap().evaluation_slice(), frs({18, 22}));
}
// evaluation_form.rs:174-202
TEST(test_evaluation) {
    auto poly = MultiLinearPolynomial<F>::new_(3, frs({0, 0, 0, 3, 0, 0, 2, 5})).unwrap();
    ASSERT(poly.evaluate({Fr::from(2), Fr::from(3)}).is_err());
    ASSERT_EQ(poly.evaluate({Fr::from(2), Fr::from(3), Fr::from(4)}).unwrap(), Fr::from(48));
}
// product_poly.rs:98-121
TEST(test_product_poly_new) {
    auto p1 = MultiLinearPolynomial<F>::new_(2, frs({2, 8, 10, 14})).unwrap();
    auto p2 = MultiLinearPolynomial<F>::new_(1, frs({3, 1})).unwrap();
    ASSERT(ProductPoly<F>::new_({p1, p2}).is_err());
    ASSERT(ProductPoly<F>::new_({}).is_err());
    ASSERT(ProductPoly<F>::new_({p1, p1}).is_ok());
}
// product_poly.rs:154-176 + :179-196
TEST(test_product_poly_partial_evaluate_and_prod_reduce) {
    auto p1 = MultiLinearPolynomial<F>::new_(2, frs({2, 8, 10, 14})).unwrap();
    auto p2 = MultiLinearPolynomial<F>::new_(2, frs({2, 8, 10, 22})).unwrap();
    auto prod = ProductPoly<F>::new_({p1, p2}).unwrap();
    ASSERT_EQ(prod.prod_reduce(), frs({4, 64, 100, 308}));
    auto pe = prod.partial_evaluate(1, {Fr::from(10)}).unwrap();
    ASSERT(pe.polynomials()[0] == p1.partial_evaluate(1, {Fr::from(10)}).unwrap());
    ASSERT(pe.polynomials()[1] == p2.partial_evaluate(1, {Fr::from(10)}).unwrap());
    ASSERT_EQ(pe.polynomials()[0].evaluation_slice(), frs({62, 50}));
}
// sumcheck/src/lib.rs:53-62
TEST(test_sumcheck_correct_sum_multilinear) {
    auto p = MultiLinearPolynomial<F>::new_(3, frs({0, 0, 0, 3, 0, 0, 2, 5})).unwrap();   // 2ab + 3bc
    auto prod_poly = ProductPoly<F>::new_({p}).unwrap();
    auto proof = SumcheckProver<1, F>::prove(prod_poly, Fr::from(10)).unwrap();
    ASSERT(SumcheckVerifier<F>::verify(prod_poly, proof).expect("proof is invalid"));
}
// sumcheck/src/lib.rs:64-100 : (2a + 3) * (ab)
TEST(test_correct_sum_multivariate_deg_2) {
    auto p1 = MultiLinearPolynomial<F>::new_(2, frs({3, 3, 5, 5})).unwrap();
    auto p2 = MultiLinearPolynomial<F>::new_(2, frs({0, 0, 0, 1})).unwrap();
    auto p = ProductPoly<F>::new_({p1, p2}).unwrap();
    auto proof = SumcheckProver<2, F>::prove(p, Fr::from(5)).unwrap();
    ASSERT(SumcheckVerifier<F>::verify(p, proof).expect("proof is invalid"));
}
// sumcheck/src/lib.rs:102-112
TEST(test_correct_sum_prove_partial) {
    auto p = MultiLinearPolynomial<F>::new_(3, frs({0, 0, 0, 3, 0, 0, 2, 5})).unwrap();
    auto prod_poly = ProductPoly<F>::new_({p}).unwrap();
    auto proof = SumcheckProver<1, F>::prove_partial(prod_poly, Fr::from(10)).unwrap().first;
    auto subclaim = SumcheckVerifier<F>::verify_partial(proof).expect("proof is invalid");
    ASSERT_EQ(prod_poly.evaluate(subclaim.challenges).unwrap(), subclaim.sum);
}
// sumcheck/src/lib.rs:114-122
TEST(test_invalid_sum) {
    auto p = MultiLinearPolynomial<F>::new_(3, frs({0, 0, 0, 3, 0, 0, 2, 5})).unwrap();
    auto prod_poly = ProductPoly<F>::new_({p}).unwrap();
    auto proof = SumcheckProver<1, F>::prove(prod_poly, Fr::from(12)).unwrap();
    auto res = SumcheckVerifier<F>::verify(prod_poly, proof);
    ASSERT(res.is_err());
    ASSERT(std::string(res.err()) == "verifier check failed: claimed_sum != p(0) + p(1)");
}
// fft/src/lib.rs:78-82 (use ark_bls12_377::Fr)
TEST(test_fft) {
    using Fq = Fe<Bls12_377Fr>;
    std::vector<Fq> a = {Fq::from(0), Fq::from(2), Fq::from(34), Fq::from(3434)};
    ASSERT_EQ(ifft<Bls12_377Fr>(fft<Bls12_377Fr>(a)), a);
}

// pairing_index.rs:61-96 (the literal pair lists), evaluation_form.rs:4 (Clone, PartialEq), transcript/src/lib.rs:32-34
TEST(test_index_pair_clone_eq_sample_n) {
    using PV = std::vector<std::pair<size_t, size_t>>;
    ASSERT_EQ(index_pair(3, 0), (PV{{0, 4}, {1, 5}, {2, 6}, {3, 7}}));
    ASSERT_EQ(index_pair(3, 1), (PV{{0, 2}, {1, 3}, {4, 6}, {5, 7}}));
    ASSERT_EQ(index_pair(3, 2), (PV{{0, 1}, {2, 3}, {4, 5}, {6, 7}}));
    ASSERT_EQ(index_pair(1, 0), (PV{{0, 1}}));
    ASSERT_EQ(mask(1), (size_t)1);        // pairing_index.rs:22-23 doc examples
    ASSERT_EQ(mask(3), (size_t)0b111);
    ASSERT_EQ(mask(0), (size_t)0);
    auto p = MultiLinearPolynomial<F>::new_(2, frs({3, 1, 2, 5})).unwrap();
    auto q = MultiLinearPolynomial<F>::new_(2, frs({3, 1, 2, 5})).unwrap();
    auto r = MultiLinearPolynomial<F>::new_(2, frs({3, 1, 2, 6})).unwrap();
    ASSERT(p == q);
    ASSERT(!(p == r));
    Transcript t1, t2;
    t1.append({1, 2, 3});
    t2.append({1, 2, 3});
    auto many = t1.sample_n_field_elements<F>(3);
    for (int i = 0; i < 3; ++i) ASSERT_EQ(many[i], t2.sample_field_element<F>());
    // fft_internal with the root fft itself uses reproduces fft (fft/src/lib.rs:4-8)
    auto a = frs({0, 2, 34, 3434});
    Fr w;
    ASSERT(zk_field_root_of_unity(F::id, 2, w.l.data()) == ZK_OK);
    ASSERT_EQ(fft_internal<F>(a, w), fft<F>(a));
}

// three independent prove_partial calls of sumcheck/src/lib.rs:64-112's polynomials proved side by side (zk_sumcheck_prove_batch): each
// equals the single call, and each sub-claim is the product at the challenge point
TEST(test_prove_partial_batch_equals_single_calls) {
    std::vector<ProductPoly<F>> polys;
    std::vector<Fr> sums = {Fr::from(5), Fr::from(5), Fr::from(7)};   // (the third: a wrong claim is proved all the same)
    for (int i = 0; i < 3; ++i) {
        auto p1 = MultiLinearPolynomial<F>::new_(2, frs({3, 3, 5, 5})).unwrap();
        auto p2 = MultiLinearPolynomial<F>::new_(2, frs({0, 0, 0, 1})).unwrap();
        polys.push_back(ProductPoly<F>::new_({p1, p2}).unwrap());
    }
    auto got = SumcheckProver<2, F>::prove_partial_batch(polys, sums).unwrap();
    ASSERT_EQ(got.size(), (size_t)3);
    for (int i = 0; i < 3; ++i) {
        auto one = SumcheckProver<2, F>::prove_partial(polys[i], sums[i]).unwrap();
        ASSERT(got[i].first.round_polys == one.first.round_polys);
        ASSERT_EQ(got[i].second, one.second);
    }
    auto subclaim = SumcheckVerifier<F>::verify_partial(got[0].first).expect("proof is invalid");
    ASSERT_EQ(polys[0].evaluate(subclaim.challenges).unwrap(), subclaim.sum);
}

int main() {
    try {
        run_test_new_multilinear_poly();
        run_test_partial_evaluate_single_variable();
        run_test_partial_evaluate_consecutive_variables();
        run_test_evaluation();
        run_test_product_poly_new();
        run_test_product_poly_partial_evaluate_and_prod_reduce();
        run_test_sumcheck_correct_sum_multilinear();
        run_test_correct_sum_multivariate_deg_2();
        run_test_correct_sum_prove_partial();
        run_test_invalid_sum();
        run_test_fft();
        run_test_index_pair_clone_eq_sample_n();
        run_test_prove_partial_batch_equals_single_calls();
    } catch (const std::exception &e) {
        std::printf("EXCEPTION: %s\n", e.what());
        return 2;
    }
    std::printf(failures ? "FAILED (%d)\n" : "ok: 12 reference tests + 1 batch test passed%.0d\n", failures);
    return failures ? 1 : 0;
}
