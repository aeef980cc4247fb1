// GKR-shaped driver through the C++ host mirror (zk_amd/host/zk.hpp) and the C ABI -- no Python in the data path.
// the reference has no gkr crate; its sumcheck tests (sumcheck/src/lib.rs:53-122) are the model for the
// shape of these (prove, verify, reject a wrong claim).  Runs on the GPU box; driven by tests/test_gpu_cpp_host.py.
#include <cstdio>
#include <stdexcept>
#include <string>

#include "../../zk_amd/host/zk.hpp"

using namespace zk;
using F = Bls12_381Fr;
using Fr = Fe<F>;

static int failures = 0;
#define ASSERT(cond) do { if (!(cond)) { std::printf("  ASSERT FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } } while (0)

static std::vector<Fr> frs(std::initializer_list<long> v) {
    std::vector<Fr> out;
    for (long x : v) out.push_back(Fr::from_i64(x));
    return out;
}

int main() {
    try {
        // (a + b) * (c * d) on 1,2,3,4 = 36: layer 1 = [a+b, c*d], layer 0 = [mul]
        std::printf("test gkr_small_circuit\n");
        auto circ = Circuit<F>::new_({Layer{0, 1, {1}, {0}, {1}}, Layer{1, 2, {0, 1}, {0, 2}, {1, 3}}}).unwrap();
        auto input = MultiLinearPolynomial<F>::new_(2, frs({1, 2, 3, 4})).unwrap();
        ASSERT(circ.evaluate(input).unwrap().evaluation_slice() == frs({36}));
        std::array<uint8_t, 32> seed{};
        auto res = circ.prove(input, seed).unwrap();
        ASSERT(res.first.evaluation_slice() == frs({36}));
        ASSERT(res.second.elements.size() == (6 * 1 + 2) + (6 * 2 + 2));
        ASSERT(circ.verify(input, res.first, seed, res.second).unwrap() == true);
        // a wrong output, a wrong input and a corrupted proof are rejected
        auto wrong_out = MultiLinearPolynomial<F>::new_(0, frs({37})).unwrap();
        ASSERT(circ.verify(input, wrong_out, seed, res.second).unwrap() == false);
        auto wrong_in = MultiLinearPolynomial<F>::new_(2, frs({1, 2, 3, 5})).unwrap();
        ASSERT(circ.verify(wrong_in, res.first, seed, res.second).unwrap() == false);
        auto bad = res.second;
        bad.elements[3] = Fr::from(7);
        ASSERT(circ.verify(input, res.first, seed, bad).unwrap() == false);
        // layer sizes must chain; the input must have log_in variables
        ASSERT(Circuit<F>::new_({Layer{0, 1, {1}, {0}, {1}}, Layer{2, 2, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}}}).is_err());
        ASSERT(circ.prove(MultiLinearPolynomial<F>::new_(1, frs({1, 2})).unwrap(), seed).is_err());

        // prove_partial on a sum of products: a*b + c, then verify_partial and the factors at the point
        std::printf("test prove_partial_terms\n");
        auto a = MultiLinearPolynomial<F>::new_(2, frs({1, 2, 3, 4})).unwrap();
        auto b = MultiLinearPolynomial<F>::new_(2, frs({5, 6, 7, 8})).unwrap();
        auto c = MultiLinearPolynomial<F>::new_(2, frs({9, 10, 11, 12})).unwrap();
        const Fr sum = Fr::from(5 + 12 + 21 + 32 + 9 + 10 + 11 + 12);
        auto tp = prove_partial_terms<2, F>({{a, b}, {c}}, sum).unwrap();
        SumcheckProof<F> proof{sum, tp.round_polys};
        auto sub = SumcheckVerifier<F>::verify_partial(proof).expect("proof is invalid");
        ASSERT(sub.challenges == tp.challenges);
        ASSERT(tp.finals[0] == a.evaluate(tp.challenges).unwrap());
        ASSERT(tp.finals[1] == b.evaluate(tp.challenges).unwrap());
        ASSERT(tp.finals[2] == c.evaluate(tp.challenges).unwrap());
        SumcheckProof<F> wrong{Fr::from(1), tp.round_polys};
        ASSERT(SumcheckVerifier<F>::verify_partial(wrong).is_err());
    } catch (const std::exception &e) {
        std::printf("EXCEPTION: %s\n", e.what());
        return 2;
    }
    std::printf(failures ? "FAILED (%d)\n" : "ok: gkr host tests passed%.0d\n", failures);
    return failures ? 1 : 0;
}
