"""Randomised soak: prove_partial / prove_partial_terms / evaluate / fold on random (field, k, D, n) against the CPU oracle and
the big-int model, bit for bit.  Not part of the test-suite (minutes); run on the GPU box: python tools/soak.py [seconds] [seed]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import zk_amd
from oracle import binding as orc
from oracle import gkr_ref
from zk_amd import MultiLinearPolynomial as MLE
from zk_amd import ProductPoly, SumcheckProver, gkr

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctxs = {f: zk_amd.Context(f, 0) for f in (0, 1, 2)}
t_end = time.time() + budget
n_cases = 0
t_beat = time.time() + 60
while time.time() < t_end:
    if time.time() > t_beat:
        print(f"  ... {n_cases} cases", flush=True)
        t_beat += 60
    f = rng.randrange(3)
    c = ctxs[f]
    p = zk_amd.modulus(f)
    kind = rng.choice(["prove", "prove", "terms", "evaluate", "fold", "gkr", "gkr_wide", "evaluate_big", "prod_reduce", "to_bytes", "coeff", "shard",
                       "batch", "batch", "prove_absorb", "prove_big"])
    if kind == "batch":   # zk_sumcheck_prove_batch: B independent proofs side by side (round 6), each against the oracle's own proof
        k, D = rng.choice([(2, 2), (2, 2), (3, 3), (3, 3), (1, 1), (2, 3), (4, 4)])
        n = rng.choice([1, 2, 3, 5, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17] if k <= 3 else [1, 4, 9, 11])
        B = rng.choice([2, 3, 5, 8, 9])
        cases = []
        for _ in range(B):
            tabs = [orc.fill_random(f, rng.randrange(1 << 30), 1 << n) for _ in range(k)]
            s = orc.fill_random(f, rng.randrange(1 << 30), 1)[0]
            cases.append((tabs, s, orc.sumcheck_prove(f, n, tabs, D, s, False)))
        polys = [ProductPoly.new([MLE.new(c, n, t) for t in tabs]) for tabs, _, _ in cases]
        got = SumcheckProver(D).prove_partial_batch(polys, np.stack([s for _, s, _ in cases]), consume=rng.random() < 0.5)
        for (proof, ch), (_, _, (want_rp, want_ch)) in zip(got, cases):
            assert np.array_equal(proof.round_polys, want_rp) and np.array_equal(ch, want_ch), ("batch", f, k, D, n, B)
    elif kind == "prove_big":   # sizes where the shipped thresholds pick the LDS-DMA round kernels (round 0 of two tables from 2^21 pairs)
        k, D = rng.choice([(2, 2), (2, 2), (3, 3)])
        n = rng.choice([22, 23] if k == 2 else [21, 22])
        tabs = [orc.fill_random(f, rng.randrange(1 << 30), 1 << n) for _ in range(k)]
        s = orc.fill_random(f, rng.randrange(1 << 30), 1)[0]
        want_rp, want_ch = orc.sumcheck_prove(f, n, tabs, D, s, False)
        pp = ProductPoly.new([MLE.new(c, n, t) for t in tabs])
        proof, ch = SumcheckProver(D).prove_partial(pp, s, consume=True)
        assert np.array_equal(proof.round_polys, want_rp) and np.array_equal(ch, want_ch), ("prove_big", f, k, D, n)
    elif kind == "prove_absorb":   # prove (tables absorbed, prover.rs:15-20) where the chunked serialiser runs: 1, 2 and 4 chunks per table
        k, D = rng.choice([(1, 1), (2, 2), (2, 2), (3, 3)])
        n = rng.choice([18, 19, 20, 21])
        tabs = [orc.fill_random(f, rng.randrange(1 << 30), 1 << n) for _ in range(k)]
        s = orc.fill_random(f, rng.randrange(1 << 30), 1)[0]
        want_rp, _ = orc.sumcheck_prove(f, n, tabs, D, s, True)
        pp = ProductPoly.new([MLE.new(c, n, t) for t in tabs])
        proof = SumcheckProver(D).prove(pp, s, consume=rng.random() < 0.5)
        assert np.array_equal(proof.round_polys, want_rp), ("prove_absorb", f, k, D, n)
        for q in pp.polynomials:
            q.free()
    elif kind == "prove":
        k = rng.choice([1, 2, 2, 3, 3, 4, 5, 8])
        D = rng.choice([max(1, k), k, k + 1, rng.randrange(1, 7)])
        n = rng.choice([1, 2, 3, 5, 8, 10, 11, 12, 13, 14, 15, 16, 17] if k <= 3 else [1, 4, 9, 11, 13])
        tabs = [orc.fill_random(f, rng.randrange(1 << 30), 1 << n) for _ in range(k)]
        s = orc.fill_random(f, rng.randrange(1 << 30), 1)[0]   # any claimed sum: the transcript must still match
        want_rp, want_ch = orc.sumcheck_prove(f, n, tabs, D, s, False)
        pp = ProductPoly.new([MLE.new(c, n, t) for t in tabs])
        proof, ch = SumcheckProver(D).prove_partial(pp, s, consume=rng.random() < 0.5)
        assert np.array_equal(proof.round_polys, want_rp) and np.array_equal(ch, want_ch), ("prove", f, k, D, n)
    elif kind == "shard":   # W shard provers in this process, the lane all-reduce by hand, vs the oracle's unsharded proof (round 5:
        # at 2^17+ local pairs the lanes carry [S(0), 0, .., L] and k_lanes_transcript derives S(1), S(D) from the reduced values)
        import torch
        from zk_amd.distributed import GpuShardBackend, shard_of
        world = rng.choice([1, 2, 2, 4, 8])
        k, D = rng.choice([(2, 2), (2, 2), (3, 3), (1, 1), (2, 3), (1, 2)])
        w = world.bit_length() - 1
        n = w + rng.choice([1, 2, 5, 9, 12, 14, 17, 18] if k <= 2 else [1, 4, 9, 13, 17])
        tabs = [orc.fill_random(f, rng.randrange(1 << 30), 1 << n) for _ in range(k)]
        s = orc.fill_random(f, rng.randrange(1 << 30), 1)[0]
        want_rp, want_ch = orc.sumcheck_prove(f, n, tabs, D, s, False)
        backs = [GpuShardBackend(ProductPoly.new([MLE.new(c, n - w, shard_of(t, g, world)) for t in tabs]), D, s, world) for g in range(world)]
        gb = rng.choice([0, 0, 3, 10, 13])
        while backs[0].local_vars_left() > gb:
            lanes = [b.round_begin() for b in backs]
            total = torch.stack(lanes).sum(dim=0)
            for b, l in zip(backs, lanes):
                l.copy_(total)
                b.round_finish()
        gathered = torch.cat([b.tail().clone() for b in backs])
        for b in backs:
            b.tail_rounds(gathered)
        for b in backs:
            rp, ch = b.results()
            assert np.array_equal(rp, want_rp) and np.array_equal(ch, want_ch), ("shard", f, world, k, D, n, gb)
            b.close()
    elif kind == "terms":
        shape = rng.choice([[2, 1], [2, 1], [3, 1], [1, 1], [2, 2], [2, 1, 1], [3, 2]])
        D = max(shape) + rng.randrange(2)
        n = rng.choice([1, 3, 6, 9, 10, 11, 12, 13, 14])
        tabs = [[[rng.randrange(p) for _ in range(1 << n)] for _ in range(kk)] for kk in shape]
        s = rng.randrange(p)
        want = gkr_ref.prove_partial_terms(f, tabs, D, s)
        poly = gkr.SumOfProductsPoly([[MLE.new(c, n, zk_amd.fe_from_ints(f, t)) for t in term] for term in tabs])
        rp, ch, fin = gkr.prove_partial_terms(poly, D, zk_amd.fe_from_int(f, s))
        assert [zk_amd.fe_to_ints(f, r) for r in rp] == want[0] and zk_amd.fe_to_ints(f, ch) == want[1], ("terms", f, shape, D, n)
        assert zk_amd.fe_to_ints(f, fin) == want[2], ("terms finals", f, shape, D, n)
    elif kind == "gkr":
        depth = rng.randrange(1, 4)
        logs = [rng.randrange(0, 6)] + [rng.randrange(1, 7) for _ in range(depth)]
        layers = []
        for i in range(depth):
            ng, nin = 1 << logs[i], 1 << logs[i + 1]
            heavy = rng.random() < 0.2
            layers.append((logs[i], logs[i + 1], [rng.randrange(2) for _ in range(ng)],
                           [0 if heavy else rng.randrange(nin) for _ in range(ng)], [rng.randrange(nin) for _ in range(ng)]))
        inputs = [rng.randrange(p) for _ in range(1 << logs[-1])]
        seed = bytes(rng.randrange(256) for _ in range(32))
        want_out, want_proof = gkr_ref.gkr_prove(f, layers, inputs, seed)
        circ = gkr.Circuit(c)
        for lo, li, op, left, right in layers:
            circ.add_layer(lo, li, op, left, right)
        x = MLE.new(c, logs[-1], zk_amd.fe_from_ints(f, inputs))
        out, proof = gkr.gkr_prove(circ, x, seed)
        assert zk_amd.fe_to_ints(f, proof) == want_proof and zk_amd.fe_to_ints(f, out.evaluation_slice()) == want_out, ("gkr", f, logs)
        assert gkr.gkr_verify(circ, x, out, seed, proof), ("gkr verify", f, logs)
        bad = proof.copy()
        bad[rng.randrange(len(want_proof))] = zk_amd.fe_from_int(f, rng.randrange(p))
        assert not gkr.gkr_verify(circ, x, out, seed, bad) or zk_amd.fe_to_ints(f, bad) == want_proof, ("gkr tamper", f, logs)
    elif kind == "gkr_wide":
        # widths around the bookkeeping kernels' tiling (224 rows per workgroup, 256 entries per chunk, rows of more than 256
        # entries on their own kernel and emptied in the light CSR): uniform, skewed (rows of tens of entries: several chunks)
        # and heavy (a few rows take most of the gates) wirings, against the big-int model
        depth = rng.randrange(1, 3)
        logs = [rng.randrange(6, 12)] + [rng.randrange(5, 11) for _ in range(depth)]
        layers = []
        for i in range(depth):
            ng, nin = 1 << logs[i], 1 << logs[i + 1]
            mode = rng.choice(["uniform", "skew", "heavy", "heavy_both"])
            def pick(side):
                if mode == "skew":
                    return rng.randrange(max(nin // 32, 1)) if rng.random() < 0.7 else rng.randrange(nin)
                if mode == "heavy" and side == 0 or mode == "heavy_both":
                    return rng.choice([3 % nin, nin - 1]) if rng.random() < 0.6 else rng.randrange(nin)
                return rng.randrange(nin)
            layers.append((logs[i], logs[i + 1], [rng.randrange(2) for _ in range(ng)], [pick(0) for _ in range(ng)], [pick(1) for _ in range(ng)]))
        inputs = [rng.randrange(p) for _ in range(1 << logs[-1])]
        seed = bytes(rng.randrange(256) for _ in range(32))
        want_out, want_proof = gkr_ref.gkr_prove(f, layers, inputs, seed)
        circ = gkr.Circuit(c)
        for lo, li, op, left, right in layers:
            circ.add_layer(lo, li, op, left, right)
        x = MLE.new(c, logs[-1], zk_amd.fe_from_ints(f, inputs))
        out, proof = gkr.gkr_prove(circ, x, seed)
        assert zk_amd.fe_to_ints(f, proof) == want_proof and zk_amd.fe_to_ints(f, out.evaluation_slice()) == want_out, ("gkr_wide", f, logs)
        assert gkr.gkr_verify(circ, x, out, seed, proof), ("gkr_wide verify", f, logs)
        bad = proof.copy()
        bad[rng.randrange(len(want_proof))] = zk_amd.fe_from_int(f, rng.randrange(p))
        assert not gkr.gkr_verify(circ, x, out, seed, bad) or zk_amd.fe_to_ints(f, bad) == want_proof, ("gkr_wide tamper", f, logs)
        circ.free()
    elif kind == "evaluate_big":   # k_eval_stream from 21 variables (L = 12, 13), k_eval_low below
        if rng.random() < 0.7:
            n_cases -= 1           # (rare: a 2^21-2^22 table costs seconds of host generation)
        else:
            n = rng.choice([20, 21, 21, 22])
            t = orc.fill_random(f, rng.randrange(1 << 30), 1 << n)
            pt = orc.fill_random(f, rng.randrange(1 << 30), n)
            if rng.random() < 0.3:
                pt[rng.randrange(n)] = orc.from_int(f, rng.choice([0, 1, p - 1]))
            m = MLE.new(c, n, t)
            assert np.array_equal(m.evaluate(pt), orc.mle_evaluate(f, n, t, pt)), ("evaluate_big", f, n)
            m.free()
    elif kind == "prod_reduce":
        k, n = rng.choice([1, 2, 2, 3, 5]), rng.randrange(0, 15)
        tabs = [orc.fill_random(f, rng.randrange(1 << 30), 1 << n) for _ in range(k)]
        got = ProductPoly.new([MLE.new(c, n, t) for t in tabs]).prod_reduce()
        assert np.array_equal(got, orc.prod_reduce(f, n, tabs)), ("prod_reduce", f, k, n)
    elif kind == "to_bytes":
        n = rng.randrange(0, 21)
        t = orc.fill_random(f, rng.randrange(1 << 30), 1 << n)
        assert MLE.new(c, n, t).to_bytes() == orc.mle_to_bytes(f, n, t), ("to_bytes", f, n)
    elif kind == "coeff":
        n = rng.randrange(1, 12)
        nt = rng.randrange(1, min(1 << n, 400) + 1)
        keys = sorted(rng.sample(range(1 << n), nt))
        coeffs = orc.fill_random(f, rng.randrange(1 << 30), nt)
        poly = zk_amd.CoeffMultilinearPolynomial.new_with_coefficient(f, n, {kk: coeffs[i] for i, kk in enumerate(keys)})
        got = poly.to_evaluation_form(c).evaluation_slice()
        assert np.array_equal(got, orc.coeff_to_evaluation(f, n, keys, coeffs)), ("coeff", f, n, nt)
    elif kind == "evaluate":
        n = rng.randrange(0, 19)
        t = orc.fill_random(f, rng.randrange(1 << 30), 1 << n)
        pt = orc.fill_random(f, rng.randrange(1 << 30), max(n, 1))[:n]
        got = MLE.new(c, n, t).evaluate(pt)
        assert np.array_equal(got, orc.mle_evaluate(f, n, t, pt)), ("evaluate", f, n)
    else:
        n = rng.randrange(1, 17)
        iv = rng.randrange(n)
        na = rng.randrange(1, n - iv + 1)
        t = orc.fill_random(f, rng.randrange(1 << 30), 1 << n)
        a = orc.fill_random(f, rng.randrange(1 << 30), na)
        got = MLE.new(c, n, t).partial_evaluate(iv, a).evaluation_slice()
        assert np.array_equal(got, orc.mle_partial_evaluate(f, n, t, iv, a)), ("fold", f, n, iv, na)
    n_cases += 1
print(f"soak ok: {n_cases} random cases bit-exact in {budget:.0f} s")
