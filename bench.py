#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X hot path (BASELINE.json metric).

Metric: field-ops/sec of the MLE fold (one sumcheck-round restriction, `partial_evaluate(0, [r])`,
polynomial/src/multilinear/evaluation_form.rs:40-80) on a 2^24-element BN254-Fr table, 3 field ops per pair
(1 mul + 2 sub, evaluation_form.rs:68).  A "step" is one fold of the WHOLE 2^24 table: read 512 MiB, write 256 MiB
(algorithmic bytes 48 * 2^24 = 805,306,368 B).  Inputs are synthetic (device-generated, resident in HBM
before the timed region).

N = 1: the table sits on one GPU.  N > 1 ("strong" scaling, BASELINE config 3 / north_star's ">= 6x at 8 GPUs"): the SAME
2^24 table is sharded by index mod N, every rank folds its 2^24 / N elements -- the fold has no exchange
step, so there is no data-path collective -- and the sumcheck prover on the same sharded table (n = 24, k = 2, D = 2: one
RCCL all-reduce of (D+1)*8 lanes per round, inside zk_shard_prover_run) is timed next to it, with the per-round collective
latency.  The sumcheck prover wall-clock (second half of the metric) is `sumcheck_prover_wall_clock_ms`.

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line.  N > 1 runs one rank per GPU over RCCL:
either the caller launches it (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...`,
WORLD_SIZE set: rank 0 prints the line), or the plain command is given and bench.py starts that same launcher itself as a
child job (self_launch: the parent makes no GPU call, relays rank 0's line and the job's exit code, enforces
--launch-timeout).
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_VARS = 24
ALG_BYTES_PER_FOLD = 48 * (1 << N_VARS)
FIELD_OPS_PER_FOLD = 3 * (1 << (N_VARS - 1))
HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec


def host_cores():
    # the GPU box gives a one-GPU job a 16-core share even though it reports every core of the host
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


PROVER_SEED = 0x5EED0000   # + n: factor f of the n-variable prover is the generator stream [f << n, (f + 1) << n)


def oracle_prover_rows(field, ns, tabs, claimed, cache):
    """the CPU restatement's prove_partial (k = 2, D = 2) on `tabs`: faithful, single thread (prover.rs:33-73 as written)
    and -- from 2^20 -- the fused OpenMP form, bit-compared with it.  Fills cache[ns] = {faithful_ms, fused_ms, threads, proof}."""
    import numpy as np

    from oracle import binding as orc

    t1 = time.perf_counter()
    want = orc.sumcheck_prove(field, ns, tabs, 2, claimed, False)
    row = {"faithful_ms": (time.perf_counter() - t1) * 1e3, "proof": want}
    if ns >= 20:
        t1 = time.perf_counter()
        rp, ch, used_p = orc.sumcheck_prove_fused_parallel(field, ns, tabs, 2, claimed, host_cores())
        row["fused_ms"] = (time.perf_counter() - t1) * 1e3
        row["threads"] = used_p
        assert np.array_equal(rp, want[0]) and np.array_equal(ch, want[1]), "fused CPU prover differs from the faithful one"
    cache[ns] = row
    return row


def parity_gate(ctx, field, table, out, r):
    """BASELINE.md 3 / SURVEY 8d: "GPU output == CPU restatement output, limb for limb, on the timed inputs ... bit-compare
    outputs before timing".  (a) the timed fold: all 2^23 outputs of partial_evaluate(0, [r]) on the timed 2^24 table against
    the oracle's faithful fold (evaluation_form.rs:40-80); (b) the n = 20 and n = 24 proofs of the timed prover inputs (every
    round polynomial and challenge) against the oracle's faithful prover (prover.rs:33-73); (c) to_bytes of the 2^24 table; (d)
    `prove` with the tables absorbed at n = 20 / 24 -- every call bench.py publishes a time for.  The oracle runs of (b) and (d)
    are the cpu_baseline's prover rows: same tables, one run.  Returns (gate dict, cache for cpu_baseline)."""
    import numpy as np

    import zk_amd
    from oracle import binding as orc

    gate, cache = {}, {}
    table.fold_into(r, out)
    tab = table.evaluation_slice()
    t1 = time.perf_counter()
    want = orc.mle_partial_evaluate(field, N_VARS, tab, 0, np.asarray(r).reshape(1, 4))
    cache["fold_2p24_faithful_ms"] = (time.perf_counter() - t1) * 1e3
    gate["fold_2p24"] = bool(np.array_equal(out.evaluation_slice(), want))
    gate["generator_matches_oracle"] = bool(np.array_equal(tab[:1 << 16], orc.fill_random(field, 0x5EED0000 + 24, 1 << 16)))
    # (c) to_bytes of the same 2^24 table (evaluation_form.rs:97-103: 32 chunks of 16 MiB through the device serialiser and the
    # copy-out helpers), every byte against the oracle's
    t1 = time.perf_counter()
    want_bytes = np.frombuffer(orc.mle_to_bytes(field, N_VARS, tab), dtype=np.uint8)
    cache["to_bytes_2p24_faithful_ms"] = (time.perf_counter() - t1) * 1e3
    gate["to_bytes_2p24"] = bool(np.array_equal(table.to_bytes_array(), want_bytes))
    del want, tab, want_bytes
    for n in (20, 24):
        polys = [zk_amd.MultiLinearPolynomial.random(ctx, n, PROVER_SEED + n, f << n) for f in range(2)]
        tabs = [q.evaluation_slice() for q in polys]
        pp = zk_amd.ProductPoly.new(polys)
        s = pp.round_sums(1)
        claimed = zk_amd.fe_from_int(field, zk_amd.fe_to_int(field, s[0]) + zk_amd.fe_to_int(field, s[1]))
        proof, ch = zk_amd.SumcheckProver(2).prove_partial(pp, claimed)
        row = oracle_prover_rows(field, n, tabs, claimed, cache)
        gate[f"prove_n{n}"] = bool(np.array_equal(proof.round_polys, row["proof"][0]) and np.array_equal(ch, row["proof"][1]))
        # (d) `prove` (prover.rs:15-20: poly.to_bytes() absorbed first, 2 / 32 chunks per table through absorb_tables) on the same tables
        t1 = time.perf_counter()
        want_abs = orc.sumcheck_prove(field, n, tabs, 2, claimed, True)
        row["faithful_absorbing_ms"] = (time.perf_counter() - t1) * 1e3
        got_abs = zk_amd.SumcheckProver(2).prove(pp, claimed)
        gate[f"prove_absorbing_n{n}"] = bool(np.array_equal(got_abs.round_polys, want_abs[0]))
        for q in polys:
            q.free()
        del tabs
    return gate, cache


def gate_set(result, key, ok):
    """record one more parity check of a published row (rows measured outside parity_gate() are checked where their inputs live)"""
    if isinstance(result.get("parity_gate"), dict):
        result["parity_gate"][key] = bool(ok)


def cpu_baseline(field, cache=None):
    """CPU restatement (oracle) of the same path on bounded samples, rank 0 at N = 1 only.  Faithful rows are single
    threaded (the reference is); the "optimised" rows use every core of the box's share.  The prover rows at n = 20 / 24 are
    the parity gate's oracle runs when the gate ran (same tables as the GPU leg, bit-compared with it)."""
    import numpy as np

    from oracle import binding as orc

    cache = dict(cache or {})
    ncores = host_cores()
    n = 21   # 2^21 elements = 64 MiB: same streaming pattern, bounded run time
    tab = orc.fill_random(field, 0x5EED0000 + 24, 1 << n)
    r = orc.fill_random(field, 0xC4A11, 1)
    t0 = time.perf_counter()
    reps = 0
    while True:
        out = orc.mle_partial_evaluate(field, n, tab, 0, r)
        reps += 1
        if time.perf_counter() - t0 > 6.0:
            break
    dt = time.perf_counter() - t0
    ops = 3 * (1 << (n - 1)) * reps
    t1 = time.perf_counter()
    reps_par = 0
    while True:
        out_par, used = orc.fold_msb_parallel(field, n, tab, r[0], threads=ncores)
        reps_par += 1
        if time.perf_counter() - t1 > 3.0:
            break
    dt_par = time.perf_counter() - t1
    assert np.array_equal(out_par, out), "parallel CPU fold differs from the faithful fold"
    # the other half of the metric on the CPU: reference-faithful prover ((D+2)*k folds + (D+1) prod_reduce per round),
    # configs[0] (n = 12), configs[1] (n = 20) and configs[2]'s problem size (n = 24) of BASELINE.json
    cpu_prove, cpu_fused, used_p = {}, {}, ncores
    for ns in (12, 16, 20, 24):
        if ns not in cache:   # the same tables the GPU leg proves (bench.py: PROVER_SEED), claimed sum = the true sum
            tabs = [orc.fill_random(field, PROVER_SEED + ns, 1 << ns, first_index=f << ns) for f in range(2)]
            s = np.zeros(4, dtype=np.uint64)
            for e in orc.prod_reduce(field, ns, tabs) if ns <= 16 else []:
                s = orc.add(field, s, e)
            oracle_prover_rows(field, ns, tabs, s, cache)
            del tabs
        cpu_prove[ns] = cache[ns]["faithful_ms"]
        if "fused_ms" in cache[ns]:
            cpu_fused[ns] = cache[ns]["fused_ms"]
            used_p = cache[ns]["threads"]
    # the reference's own criterion bench on the CPU restatement: evaluate at 20 variables (n clone + fold + copy steps)
    t20 = orc.fill_random(field, 0x5EED0E00 + 20, 1 << 20)
    pt20 = orc.fill_random(field, 0xE7A1, 20)
    t1 = time.perf_counter()
    orc.mle_evaluate(field, 20, t20, pt20)
    cpu_eval_ms = (time.perf_counter() - t1) * 1e3
    # config 5 on the CPU: the reference's recursive fft (fft/src/lib.rs:21-46: one omega.pow per butterfly output), faithful,
    # single thread, at 2^16 / 2^18 / 2^20; cost model n * log2(n) butterflies * ~log2(n) multiplies per pow
    fft_ms = {}
    for lg in (16, 18, 20):
        x = orc.fill_random(field, 0x5EED0005, 1 << lg)
        t1 = time.perf_counter()
        orc.fft(field, x)
        fft_ms[lg] = (time.perf_counter() - t1) * 1e3
    fft_2p24_est = fft_ms[20] * 16.0 * (24.0 / 20.0) ** 2
    return {
        "value": ops / dt,
        "unit": "field-ops/s",
        "cores": 1,
        "kind": "port",
        "evaluate_ms_n20": cpu_eval_ms,
        "sample": f"{reps} folds of a 2^{n}-element BN254-Fr table (clone + fold + copy as evaluation_form.rs:49-79), "
                  f"{dt:.1f} s, single thread (the reference is single-threaded)",
        "sumcheck_prove_partial_ms_n12_k2_d2": cpu_prove[12],
        "sumcheck_prove_partial_ms_n16_k2_d2": cpu_prove[16],
        "sumcheck_prove_partial_ms_n20_k2_d2": cpu_prove[20],
        "sumcheck_prove_partial_ms_n24_k2_d2": cpu_prove[24],
        "prover_rows_source": ("the parity gate's oracle runs: the SAME tables and claimed sums the GPU leg proves, proofs "
                               "bit-compared" if cache.get("from_gate") else "same generator tables as the GPU leg (gate not run)"),
        "fft_faithful_recursive_ms": {f"2p{lg}": v for lg, v in fft_ms.items()},
        "fft_faithful_recursive_ms_2p24_extrapolated": fft_2p24_est,
        "fft_extrapolation": "T(2^24) = T(2^20) * 16 * (24/20)^2: n*log2(n)/2 butterflies, each two pow() of ~log2(n) multiplies",
        "optimised": {"value": 3 * (1 << (n - 1)) * reps_par / dt_par, "unit": "field-ops/s", "cores": used,
                      "sample": f"{reps_par} fused out-of-place folds of the same table, OpenMP, {dt_par:.1f} s",
                      "sumcheck_prove_partial_ms_n20_k2_d2": cpu_fused[20], "sumcheck_prove_partial_ms_n24_k2_d2": cpu_fused[24],
                      "prover": f"fused rounds (one pass for the D+1 products, one for the fold), OpenMP, {used_p} threads, "
                                "bit-identical proof"},
    }


def pmc_traffic():
    """HBM bytes per k_fold_msb launch, measured now: two child runs of tools/pmc_fold.py under rocprofv3 (FETCH_SIZE and
    WRITE_SIZE in separate --pmc passes with --kernel-trace only; KiB; FETCH_SIZE doubled on gfx950 for 16-B/lane
    streaming reads -- MI355X_MICROARCH.md "HBM").  Returns (bytes or None, note)."""
    import csv
    import glob

    if "rocprof" in os.environ.get("LD_PRELOAD", "") or os.environ.get("ROCPROFILER_REGISTER_FORCE_LOAD"):
        return None, "bench.py itself runs under a profiler: nested PMC collection skipped"
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not on PATH"
    vals = {}
    tmp = tempfile.mkdtemp(prefix="zk_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            p = subprocess.run([exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "--",
                                sys.executable, os.path.join(ROOT, "tools", "pmc_fold.py"), str(N_VARS), "3"],
                               cwd="/tmp", env=env, capture_output=True, text=True, timeout=240)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if p.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {ctr} failed (rc {p.returncode}): {(p.stderr or p.stdout)[-200:]}"
            got = [float(row["Counter_Value"]) for row in csv.DictReader(open(files[0]))
                   if "k_fold_msb" in row["Kernel_Name"] and row.get("Counter_Name", ctr) == ctr]
            if not got:
                return None, f"no k_fold_msb rows in the {ctr} pass"
            vals[ctr] = sum(got) / len(got)
        return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, \
            (f"live: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/pmc_fold.py; "
             f"FETCH_SIZE {vals['FETCH_SIZE']:.0f} KiB x2 (gfx950) + WRITE_SIZE {vals['WRITE_SIZE']:.0f} KiB")
    except Exception as e:   # never costs the headline line
        return None, "PMC leg failed: " + repr(e)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _field_sum(zk_amd, field, elems):
    p = zk_amd.modulus(field)
    return zk_amd.fe_from_int(field, sum(zk_amd.fe_to_int(field, e) for e in elems) % p)


def sharded_leg(args, ctx, field, dist, torch, rank, world, ns, tdev, rehearse, result):
    """N > 1 (and torchrun at N = 1): the n = 24, k = 2, D = 2 prover on tables sharded by index mod world
    (zk_shard_prover_run: prover.rs:44-68 with one all-reduce per round).  The timing is only published when the proof is
    RIGHT: (a) the claimed sum is the true sum (all ranks' local sums added), (b) the proof passes verify_partial and its
    subclaim equals the product of the factors evaluated at the challenges (verifier.rs:15-33's final check, evaluated on the
    shards), (c) on rank 0 it equals, bit for bit, the proof of the UNSHARDED 2^24 tables made on rank 0's own GPU."""
    import numpy as np

    import zk_amd
    from zk_amd.distributed import GpuShardBackend, HostComm, RcclComm, ntt_sharded

    ex = result.setdefault("extra", {})
    lw = world.bit_length() - 1
    n = ns + lw
    comm = HostComm(ctx) if rehearse else RcclComm(ctx)
    # what the TRANSPORT says about itself (zk_comm_info: ncclGetVersion / ncclCommCount / ncclCommUserRank), gathered from every
    # rank: a record with n_gpus = N must show N distinct RCCL ranks of an N-rank communicator, each on its own device
    info = comm.info()
    it = torch.tensor([info["rccl_version"], info["ranks"], info["rank"], ctx.device],
                      dtype=torch.int64, device=tdev)
    infos = [torch.empty_like(it) for _ in range(world)]
    dist.all_gather(infos, it)
    infos = [[int(v) for v in t.cpu()] for t in infos]
    ex["comm_transport"] = "host callbacks (rehearsal)" if rehearse else "rccl"
    ex["rccl_version"] = infos[0][0]
    ex["comm_reported_ranks"] = sorted({r[1] for r in infos})
    ex["comm_rank_ids"] = sorted(r[2] for r in infos)
    ex["rank_devices"] = [r[3] for r in infos]
    ex["comm_confirms_n_ranks"] = ex["comm_reported_ranks"] == [world] and ex["comm_rank_ids"] == list(range(world))
    seeds = (0x5EED0100, 0x5EED0200)
    shards = [zk_amd.MultiLinearPolynomial.random(ctx, ns, sd, first_index=rank << ns) for sd in seeds]

    def gather_elems(local):   # (m, 4) uint64 per rank -> (world, m, 4) on every rank
        t = torch.from_numpy(np.ascontiguousarray(local, dtype=np.uint64).view(np.int64).copy()).to(tdev)
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        return np.stack([o.cpu().numpy().view(np.uint64) for o in outs])

    # (a) the true claimed sum: sum over ranks of the local S(0) + S(1)
    s_loc = zk_amd.ProductPoly.new(shards).round_sums(1)
    claimed = _field_sum(zk_amd, field, gather_elems(s_loc).reshape(-1, 4))

    def run_once(gb, phases=False):
        pp = zk_amd.ProductPoly.new([q.clone() for q in shards])
        backend = GpuShardBackend(pp, 2, claimed, world, torch_stream=False)
        ctx.synchronize()
        dist.barrier()
        t1 = time.perf_counter()
        res = backend.run_phases(comm, gb) if phases else backend.run(comm, gb)   # whole loop inside the library; synchronises
        dt = time.perf_counter() - t1
        backend.close()
        for q in pp.polynomials:
            q.free()
        return res, dt

    # gather_below g: rounds run sharded (one all-reduce each) while the local tables have more than 2^g elements, then one
    # all-gather and the remaining rounds replicated.  Where the two meet depends on the fabric's small-message latency (a
    # collective round costs two small launches + the all-reduce, a replicated one a round kernel on N x the data), so the
    # line reports a few settings and names the best.
    per_gb, proofs, phases_by_gb = {}, {}, {}
    for gb in (10, 13, 16):
        if gb >= ns:
            continue
        ts = []
        for it in range(7):
            (rp, ch), dt = run_once(gb)
            ts.append(dt)
        tt = torch.tensor([sorted(ts[2:])[len(ts[2:]) // 2]], dtype=torch.float64, device=tdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        per_gb[gb] = float(tt.item()) * 1e3
        proofs[gb] = (rp, ch)
        (_, _, ph), _ = run_once(gb, phases=True)
        pt = torch.tensor([ph["local_kernels_ms"], ph["allreduce_ms"], ph["gather_ms"], ph["tail_rounds_ms"]], dtype=torch.float64,
                          device=tdev)
        dist.all_reduce(pt, op=dist.ReduceOp.MAX)
        phases_by_gb[gb] = dict(zip(("local_kernels_ms", "allreduce_ms", "gather_ms", "tail_rounds_ms"), [float(v) for v in pt.cpu()]))

    # (b) every rank holds the same proof; it verifies; the subclaim is the product at the challenges (evaluated on the shards:
    # the last lw variables select the rank -- index = local * world + rank -- so f(ch) = sum_r eq(ch[ns:], r) * f_r(ch[:ns]))
    verified, same_all = bool(per_gb), True
    p = zk_amd.modulus(field)
    for gb, (rp, ch) in proofs.items():
        all_ch = gather_elems(ch)
        all_rp = gather_elems(rp.reshape(-1, 4))
        same_all = same_all and all(np.array_equal(all_ch[q], all_ch[0]) and np.array_equal(all_rp[q], all_rp[0]) for q in range(world))
        sub = zk_amd.SumcheckVerifier.verify_partial(field, zk_amd.SumcheckProof(claimed, rp))   # raises when a round check fails
        vals = gather_elems(np.stack([q.evaluate(ch[:ns]) for q in shards]))                     # (world, k, 4)
        cs = [zk_amd.fe_to_int(field, e) for e in ch[ns:]]
        prod = 1
        for f in range(len(shards)):
            acc = 0
            for q in range(world):
                w = 1
                for i, c_i in enumerate(cs):   # variable ns + i is bit (lw - 1 - i) of the rank
                    w = w * (c_i if (q >> (lw - 1 - i)) & 1 else (1 - c_i)) % p
                acc = (acc + w * zk_amd.fe_to_int(field, vals[q][f])) % p
            prod = prod * acc % p
        verified = verified and np.array_equal(sub.challenges, ch) and zk_amd.fe_to_int(field, sub.sum) == prod
    vt = torch.tensor([int(verified and same_all)], device=tdev)
    dist.all_reduce(vt, op=dist.ReduceOp.MIN)
    verified_all = bool(vt.item())

    # (c) rank 0: the unsharded tables (global index = local * world + rank) proved on one GPU give the same proof
    equals_unsharded = None
    if rank == 0 and per_gb:
        full = []
        for sd in seeds:
            host = np.empty((1 << n, 4), dtype=np.uint64)
            for q in range(world):
                t = zk_amd.MultiLinearPolynomial.random(ctx, ns, sd, first_index=q << ns)
                host[q::world] = t.evaluation_slice()
                t.free()
            full.append(zk_amd.MultiLinearPolynomial.new(ctx, n, host))
            del host
        proof_u, ch_u = zk_amd.SumcheckProver(2).prove_partial(zk_amd.ProductPoly.new(full), claimed)
        equals_unsharded = all(np.array_equal(proof_u.round_polys, rp) and np.array_equal(ch_u, ch) for rp, ch in proofs.values())
        for q in full:
            q.free()

    # latency of the round's one collective: (D+1)*8 uint64 lanes, back to back on the stream
    lanes = torch.zeros(24, dtype=torch.int64, device=tdev)
    for _ in range(20):
        dist.all_reduce(lanes)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(200):
        dist.all_reduce(lanes)
    torch.cuda.synchronize()
    coll_us = (time.perf_counter() - t1) / 200 * 1e6
    # one verdict for every rank (rank 0 alone knows whether the sharded proof equals the unsharded one): a rank that left with
    # a different exit path would strand the others in the closing barrier
    okt = torch.tensor([int(verified_all and (equals_unsharded is not False))], device=tdev)
    dist.all_reduce(okt, op=dist.ReduceOp.MIN)
    ok = bool(okt.item())
    if rank == 0 and per_gb:
        key = f"n24_k2_d2_world{world}"
        best = min(per_gb, key=per_gb.get)
        ex["sharded_proof_verified"] = verified_all
        ex["sharded_proof_identical_on_all_ranks"] = same_all
        ex["sharded_proof_equals_unsharded_proof"] = equals_unsharded
        ex["parity_gate_sharded"] = {"proof_verified": verified_all, "equals_unsharded": equals_unsharded}
        if ok:   # a wrong proof has no timing worth publishing
            for gb, ms in per_gb.items():
                ex[f"sharded_sumcheck_ms_{key}_gather_below{gb}"] = ms
                ex[f"sharded_sumcheck_phases_ms_{key}_gather_below{gb}"] = phases_by_gb[gb]
            ex[f"sharded_sumcheck_ms_{key}"] = per_gb[best]
            ex["sharded_sumcheck_gather_below"] = best
            ex["sharded_sumcheck_phases_note"] = ("HIP-event breakdown of a separate run (each event record stalls the stream a few "
                                                  "us, so the phases add up to more than the timed run): max over ranks")
            result.setdefault("sumcheck_prover_wall_clock_ms", {})[key] = per_gb[best]
        else:
            ex["sharded_error"] = "sharded proof wrong (not verified / differs from the unsharded proof): timing withheld"
        ex["sharded_sumcheck_local_vars"] = ns
        ex["sharded_sumcheck_collective_rounds"] = {gb: max(ns - gb, 0) for gb in per_gb}
        ex["allreduce_24_lanes_latency_us"] = coll_us
    # four-step NTT across the ranks (one all-to-all): the 2^24-point transform, 2^(24 - log2 N) points per rank; forward then
    # inverse must give the input back on every rank
    try:
        xs = zk_amd.MultiLinearPolynomial.random(ctx, ns, 0x5EED0300, first_index=rank << ns)
        y = ntt_sharded(comm, xs, False)
        back = ntt_sharded(comm, y, True)
        rt = torch.tensor([int(bool(back == xs))], device=tdev)
        dist.all_reduce(rt, op=dist.ReduceOp.MIN)
        y.free(); back.free()
        ctx.synchronize()
        tn = []
        for it in range(5):
            dist.barrier()
            t1 = time.perf_counter()
            y = ntt_sharded(comm, xs, False)
            ctx.synchronize()
            tn.append(time.perf_counter() - t1)
            y.free()
        tt = torch.tensor([sorted(tn)[2]], dtype=torch.float64, device=tdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        if rank == 0:
            ex[f"sharded_ntt_ms_2p24_world{world}"] = float(tt.item()) * 1e3
            ex["sharded_ntt_roundtrip_exact"] = bool(rt.item())
    except Exception as e:
        if rank == 0:
            ex["sharded_ntt_error"] = repr(e)
    for q in shards:
        q.free()
    ctx.synchronize()
    comm.close()
    return 0 if ok else 5


def self_launch(n, argv, limit_s):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: run the SAME command under
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` (one fresh child process per GPU, RCCL between them) and
    relay rank 0's JSON line.  Called before torch / zk_amd are imported: this parent never initialises the GPU, never
    os.exec*s; it waits for the child job, ends its whole process group when the wall-clock limit passes, and returns the
    job's exit code (124 on the limit, 1 when the job printed no line)."""
    import signal
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True, start_new_session=True)
    try:
        out, _ = proc.communicate(timeout=limit_s)
        rc = proc.returncode
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGTERM)
            out, _ = proc.communicate(timeout=20)
        except subprocess.TimeoutExpired:
            os.killpg(proc.pid, signal.SIGKILL)
            out, _ = proc.communicate()
        except ProcessLookupError:
            out = ""
        sys.stderr.write(f"bench.py: the {n}-rank job passed its {limit_s:.0f} s limit and was ended\n")
        rc = 124
    lines = [ln for ln in (out or "").splitlines() if ln.startswith("{")]
    for ln in (out or "").splitlines():
        if not ln.startswith("{"):
            sys.stderr.write(ln + "\n")
    if os.environ.get("ZK_BENCH_TRACE_PARENT_IMPORTS") == "1":
        sys.stderr.write(f"bench.py: parent_imported_torch={'torch' in sys.modules} parent_imported_zk_amd={'zk_amd' in sys.modules}\n")
    if lines:
        print(lines[-1], flush=True)
    elif rc == 0:
        sys.stderr.write("bench.py: the child job ended without a JSON line\n")
        rc = 1
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--event-group", type=int, default=0,
                    help="launches per HIP-event bracket in the timed region (an event record stalls the stream ~4 us: 3 %% of a 2^24 "
                         "fold, 25 %% of a 2^21 shard's); 1 = one event per launch boundary; default: 8, one bracket for <= 32 steps")
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--no-pmc", action="store_true")
    ap.add_argument("--no-parity-gate", action="store_true",
                    help="skip the bit-compare of the timed fold and the timed prover inputs with the CPU oracle (profiler runs)")
    ap.add_argument("--prewarm-ms", type=float, default=250.0,
                    help="untimed folds before the W warm-up steps so that the shader clock has ramped (the parity gate leaves the "
                         "GPU idle for seconds); reported in the line")
    ap.add_argument("--launch-timeout", type=float, default=1500.0,
                    help="plain `bench.py --gpus N` (N > 1, no torchrun): wall-clock limit of the N-rank child job, seconds")
    args = ap.parse_args()

    if args.gpus < 1 or args.gpus & (args.gpus - 1):
        raise SystemExit("the table shards by index mod N: --gpus must be a power of two")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process has made no HIP call and imported neither torch nor zk_amd; it only
        # starts the N ranks as FRESH children and relays their line
        sys.exit(self_launch(args.gpus, sys.argv[1:], args.launch_timeout))

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import zk_amd

    field = zk_amd.BN254_FR
    # ZK_BENCH_REHEARSE=1: run the N > 1 code path with every rank on GPU 0 (gloo group, host-staged collectives): RCCL refuses two
    # ranks on one device, and a one-GPU box is all the development loop has.  The numbers of such a run mean nothing.
    rehearse = os.environ.get("ZK_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    tdev = "cpu" if rehearse else "cuda"
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ or os.environ.get("ZK_BENCH_FORCE_DIST") == "1":
        # launched by torch.distributed.run: one rank per GPU over RCCL (also taken at world == 1 so the path is testable)
        import torch.distributed as dist

        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    lw = world.bit_length() - 1
    local_vars = N_VARS - lw   # strong scaling: the 2^24 table is sharded by index mod world
    ctx = zk_amd.Context(field, local_rank)
    # The synthetic table is i.i.d. uniform, so a shard simply draws its own index range of the generator stream.
    table = zk_amd.MultiLinearPolynomial.random(ctx, local_vars, 0x5EED0000 + 24, first_index=rank << local_vars)
    out = zk_amd.MultiLinearPolynomial.alloc(ctx, local_vars - 1)
    tr = zk_amd.Transcript()
    tr.append(b"zk_amd bench challenge")
    r = tr.sample_field_element(field)       # a uniform challenge (not 0/1: generic path)
    ctx.synchronize()

    # ---- parity gate: BEFORE anything is timed (rank 0, N = 1: the oracle is test infrastructure, the checker only)
    gate, cpu_cache = None, {}
    if rank == 0 and world == 1 and not args.no_parity_gate:
        gate, cpu_cache = parity_gate(ctx, field, table, out, r)
        cpu_cache["from_gate"] = True
        if not all(gate.values()):
            print(json.dumps({"metric": "field-ops/sec (MLE fold, 2^24 evals, BN254 Fr) + sumcheck prover wall-clock", "value": None,
                              "error": "parity gate failed: the GPU path differs from the CPU oracle on the timed inputs; nothing timed",
                              "parity_gate": gate}))
            sys.exit(1)

    # the gate (seconds of host work) leaves the GPU at idle clocks: ramp them before the W warm-up steps
    prewarm_launches = 0
    t_pw = time.perf_counter()
    while (time.perf_counter() - t_pw) * 1e3 < args.prewarm_ms:
        for _ in range(64):
            table.fold_into(r, out)
        ctx.synchronize()
        prewarm_launches += 64
    for _ in range(args.warmup):
        table.fold_into(r, out)
    ctx.synchronize()

    barrier()
    t0 = time.perf_counter()
    # the K timed steps, bracketed by HIP events on the launch stream every `event_group` launches (zk_bench_fold_samples):
    # a sample is the average launch duration inside its bracket
    # (default: 8 launches per bracket; ONE bracket when there are at most 32 steps -- every event record stalls the stream ~4 us, which
    # is 0.5-1 us per launch of a short run's average and no part of the kernel)
    eg = args.event_group if args.event_group > 0 else (args.steps if args.steps <= 32 else 8)
    group = max(1, min(eg, args.steps))
    samples = table.bench_fold_samples(r, out, args.steps, group)
    ctx.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0   # this rank's K steps are complete; the MAX over ranks below is the job's time
    barrier()                       # the closing barrier + synchronize of the bracket (its collective latency is not one of the K steps)
    dt_with_barrier = time.perf_counter() - t0
    if dist is not None:
        tb = torch.tensor([dt_with_barrier], dtype=torch.float64, device=tdev)
        dist.all_reduce(tb, op=dist.ReduceOp.MAX)
        dt_with_barrier = float(tb.item())
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    weights = np.full(len(samples), group, dtype=np.float64)
    weights[-1] = args.steps - group * (len(samples) - 1)
    kernel_ms = float((samples * weights).sum() / args.steps)
    # median / min over >= 20 samples.  A short run (<= 32 steps) is ONE bracket = one sample: its spread statistics come
    # from a second block of 200 launches in brackets of 8 (25 samples), outside the driver-timed region, on the same buffers
    if len(samples) >= 20:
        stat_samples, stats_source = samples, f"the {len(samples)} brackets of the timed steps"
    else:
        stat_samples = table.bench_fold_samples(r, out, 200, 8)
        ctx.synchronize()
        stats_source = ("a second block of 200 launches in brackets of 8 (25 samples) right after the timed region; kernel_ms itself is "
                        f"the {len(samples)} bracket(s) of the timed steps")
    kernel_ms_median, kernel_ms_min = float(np.median(stat_samples)), float(stat_samples.min())

    alg_bytes_launch = 48 * (1 << local_vars)   # this rank's launch
    total_ops = FIELD_OPS_PER_FOLD * args.steps
    achieved_gbps = alg_bytes_launch / (kernel_ms * 1e-3) / 1e9
    result = {
        "metric": "field-ops/sec (MLE fold, 2^24 evals, BN254 Fr) + sumcheck prover wall-clock",
        "value": total_ops / dt,
        "unit": "field-ops/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "timed_region": ("barrier + synchronize; t0; the K steps; synchronize; t1 on every rank; MAX over ranks of t1 - t0; then the closing "
                         "barrier.  ms_per_step_incl_closing_barrier carries the collective's latency too (the same thing at N = 1)"),
        "ms_per_step_incl_closing_barrier": dt_with_barrier / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",   # total work is fixed (ONE 2^24 table) as N grows
        "vs_baseline": None,
        "dtype": "u256 (8 x u32 Montgomery limbs, integer)",
        "data": "synthetic" if not rehearse else "synthetic (REHEARSAL: all ranks on one GPU, not a measurement)",
        "config": {"workload": "mle_fold_msb of ONE 2^24-element BN254-Fr table (partial_evaluate(0,[r])), sharded by index mod n_gpus",
                   "n_vars": N_VARS, "field": "bn254_fr", "elements_per_gpu": 1 << local_vars,
                   "shard": "index mod n_gpus (no collective in the fold)"},
        "timed_region_s": dt,
        "library": __import__("zk_amd._lib", fromlist=["lib_info"]).lib_info(),
        "parity_gate": gate if gate is not None else "not run (--no-parity-gate or N > 1: see sharded_proof_* there)",
        "clock_prewarm": {"ms": args.prewarm_ms, "untimed_launches": prewarm_launches},
        "roofline": {"bound": "hbm", "achieved": achieved_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved_gbps / HBM_PEAK_GBPS, "traffic": None, "traffic_source": None,
                     "kernel": "zk::k_fold_msb", "kernel_ms": kernel_ms, "kernel_ms_median": kernel_ms_median,
                     "kernel_ms_min": kernel_ms_min, "launches_timed": int(args.steps), "launches_per_event_bracket": group,
                     "samples": int(len(samples)), "median_min_source": stats_source, "median_min_samples": int(len(stat_samples)),
                     "frac_at_median": alg_bytes_launch / (kernel_ms_median * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                     "algorithmic_bytes": alg_bytes_launch},
    }
    if world > 1:
        result["roofline"]["note"] = (f"per-rank launch on a 2^{local_vars}-element shard ({(32 << local_vars) >> 20} MiB: within the "
                                      "256 MiB Infinity Cache from 2^22 down, so not an HBM measurement)")

    if rank == 0 and not args.no_extra and world == 1:
        extra = {}
        try:
            def settle(ms=800.0):
                """These rows are single calls bracketed by a synchronise.  Right after idle seconds (the gate's oracle runs, a big download)
                the part goes through a clock transient in which the SAME call measures 144 or 204 us (tools/r06_a_rows_probe.py,
                profiles/r06_a_rows_probe.log); a stretch of untimed folds first puts it in the state a busy prover leaves it in."""
                t_s = time.perf_counter()
                while (time.perf_counter() - t_s) * 1e3 < ms:
                    for _ in range(64):
                        table.fold_into(r, out)
                    ctx.synchronize()

            # second half of the metric: sumcheck prover wall-clock, k=2, D=2: prove_partial (prover.rs:24-30) and prove
            # (prover.rs:15-20: the tables are serialised and absorbed first -- a serial host Keccak the reference mandates)
            for n in (20, 24):
                # the tables (and claimed sum) the parity gate proved with the oracle above: PROVER_SEED + n, factor f at f << n
                A = zk_amd.MultiLinearPolynomial.random(ctx, n, PROVER_SEED + n, 0)
                B = zk_amd.MultiLinearPolynomial.random(ctx, n, PROVER_SEED + n, 1 << n)
                pp = zk_amd.ProductPoly.new([A, B])
                s = pp.round_sums(1)
                claimed = zk_amd.fe_from_int(field, zk_amd.fe_to_int(field, s[0]) + zk_amd.fe_to_int(field, s[1]))
                prover = zk_amd.SumcheckProver(2)
                for _ in range(3):
                    prover.prove_partial(pp, claimed)   # warm (the rows below are latency-bound: their own calls are the warm-up)
                # the whole call under std::chrono inside the library (every launch, the transcript, the download of the
                # proof, the one host wait) -- what a compiled host sees; the same call through this Python binding beside it
                ms = sorted(zk_amd.bench_prove_partial(pp, 2, claimed, 11))
                extra[f"sumcheck_prove_partial_ms_n{n}_k2_d2"] = ms[len(ms) // 2]
                extra[f"sumcheck_prove_partial_ms_n{n}_k2_d2_min"] = ms[0]
                ts = []
                for _ in range(11):
                    ctx.synchronize()
                    t1 = time.perf_counter()
                    prover.prove_partial(pp, claimed)
                    ts.append(time.perf_counter() - t1)
                extra[f"sumcheck_prove_partial_via_python_ms_n{n}_k2_d2"] = sorted(ts)[len(ts) // 2] * 1e3
                ts = []
                for _ in range(3 if n == 20 else 1):
                    ctx.synchronize()
                    t1 = time.perf_counter()
                    prover.prove(pp, claimed)
                    ts.append(time.perf_counter() - t1)
                extra[f"sumcheck_prove_absorbing_ms_n{n}_k2_d2"] = sorted(ts)[len(ts) // 2] * 1e3
                A.free(); B.free()
            extra["sumcheck_prove_absorbing_note"] = ("prove = host Keccak-256 over k*2^n*32 table bytes (serial sponge, prover.rs:17) "
                                                      "with the device serialiser + copy of chunk i+1 overlapped, then prove_partial's rounds")
            # config[1]: the 2^20 fold (32 MiB table: Infinity-Cache resident, not an HBM measurement)
            t20 = zk_amd.MultiLinearPolynomial.random(ctx, 20, 0x5EED0014, 0)
            o20 = zk_amd.MultiLinearPolynomial.alloc(ctx, 19)
            extra["fold_2p20_us"] = t20.bench_fold(r, o20, 50) * 1e3
            t20.free(); o20.free()
            # the reference's own criterion bench (polynomial/benches/polynomial_evaluation.rs): evaluate at 18..21 variables
            tr2 = zk_amd.Transcript()
            tr2.append(b"zk_amd bench evaluate")
            for n in (18, 19, 20, 21):
                tn = zk_amd.MultiLinearPolynomial.random(ctx, n, 0x5EED0E00 + n, 0)
                pt = tr2.sample_n_field_elements(field, n)
                tn.evaluate(pt)   # warm: ONE call (profiles/r06_eval_r04_vs_head_ab.log: ten warm-up calls make the n = 21 median 1.5-2 us
                                  # slower on the round-4 library and on this one alike -- the round-5 driver "regression" was this row's warm-up)
                ms = sorted(zk_amd.bench_evaluate(tn, pt, 21))      # std::chrono around zk_mle_evaluate inside the library
                extra[f"evaluate_us_n{n}"] = ms[10] * 1e3
                extra[f"evaluate_us_n{n}_min"] = ms[0] * 1e3
                ts = []
                for _ in range(11):
                    ctx.synchronize()
                    t1 = time.perf_counter()
                    tn.evaluate(pt)
                    ts.append(time.perf_counter() - t1)
                extra[f"evaluate_via_python_us_n{n}"] = sorted(ts)[5] * 1e6
                tn.free()
            # ... the same bench on ITS field (ark_bls12_381::Fr, polynomial_evaluation.rs:12-15) and at the metric's table size
            ctx381 = zk_amd.Context(zk_amd.BLS12_381_FR, 0)
            for fld, cx, tag, sizes in ((zk_amd.BLS12_381_FR, ctx381, "bls12_381", (18, 19, 20, 21)), (field, ctx, "bn254", (22, 23, 24))):
                for n in sizes:
                    tn = zk_amd.MultiLinearPolynomial.random(cx, n, 0x5EED0E00 + n, 0)
                    pt = tr2.sample_n_field_elements(fld, n)
                    tn.evaluate(pt)   # warm (one call, as above)
                    ms = sorted(zk_amd.bench_evaluate(tn, pt, 21))
                    extra[f"evaluate_us_n{n}_{tag}"] = ms[10] * 1e3
                    extra[f"evaluate_us_n{n}_{tag}_min"] = ms[0] * 1e3
                    tn.free()
            ctx381.close()
            ev24 = extra["evaluate_us_n24_bn254"]
            t24 = zk_amd.MultiLinearPolynomial.random(ctx, 24, 0x5EED0E00 + 24, 0)
            pt24 = tr2.sample_n_field_elements(field, 24)
            dev24 = zk_amd.bench_evaluate_device(t24, pt24, 40) * 1e3    # us: HIP events around 40 back-to-back evaluates, no host wait
            if not args.no_parity_gate:   # the timed 2^24 evaluate against the oracle's n folds (evaluation_form.rs:83-89) on the same table and point
                from oracle import binding as orc
                gate_set(result, "evaluate_n24", np.array_equal(t24.evaluate(pt24), orc.mle_evaluate(field, 24, t24.evaluation_slice(), pt24)))
            t24.free()
            extra["evaluate_device_us_n24_bn254"] = dev24
            result["roofline_evaluate"] = {"bound": "hbm", "workload": "MultiLinearPolynomial::evaluate, 2^24 BN254-Fr elements: k_eval_stream over the low "
                                           "15 variables + one k_eval_low workgroup for the other 9",
                                           "algorithmic_bytes": 32 << 24, "achieved": (32 << 24) / (dev24 * 1e-6) / 1e9, "peak": HBM_PEAK_GBPS,
                                           "unit": "GB/s", "frac": (32 << 24) / (dev24 * 1e-6) / 1e9 / HBM_PEAK_GBPS, "device_us": dev24,
                                           "call_us": ev24, "frac_of_the_whole_call": (32 << 24) / (ev24 * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                                           "timing": "device_us: HIP events on the launch stream around 40 back-to-back evaluates (both launches of each, "
                                                     "zk_bench_evaluate_device); call_us: std::chrono around one zk_mle_evaluate incl. launch and completion",
                                           "profile": "profiles/r06_evaluate_kernel_stats_and_pmc.log (rocprofv3: the evaluate kernels, VALU and "
                                                      "FETCH_SIZE counters)"}
            # the (a)-rows that had no number: prod_reduce (product_poly.rs:66-74), partial_evaluate at general positions
            # (evaluation_form.rs:40-80), to_bytes (:97-103), to_evaluation_form (coefficient_form.rs:340-347) -- device-resident
            # calls bracketed by a synchronise, median of 9, against their algorithmic bytes
            rows = {}

            def timed(fn, reps=9):
                ts = []
                for _ in range(reps):
                    ctx.synchronize()
                    t1 = time.perf_counter()
                    keep = fn()
                    ctx.synchronize()
                    ts.append(time.perf_counter() - t1)
                    if keep is not None:
                        keep.free()
                return sorted(ts)[len(ts) // 2]

            def row(name, seconds, nbytes, what):
                rows[name] = {"us": seconds * 1e6, "algorithmic_bytes": nbytes, "GBps": nbytes / seconds / 1e9,
                              "hbm_frac": nbytes / seconds / 1e9 / HBM_PEAK_GBPS, "what": what}

            extra["rows_2p24_note"] = ("each row: median of 9 single calls through the allocating API, a synchronise on both sides; 0.8 s of untimed folds run "
                                       "before each group of rows (clock transient after idle: profiles/r06_a_rows_probe.log); the same before the batch, "
                                       "k = 3 n = 22, GKR and NTT rows, each of which follows seconds of host work (its gate's oracle runs)")
            n = 24
            tabs = [zk_amd.MultiLinearPolynomial.random(ctx, n, 0x5EED0F00 + f, 0) for f in range(3)]
            settle()
            for k in (2, 3):
                pk = zk_amd.ProductPoly.new(tabs[:k])
                pk.prod_reduce_device().free()
                row(f"prod_reduce_k{k}_2p24", timed(lambda: pk.prod_reduce_device()), (k + 1) * (32 << n), f"k_prod_reduce_run: {k} tables read, one written ({k - 1} carry-free table x table multiplications per element, fe_mul_tt)")
            asg = tr2.sample_n_field_elements(field, 1)
            if not args.no_parity_gate:
                # the timed rows against the oracle on the SAME 2^24 tables: prod_reduce k = 2 on every element, k = 3 on 64 runs of 1024
                # elements (the device generator is the oracle's, so a run of the inputs is regenerated, not downloaded), the three
                # general-position folds on every output
                from oracle import binding as orc
                host0 = tabs[0].evaluation_slice()
                host1 = tabs[1].evaluation_slice()
                pr2 = zk_amd.ProductPoly.new(tabs[:2]).prod_reduce_device()
                ok = np.array_equal(pr2.evaluation_slice(), orc.prod_reduce(field, n, [host0, host1]))
                pr2.free()
                del host1
                pr3 = zk_amd.ProductPoly.new(tabs[:3]).prod_reduce_device()
                got3 = pr3.evaluation_slice()
                pr3.free()
                rs = np.random.default_rng(0x9A7E)
                for first in rs.integers(0, (1 << n) - 1024, 64):
                    runs = [orc.fill_random(field, 0x5EED0F00 + f, 1024, first_index=int(first)) for f in range(3)]
                    ok = ok and np.array_equal(got3[int(first):int(first) + 1024], orc.prod_reduce(field, 10, runs))
                del got3
                gate_set(result, "prod_reduce_2p24", ok)
                okf = True
                for v in (1, n // 2, n - 1):
                    pe = tabs[0].partial_evaluate(v, asg)
                    okf = okf and np.array_equal(pe.evaluation_slice(), orc.mle_partial_evaluate(field, n, host0, v, asg))
                    pe.free()
                gate_set(result, "partial_evaluate_2p24_general_positions", okf)
                del host0
            settle()
            for v in (1, n // 2, n - 1):
                tabs[0].partial_evaluate(v, asg).free()
                row(f"partial_evaluate_2p24_var{v}", timed(lambda: tabs[0].partial_evaluate(v, asg)), 48 << n,
                    f"{'k_fold_run' if n - 1 - v >= 6 else 'k_fold_low'} at initial_var = {v} (index bit {n - 1 - v}), through the allocating call: 2^24 read, 2^23 written")
            tabs[0].partial_evaluate(0, asg).free()
            row("partial_evaluate_2p24_var0", timed(lambda: tabs[0].partial_evaluate(0, asg)), 48 << n, "k_fold_msb through the allocating call")
            t1 = time.perf_counter()
            buf = tabs[0].to_bytes_array()          # fresh destination: its pages are faulted in by the copy-out
            dt_fresh = time.perf_counter() - t1
            t1 = time.perf_counter()
            tabs[0].to_bytes_array(buf)             # the same destination again: pages already mapped
            dt_warm = time.perf_counter() - t1
            del buf
            rows["to_bytes_2p24"] = {"ms_fresh_destination": dt_fresh * 1e3, "ms_mapped_destination": dt_warm * 1e3,
                                     "host_GBps_mapped": (32 << n) / dt_warm / 1e9,
                                     "what": "k_to_bytes in 16-MiB chunks -> pinned staging (two buffers) -> the caller's buffer on up to four host "
                                             "threads; PCIe / host-memory bound, not a device measurement"}
            for q in tabs:
                q.free()
            rng_c = np.random.default_rng(0xC0EF)
            keys = rng_c.integers(0, 1 << n, 1 << 10, dtype=np.uint64)   # few terms: the binding's term handling stays out of the timing
            coeffs = zk_amd.fe_from_ints(field, [int(x) for x in rng_c.integers(1, 1 << 62, 1 << 10)])
            cf = zk_amd.CoeffMultilinearPolynomial.new_with_coefficient(field, n, {int(k_): c_ for k_, c_ in zip(keys, coeffs)})
            def zeta_gate(cpoly, key):
                """coefficient_form.rs:340-347 by its definition on 64 table entries: T[idx] = sum of the coefficients whose variable set is
                contained in the point's (key bit v <-> variable v <-> index bit n-1-v), big-int arithmetic"""
                tab_ = cpoly.to_evaluation_form(ctx)
                hostt = tab_.evaluation_slice()
                tab_.free()
                p_ = zk_amd.modulus(field)
                ks = np.array(sorted(cpoly.coefficients), dtype=np.uint64)
                cs = [zk_amd.fe_to_int(field, cpoly.coefficients[int(k_)]) for k_ in ks]
                rq = np.random.default_rng(0x2E7A)
                good = True
                for idx in [0, (1 << n) - 1] + [int(x) for x in rq.integers(0, 1 << n, 62)]:
                    mask = int("{:0{w}b}".format(idx, w=n)[::-1], 2)          # variables set to one at this index
                    sel = (ks & np.uint64(~mask & ((1 << n) - 1))) == 0
                    want_ = sum(c_ for c_, s_ in zip(cs, sel) if s_) % p_
                    good = good and zk_amd.fe_to_int(field, hostt[idx]) == want_
                gate_set(result, key, good)

            if not args.no_parity_gate:
                zeta_gate(cf, "to_evaluation_form_2p24_1k_terms")
            cf.to_evaluation_form(ctx).free()
            ts = []
            for _ in range(3):
                ctx.synchronize()
                t1 = time.perf_counter()
                ev_t = cf.to_evaluation_form(ctx)
                ctx.synchronize()
                ts.append(time.perf_counter() - t1)
                ev_t.free()
            row("coeff_to_evaluation_2p24_1k_terms", sorted(ts)[1], 32 << n,
                "k_zeta_first (11 index bits per 2^11-entry LDS tile, built from the sorted term list: the table is written, never read) + "
                "k_zeta_tile x 2 (7 and 6 index bits, one read and one write each); algorithmic bytes = the table written once (32 * 2^24); "
                "the three passes move 5 x that (PMC: profiles/r05_zeta_ab_and_pmc.log)")
            # ... and with 2^16 terms (a list this long is uploaded as given and ordered on the device: zeta_sort.hip)
            keys16 = np.unique(rng_c.integers(0, 1 << n, 1 << 16, dtype=np.uint64))
            co16 = zk_amd.MultiLinearPolynomial.random(ctx, 16, 0xC0EF16, 0).evaluation_slice()[:len(keys16)]
            cf16 = zk_amd.CoeffMultilinearPolynomial.new_with_coefficient(field, n, {int(k_): c_ for k_, c_ in zip(keys16, co16)})
            if not args.no_parity_gate:
                zeta_gate(cf16, "to_evaluation_form_2p24_64k_terms")
            cf16.to_evaluation_form(ctx).free()
            ts = []
            for _ in range(5):
                ctx.synchronize()
                t1 = time.perf_counter()
                ev_t = cf16.to_evaluation_form(ctx)
                ctx.synchronize()
                ts.append(time.perf_counter() - t1)
                ev_t.free()
            row("coeff_to_evaluation_2p24_64k_terms", sorted(ts)[2], 32 << n,
                f"{len(keys16)} terms uploaded as given (2.6 MB from pageable host memory), ordered on the device (k_term_indices + rocPRIM radix sort), then the "
                "same three passes; round 5: 3.53 ms with the list sorted and merged on the host")
            extra["rows_2p24"] = rows
            # config[3]: GKR-shaped load -- no gkr crate exists in the reference; what it would call is
            # prove_partial on one ProductPoly per layer: depth 8, width 2^20, product of 3 MLEs, degree 3
            layers = []
            for layer in range(8):
                layers.append(zk_amd.ProductPoly.new([zk_amd.MultiLinearPolynomial.random(ctx, 20, 0x6000 + 16 * layer + f, 0)
                                                      for f in range(3)]))
            claimed = zk_amd.fe_from_int(field, 7)   # timing only
            p3 = zk_amd.SumcheckProver(3)
            for pp in layers[:2]:
                p3.prove_partial(pp, claimed)
            ctx.synchronize()
            t1 = time.perf_counter()
            seq_proofs = [p3.prove_partial(pp, claimed) for pp in layers]
            ctx.synchronize()
            extra["gkr_shaped_depth8_width2p20_k3_d3_ms"] = (time.perf_counter() - t1) * 1e3
            # the same eight INDEPENDENT proofs in flight at once ("8 independent layers"): zk_sumcheck_prove_batch -- one launch
            # per round for all eight, each proof with its own transcript.  Gate first: every batched proof equals the back-to-back one bit for
            # bit, and layers 0 and 7 equal the faithful oracle's proofs of those tables
            sums8 = np.stack([claimed] * 8)
            bat = p3.prove_partial_batch(layers, sums8)
            ok = all(np.array_equal(a[0].round_polys, b[0].round_polys) and np.array_equal(a[1], b[1]) for a, b in zip(seq_proofs, bat))
            if not args.no_parity_gate:
                from oracle import binding as orc
                for layer in (0, 7):
                    w_rp, w_ch = orc.sumcheck_prove(field, 20, [q.evaluation_slice() for q in layers[layer].polynomials], 3, claimed, False)
                    ok = ok and bool(np.array_equal(bat[layer][0].round_polys, w_rp) and np.array_equal(bat[layer][1], w_ch))
            if isinstance(result.get("parity_gate"), dict):
                result["parity_gate"]["batch_8x_k3_n20"] = bool(ok)
            if ok:
                settle()   # (the two oracle proofs above are ~1.5 s of host work)
                ts = []
                for _ in range(9):
                    ctx.synchronize()
                    t1 = time.perf_counter()
                    p3.prove_partial_batch(layers, sums8)
                    ts.append(time.perf_counter() - t1)
                extra["gkr_shaped_depth8_width2p20_k3_d3_concurrent_ms"] = sorted(ts)[4] * 1e3
                extra["gkr_shaped_depth8_width2p20_k3_d3_concurrent_ms_min"] = sorted(ts)[0] * 1e3
                merged, replayed = zk_amd.batch_last_stats()
                extra["gkr_shaped_concurrent_note"] = (f"zk_sumcheck_prove_batch: 8 proofs, {merged} launches in all ({replayed} replayed proof by proof); "
                                                       "the back-to-back row above is the same eight proofs one call after the other")
            else:
                extra["gkr_shaped_depth8_width2p20_k3_d3_concurrent_ms"] = None
            # the same comparison for eight independent (k = 2, D = 2, n = 20) proofs (config[1]'s shape)
            l2 = [zk_amd.ProductPoly.new(pp.polynomials[:2]) for pp in layers]
            p2 = zk_amd.SumcheckProver(2)
            seq2 = [p2.prove_partial(pp, claimed) for pp in l2]
            bat2 = p2.prove_partial_batch(l2, sums8)
            ok2 = all(np.array_equal(a[0].round_polys, b[0].round_polys) and np.array_equal(a[1], b[1]) for a, b in zip(seq2, bat2))
            if isinstance(result.get("parity_gate"), dict):
                result["parity_gate"]["batch_8x_k2_n20_equals_back_to_back"] = bool(ok2)
            if ok2:
                ts_seq, ts_bat = [], []
                for _ in range(9):
                    ctx.synchronize()
                    t1 = time.perf_counter()
                    for pp in l2:
                        p2.prove_partial(pp, claimed)
                    ts_seq.append(time.perf_counter() - t1)
                    ctx.synchronize()
                    t1 = time.perf_counter()
                    p2.prove_partial_batch(l2, sums8)
                    ts_bat.append(time.perf_counter() - t1)
                extra["eight_independent_proofs_n20_k2_d2_back_to_back_ms"] = sorted(ts_seq)[4] * 1e3
                extra["eight_independent_proofs_n20_k2_d2_concurrent_ms"] = sorted(ts_bat)[4] * 1e3
            for pp in layers:
                for q in pp.polynomials:
                    q.free()
            # one BIG three-table proof (k = 3, D = 3, n = 22): the size where the fused rounds of three tables run on the LDS-DMA kernels
            # from the first fold on.  Gated like every row: the timed proof against the faithful oracle's on the same tables.
            t3 = [zk_amd.MultiLinearPolynomial.random(ctx, 22, 0x7000 + f, 0) for f in range(3)]
            pp3 = zk_amd.ProductPoly.new(t3)
            got3 = p3.prove_partial(pp3, claimed)
            if not args.no_parity_gate:
                from oracle import binding as orc
                w_rp, w_ch = orc.sumcheck_prove(field, 22, [q.evaluation_slice() for q in t3], 3, claimed, False)
                gate_set(result, "prove_k3_n22", np.array_equal(got3[0].round_polys, w_rp) and np.array_equal(got3[1], w_ch))
            settle()
            ms3 = sorted(zk_amd.bench_prove_partial(pp3, 3, claimed, 11))
            extra["sumcheck_prove_partial_ms_n22_k3_d3"] = ms3[len(ms3) // 2]
            extra["sumcheck_prove_partial_ms_n22_k3_d3_min"] = ms3[0]
            for q in t3:
                q.free()
            # config[3] as an actual layered circuit: depth 8, width 2^20, random add/mul gates with random wiring, proved by
            # the GKR-shaped driver (zk_gkr_prove: circuit evaluation + per layer two sum-of-products prove_partial calls)
            try:
                from zk_amd import gkr
                rng = np.random.default_rng(0x6B72)
                w = 20
                circ = gkr.Circuit(ctx)
                for _ in range(8):
                    circ.add_layer(w, w, rng.integers(0, 2, 1 << w, dtype=np.uint8), rng.integers(0, 1 << w, 1 << w, dtype=np.uint32),
                                   rng.integers(0, 1 << w, 1 << w, dtype=np.uint32))
                xin = zk_amd.MultiLinearPolynomial.random(ctx, w, 0x6B72, 0)
                seed = bytes(range(32))
                out_t, proof = gkr.gkr_prove(circ, xin, seed)   # warm
                settle()   # (the circuit above was drawn and uploaded by the host)
                ts = []
                for _ in range(5):
                    ctx.synchronize()
                    t1 = time.perf_counter()
                    out_t, proof = gkr.gkr_prove(circ, xin, seed)
                    ts.append(time.perf_counter() - t1)
                extra["gkr_depth8_width2p20_addmul_prove_ms"] = sorted(ts)[2] * 1e3
                ok = gkr.gkr_verify(circ, xin, out_t, seed, proof)   # warm (the first call sizes the pinned staging block)
                ts = []
                for _ in range(5):
                    ctx.synchronize()
                    t1 = time.perf_counter()
                    ok = gkr.gkr_verify(circ, xin, out_t, seed, proof) and ok
                    ts.append(time.perf_counter() - t1)
                extra["gkr_depth8_width2p20_addmul_verify_ms"] = sorted(ts)[2] * 1e3
                extra["gkr_depth8_width2p20_addmul_verified"] = bool(ok)
                extra["gkr_proof_bytes"] = int(proof.size * 8)
                out_t.free(); circ.free()
                # the same depth and width with STRUCTURED wiring (butterfly: gate z reads z and z xor 2^(layer)): the bookkeeping
                # gathers of the random circuit above are its worst case (every E[z], W[y] access is a random 32-byte read)
                circ = gkr.Circuit(ctx)
                zidx = np.arange(1 << w, dtype=np.uint32)
                for layer in range(8):
                    circ.add_layer(w, w, ((zidx >> 1) & 1).astype(np.uint8), zidx, zidx ^ np.uint32(1 << (layer + 3)))
                out_t, proof = gkr.gkr_prove(circ, xin, seed)
                ts = []
                for _ in range(3):
                    ctx.synchronize()
                    t1 = time.perf_counter()
                    out_t, proof = gkr.gkr_prove(circ, xin, seed)
                    ts.append(time.perf_counter() - t1)
                extra["gkr_depth8_width2p20_butterfly_prove_ms"] = sorted(ts)[1] * 1e3
                extra["gkr_depth8_width2p20_butterfly_verified"] = bool(gkr.gkr_verify(circ, xin, out_t, seed, proof))
                out_t.free(); xin.free(); circ.free()
            except Exception as e:
                extra["gkr_error"] = repr(e)
            # config[4]: 2^24-point NTT (3 LDS-staged passes), device resident
            x = zk_amd.MultiLinearPolynomial.random(ctx, 24, 0x5EED0005, 0)
            y = zk_amd.MultiLinearPolynomial.alloc(ctx, 24)
            if not args.no_parity_gate:
                # the timed transform: three forward and two inverse outputs against the definition X[k] = sum_j x[j] w^(jk) evaluated by the
                # oracle (fft/src/lib.rs:39-45; ~2.5 s each on one host core), and ifft(fft(x)) == x on the device (fft/src/lib.rs:78-82)
                from oracle import binding as orc
                xs = x.evaluation_slice()
                zk_amd.ntt(ctx, x, y, False)
                Xs = y.evaluation_slice()
                okn = all(np.array_equal(Xs[k_], orc.dft_point(field, xs, k_)) for k_ in (1, (1 << 23) + 12345, (1 << 24) - 1))
                back = zk_amd.MultiLinearPolynomial.alloc(ctx, 24)
                zk_amd.ntt(ctx, y, back, True)
                okn = okn and back == x
                zk_amd.ntt(ctx, x, back, True)
                Ys = back.evaluation_slice()
                okn = okn and all(np.array_equal(Ys[k_], orc.dft_point(field, xs, k_, inverse=True)) for k_ in (7, (1 << 22) + 99))
                back.free()
                del xs, Xs, Ys
                gate_set(result, "ntt_2p24", okn)
            settle()   # (the gate above is ~12 s of host work)
            extra["ntt_2p24_ms"] = zk_amd.bench_ntt(ctx, x, y, False, 10)
            extra["intt_2p24_ms"] = zk_amd.bench_ntt(ctx, x, y, True, 10)
            x.free(); y.free()
            extra["modmul_per_s_register_resident"] = ctx.bench_modmul(2000)
            extra["modmul29_per_s_register_resident"] = ctx.bench_modmul(2000, 1)
            extra["copy_gbps_1GiB"] = ctx.bench_copy(1 << 30, 10)
            # second roofline entry: the NTT is integer-ALU bound -- butterflies/s against the measured
            # register-resident multiply peak of this device (one modular multiply per butterfly, 12 * 2^24 of them)
            butterflies = (1 << 23) * 24
            peak_mm = extra["modmul29_per_s_register_resident"]
            ach_mm = butterflies / (extra["ntt_2p24_ms"] * 1e-3)
            result["roofline_ntt"] = {"bound": "integer-alu", "workload": "ntt 2^24 BN254-Fr forward (3 passes)",
                                      "achieved": ach_mm, "peak": peak_mm, "unit": "modmul/s", "frac": ach_mm / peak_mm,
                                      "peak_source": "zk_bench_modmul variant 1 (prepared-operand 29-bit multiply), this run",
                                      "hbm_frac_one_pass_bytes": (2 * 32 * (1 << 24)) / (extra["ntt_2p24_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS}
        except Exception as e:  # extras never invalidate the headline line
            extra["error"] = repr(e)
        result["extra"] = extra
        # second half of the metric, surfaced next to `value` (value itself is the fold's field-ops/s)
        result["sumcheck_prover_wall_clock_ms"] = {k.replace("sumcheck_prove_partial_ms_", ""): v for k, v in extra.items()
                                                   if k.startswith("sumcheck_prove_partial_ms_")}

    exit_code = 0
    if dist is not None and not args.no_extra:
        # BASELINE config 3: the n = 24 prover over the table sharded by index mod world: every rank holds 2^(24 - log2 N)
        # elements per factor; one RCCL all-reduce of (D+1)*8 lanes per round and one all-gather for the tail, all enqueued by
        # zk_shard_prover_run on one stream.  A stalled collective (a rank that died) must not hang the job nor pass for a
        # success: the watchdog prints the headline line with the error and ends EVERY rank with a non-zero code.
        import threading

        def _bail():
            if rank == 0:
                result.setdefault("extra", {})["sharded_error"] = "sharded-prover leg timed out (watchdog): a collective stalled"
                print(json.dumps(result), flush=True)
            os._exit(3)

        watchdog = threading.Timer(240.0, _bail)
        watchdog.daemon = True
        watchdog.start()
        try:
            exit_code = sharded_leg(args, ctx, field, dist, torch, rank, world, local_vars, tdev, rehearse, result)
        except Exception as e:
            exit_code = 4
            if rank == 0:
                result.setdefault("extra", {})["sharded_error"] = repr(e)
        finally:
            watchdog.cancel()

    if rank == 0 and world == 1 and not args.no_pmc:
        traffic, note = pmc_traffic()
        result["roofline"]["traffic"] = traffic
        result["roofline"]["traffic_source"] = note
        if traffic:
            result["roofline"]["traffic_over_algorithmic"] = traffic / ALG_BYTES_PER_FOLD

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(field, cpu_cache)
        if "fold_2p24_faithful_ms" in cpu_cache:
            result["cpu_baseline"]["fold_2p24_faithful_ms"] = cpu_cache["fold_2p24_faithful_ms"]
        if "to_bytes_2p24_faithful_ms" in cpu_cache:
            result["cpu_baseline"]["to_bytes_2p24_faithful_ms"] = cpu_cache["to_bytes_2p24_faithful_ms"]
        for ns in (20, 24):   # `prove` with the tables absorbed: the parity gate's oracle runs (faithful, one thread)
            if isinstance(cpu_cache.get(ns), dict) and "faithful_absorbing_ms" in cpu_cache[ns]:
                result["cpu_baseline"][f"sumcheck_prove_absorbing_ms_n{ns}_k2_d2"] = cpu_cache[ns]["faithful_absorbing_ms"]

    # a row whose inputs failed their check is not a measurement: the line says so and the run ends non-zero
    if rank == 0 and isinstance(result.get("parity_gate"), dict) and not all(result["parity_gate"].values()):
        failed = sorted(k for k, v in result["parity_gate"].items() if not v)
        result["error"] = f"parity gate failed for {failed}: the GPU path differs from the CPU oracle on those timed inputs"
        result["value"] = None
        exit_code = exit_code or 5
    if dist is not None:
        if exit_code == 0:
            dist.barrier()
            dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)
    if exit_code:
        os._exit(exit_code)   # a failed sharded leg: every rank ends non-zero (no barrier: a peer may be gone)


if __name__ == "__main__":
    main()
