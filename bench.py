#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X hot path (BASELINE.json metric).

Metric: field-ops/sec of the MLE fold (one sumcheck-round restriction, `partial_evaluate(0, [r])`,
polynomial/src/multilinear/evaluation_form.rs:40-80) on a 2^24-element BN254-Fr table, 3 field ops per pair
(1 mul + 2 sub, evaluation_form.rs:68).  A "step" is one fold of one 2^24 table per GPU: read 512 MiB, write 256 MiB
(algorithmic bytes 48 * 2^24 = 805,306,368 B).  Inputs are synthetic (device-generated, resident in HBM
before the timed region).  At N > 1 every rank folds its own 2^24-element shard of a 2^(24+log2 N)-variable table
(suffix shard): the fold has no exchange step, so there is no data-path collective ("weak" scaling).
The sumcheck prover wall-clock (second half of the metric) is reported in `extra`.

Contract: `python bench.py --gpus N --steps K --warmup W`; N > 1 is launched by torch.distributed.run, one rank per
GPU.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_VARS = 24
ALG_BYTES_PER_FOLD = 48 * (1 << N_VARS)
FIELD_OPS_PER_FOLD = 3 * (1 << (N_VARS - 1))
HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec


def cpu_baseline(field):
    """Reference-faithful CPU restatement (oracle, 1 thread) of the same fold on a bounded sample."""
    import numpy as np

    from oracle import binding as orc

    n = 21   # 2^21 elements = 64 MiB: same streaming pattern, bounded run time
    tab = orc.fill_random(field, 0x5EED0000 + 24, 1 << n)
    r = orc.fill_random(field, 0xC4A11, 1)
    t0 = time.perf_counter()
    reps = 0
    while True:
        out = orc.mle_partial_evaluate(field, n, tab, 0, r)
        reps += 1
        if time.perf_counter() - t0 > 10.0:
            break
    dt = time.perf_counter() - t0
    ops = 3 * (1 << (n - 1)) * reps
    # "optimised CPU" row: same fold, fused / out of place, OpenMP over all host cores
    # the GPU box gives a one-GPU job a 16-core share even though it reports every core of the host
    try:
        ncores = len(os.sched_getaffinity(0))
    except AttributeError:
        ncores = os.cpu_count() or 1
    ncores = max(1, min(ncores, 16))
    t1 = time.perf_counter()
    reps_par = 0
    while True:
        out_par, used = orc.fold_msb_parallel(field, n, tab, r[0], threads=ncores)
        reps_par += 1
        if time.perf_counter() - t1 > 5.0:
            break
    dt_par = time.perf_counter() - t1
    assert np.array_equal(out_par, out), "parallel CPU fold differs from the faithful fold"
    # the other half of the metric on the CPU: reference-faithful prover ((D+2)*k folds + (D+1) prod_reduce per round)
    cpu_prove = {}
    for ns in (12, 16, 20):   # configs[0] and configs[1] of BASELINE.json at full size, 16 for continuity
        tabs = [orc.fill_random(field, 0x5EED0000 + ns + f, 1 << ns) for f in range(2)]
        t1 = time.perf_counter()
        orc.sumcheck_prove(field, ns, tabs, 2, orc.fill_random(field, 5, 1)[0], False)
        cpu_prove[ns] = (time.perf_counter() - t1) * 1e3
    # the reference's own criterion bench on the CPU restatement: evaluate at 20 variables (n clone + fold + copy steps)
    t20 = orc.fill_random(field, 0x5EED0E00 + 20, 1 << 20)
    pt20 = orc.fill_random(field, 0xE7A1, 20)
    t1 = time.perf_counter()
    orc.mle_evaluate(field, 20, t20, pt20)
    cpu_eval_ms = (time.perf_counter() - t1) * 1e3
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
        "optimised": {"value": 3 * (1 << (n - 1)) * reps_par / dt_par, "unit": "field-ops/s", "cores": used,
                      "sample": f"{reps_par} fused out-of-place folds of the same table, OpenMP, {dt_par:.1f} s"},
    }, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch N > 1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")

    import zk_amd

    field = zk_amd.BN254_FR
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ or os.environ.get("ZK_BENCH_FORCE_DIST") == "1":
        # launched by torch.distributed.run: one rank per GPU over RCCL (also taken at world == 1 so the path is testable)
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    ctx = zk_amd.Context(field, local_rank)
    # rank g holds one 2^24-element shard of the global table (shard = index mod world).  The synthetic
    # table is i.i.d. uniform, so each shard simply draws its own index range of the generator stream.
    table = zk_amd.MultiLinearPolynomial.random(ctx, N_VARS, 0x5EED0000 + 24, first_index=rank << N_VARS)
    out = zk_amd.MultiLinearPolynomial.alloc(ctx, N_VARS - 1)
    tr = zk_amd.Transcript()
    tr.append(b"zk_amd bench challenge")
    r = tr.sample_field_element(field)       # a uniform challenge (not 0/1: generic path)
    ctx.synchronize()

    for _ in range(args.warmup):
        table.fold_into(r, out)
    ctx.synchronize()

    barrier()
    t0 = time.perf_counter()
    # the K timed steps are bracketed by HIP events on the launch stream inside zk_bench_fold
    kernel_ms = table.bench_fold(r, out, args.steps)
    ctx.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # HBM traffic per k_fold launch from the rocprofv3 PMC passes of this same command (FETCH_SIZE x2 on gfx950 +
    # WRITE_SIZE, tools/summarize_prof.py); bench.py cannot collect PMC counters itself.
    traffic, traffic_src = None, None
    try:
        import glob

        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
        if cands:
            tj = json.load(open(cands[-1]))
            if "zk::k_fold_msb" in tj:
                traffic, traffic_src = tj["zk::k_fold_msb"]["hbm_bytes_per_launch"], os.path.relpath(cands[-1], ROOT)
    except Exception:
        pass

    total_ops = FIELD_OPS_PER_FOLD * args.steps * world
    achieved_gbps = ALG_BYTES_PER_FOLD / (kernel_ms * 1e-3) / 1e9
    result = {
        "metric": "field-ops/sec (MLE fold, 2^24 evals, BN254 Fr) + sumcheck prover wall-clock",
        "value": total_ops / dt,
        "unit": "field-ops/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u256 (8 x u32 Montgomery limbs, integer)",
        "data": "synthetic",
        "config": {"workload": "mle_fold_msb 2^24 BN254-Fr elements per GPU (partial_evaluate(0,[r]))",
                   "n_vars": N_VARS, "field": "bn254_fr", "shard": "index mod n_gpus (no collective in the fold)"},
        "roofline": {"bound": "hbm", "achieved": achieved_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved_gbps / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": "zk::k_fold_msb", "kernel_ms": kernel_ms, "algorithmic_bytes": ALG_BYTES_PER_FOLD},
    }

    if rank == 0 and not args.no_extra:
        extra = {}
        try:
            # second half of the metric: sumcheck prover wall-clock (prove_partial semantics), k=2, D=2
            for n in (20, 24):
                A = zk_amd.MultiLinearPolynomial.random(ctx, n, 0x5EED0000 + n, 0)
                B = zk_amd.MultiLinearPolynomial.random(ctx, n, 0x5EED0000 + n, 1 << n)
                pp = zk_amd.ProductPoly.new([A, B])
                s = pp.round_sums(1)
                claimed = zk_amd.fe_from_int(field, zk_amd.fe_to_int(field, s[0]) + zk_amd.fe_to_int(field, s[1]))
                prover = zk_amd.SumcheckProver(2)
                prover.prove_partial(pp, claimed)   # warm
                ts = []
                for _ in range(5):
                    ctx.synchronize()
                    t1 = time.perf_counter()
                    prover.prove_partial(pp, claimed)
                    ts.append(time.perf_counter() - t1)
                extra[f"sumcheck_prove_partial_ms_n{n}_k2_d2"] = sorted(ts)[len(ts) // 2] * 1e3
                A.free(); B.free()
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
                tn.evaluate(pt)
                ts = []
                for _ in range(11):
                    ctx.synchronize()
                    t1 = time.perf_counter()
                    tn.evaluate(pt)
                    ts.append(time.perf_counter() - t1)
                extra[f"evaluate_us_n{n}"] = sorted(ts)[5] * 1e6
                tn.free()
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
            for pp in layers:
                p3.prove_partial(pp, claimed)
            ctx.synchronize()
            extra["gkr_shaped_depth8_width2p20_k3_d3_ms"] = (time.perf_counter() - t1) * 1e3
            for pp in layers:
                for q in pp.polynomials:
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
                out, proof = gkr.gkr_prove(circ, xin, seed)   # warm
                ts = []
                for _ in range(3):
                    ctx.synchronize()
                    t1 = time.perf_counter()
                    out, proof = gkr.gkr_prove(circ, xin, seed)
                    ts.append(time.perf_counter() - t1)
                extra["gkr_depth8_width2p20_addmul_prove_ms"] = sorted(ts)[1] * 1e3
                t1 = time.perf_counter()
                ok = gkr.gkr_verify(circ, xin, out, seed, proof)
                extra["gkr_depth8_width2p20_addmul_verify_ms"] = (time.perf_counter() - t1) * 1e3
                extra["gkr_depth8_width2p20_addmul_verified"] = bool(ok)
                extra["gkr_proof_bytes"] = int(proof.size * 8)
                out.free(); circ.free()
                # the same depth and width with STRUCTURED wiring (butterfly: gate z reads z and z xor 2^(layer)): the bookkeeping
                # gathers of the random circuit above are its worst case (every E[z], W[y] access is a random 32-byte read)
                circ = gkr.Circuit(ctx)
                zidx = np.arange(1 << w, dtype=np.uint32)
                for layer in range(8):
                    circ.add_layer(w, w, ((zidx >> 1) & 1).astype(np.uint8), zidx, zidx ^ np.uint32(1 << (layer + 3)))
                out, proof = gkr.gkr_prove(circ, xin, seed)
                ts = []
                for _ in range(3):
                    ctx.synchronize()
                    t1 = time.perf_counter()
                    out, proof = gkr.gkr_prove(circ, xin, seed)
                    ts.append(time.perf_counter() - t1)
                extra["gkr_depth8_width2p20_butterfly_prove_ms"] = sorted(ts)[1] * 1e3
                extra["gkr_depth8_width2p20_butterfly_verified"] = bool(gkr.gkr_verify(circ, xin, out, seed, proof))
                out.free(); xin.free(); circ.free()
            except Exception as e:
                extra["gkr_error"] = repr(e)
            # config[4]: 2^24-point NTT (3 LDS-staged passes), device resident
            x = zk_amd.MultiLinearPolynomial.random(ctx, 24, 0x5EED0005, 0)
            y = zk_amd.MultiLinearPolynomial.alloc(ctx, 24)
            extra["ntt_2p24_ms"] = zk_amd.bench_ntt(ctx, x, y, False, 5)
            extra["intt_2p24_ms"] = zk_amd.bench_ntt(ctx, x, y, True, 5)
            x.free(); y.free()
            extra["modmul_per_s_register_resident"] = ctx.bench_modmul(2000)
            extra["modmul29_per_s_register_resident"] = ctx.bench_modmul(2000, 1)
            extra["copy_gbps_1GiB"] = ctx.bench_copy(1 << 30, 10)
        except Exception as e:  # extras never invalidate the headline line
            extra["error"] = repr(e)
        result["extra"] = extra
        # second half of the metric, surfaced next to `value` (value itself is the fold's field-ops/s)
        result["sumcheck_prover_wall_clock_ms"] = {k.replace("sumcheck_prove_partial_ms_", ""): v for k, v in extra.items()
                                                   if k.startswith("sumcheck_prove_partial_ms_")}

    if dist is not None and not args.no_extra:
        # the prover over a table sharded by index mod world: every rank holds a 2^22-element shard per
        # factor; one all-reduce of (D+1)*8 int64 lanes per round, one all-gather for the tail.
        # This secondary measurement must never cost the headline line: if it stalls (a collective waiting on a rank
        # that failed), a watchdog prints the line without it and ends every rank.
        import threading

        def _bail():
            if rank == 0:
                result.setdefault("extra", {})["sharded_error"] = "sharded-prover extra timed out (watchdog)"
                print(json.dumps(result), flush=True)
            os._exit(0)

        watchdog = threading.Timer(120.0, _bail)
        watchdog.daemon = True
        watchdog.start()
        try:
            from zk_amd.distributed import GpuShardBackend, ShardedSumcheckProver

            ns = 22
            A = zk_amd.MultiLinearPolynomial.random(ctx, ns, 0x5EED0100, first_index=rank << ns)
            B = zk_amd.MultiLinearPolynomial.random(ctx, ns, 0x5EED0200, first_index=rank << ns)
            claimed = zk_amd.fe_from_int(field, 12345)   # timing only: the proof need not verify
            ts = []
            for it in range(4):
                pp = zk_amd.ProductPoly.new([A.clone(), B.clone()])
                backend = GpuShardBackend(pp, 2, claimed, world)
                torch.cuda.synchronize()
                dist.barrier()
                t1 = time.perf_counter()
                rp, ch = ShardedSumcheckProver(backend).prove_partial()
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t1)
                backend.close()
            tt = torch.tensor([sorted(ts[1:])[len(ts[1:]) // 2]], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            chk = torch.from_numpy(ch.view("int64").copy()).cuda()
            ref = chk.clone()
            dist.broadcast(ref, 0)
            same = bool((chk == ref).all().item())
            if rank == 0:
                result.setdefault("extra", {})[f"sharded_sumcheck_ms_shard2p{ns}_k2_d2_world{world}"] = float(tt.item()) * 1e3
                result["extra"]["sharded_challenges_identical_on_all_ranks"] = same
            # four-step NTT across the ranks (one all-to-all): every rank holds 2^22 points of a 2^22 * world point vector
            try:
                from zk_amd.distributed import GpuNttBackend, ShardedNtt
                xs = zk_amd.MultiLinearPolynomial.random(ctx, ns, 0x5EED0300, first_index=rank << ns)
                nb = GpuNttBackend(xs, rank, world)
                ShardedNtt(nb).forward()
                torch.cuda.synchronize()
                tn = []
                for it in range(3):
                    dist.barrier()
                    t1 = time.perf_counter()
                    ShardedNtt(nb).forward()
                    torch.cuda.synchronize()
                    tn.append(time.perf_counter() - t1)
                tt = torch.tensor([sorted(tn)[1]], dtype=torch.float64, device="cuda")
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                if rank == 0:
                    result["extra"][f"sharded_ntt_ms_2p{ns}_per_rank_world{world}"] = float(tt.item()) * 1e3
            except Exception as e:
                if rank == 0:
                    result.setdefault("extra", {})["sharded_ntt_error"] = repr(e)
            ctx.use_own_stream()
        except Exception as e:
            if rank == 0:
                result.setdefault("extra", {})["sharded_error"] = repr(e)
        finally:
            watchdog.cancel()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        base, _ = cpu_baseline(field)
        result["cpu_baseline"] = base

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
