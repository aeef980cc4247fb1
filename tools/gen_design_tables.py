#!/usr/bin/env python3
"""DESIGN.md section 5's measured block, generated from the artefacts under profiles/ so that the document cannot drift from them.

  python3 tools/gen_design_tables.py            # print the block
  python3 tools/gen_design_tables.py --write    # rewrite it in DESIGN.md (between the BEGIN / END GENERATED markers)
  python3 tools/gen_design_tables.py --check    # exit 1 when DESIGN.md's block differs from what the artefacts say

Sources (ROUND = r06; tests/test_design_tables.py runs --check in the CPU suite):
  profiles/<ROUND>_bench_final.json                 the bench line of `python3 bench.py` on one MI355X
  profiles/<ROUND>_bench_fold_kernel_stats.csv      rocprofv3 --kernel-trace --stats of the bench command (k_fold_msb's average)
  profiles/<ROUND>_bench_fold_under_rocprof.json    the bench line printed by that profiled run (its own HIP-event average)
  profiles/<ROUND>_prover_ntt_gkr_kernel_stats.md   tools/summarize_prof_all.py: per-kernel stats of prover / NTT / GKR / evaluate
  profiles/<ROUND>_world1_rccl_final.json           the bench line under torchrun at N = 1 (the sharded leg through RCCL, one rank)
"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = "r06"
BEGIN = "<!-- BEGIN GENERATED: tools/gen_design_tables.py (do not edit by hand) -->"
END = "<!-- END GENERATED -->"
PEAK = 8000.0   # GB/s, MI355X_MICROARCH.md


def P(name):
    return os.path.join(ROOT, "profiles", f"{ROUND}_{name}")


def bench_line(path):
    lines = [l for l in open(path) if l.startswith("{")]
    return json.loads(lines[-1])


def kernel_rows(md_path):
    """section -> {kernel: (calls, avg_us, min_us, max_us, pct)} from a summarize_prof_all.py file"""
    out, sec = {}, None
    for l in open(md_path):
        m = re.match(r"## (\w+)", l)
        if m:
            sec = m.group(1)
            out[sec] = {}
            continue
        m = re.match(r"\| `([^`]+)` \| (\d+) \| ([\d.]+) \| ([\d.]+) \| ([\d.]+) \| ([\d.]+) \| ([\d.]+) \|", l)
        if m and sec:
            out[sec][m.group(1).replace("zk::", "")] = (int(m.group(2)), float(m.group(3)), float(m.group(4)), float(m.group(5)), float(m.group(7)))
    return out


def frac(nbytes, us):
    return nbytes / (us * 1e-6) / 1e9 / PEAK


def build():
    b = bench_line(P("bench_final.json"))
    ex, rf = b["extra"], b["roofline"]
    L = [BEGIN, ""]
    src = f"`profiles/{ROUND}_bench_final.json`"
    L += [f"Generated from {src}, `{ROUND}_bench_fold_kernel_stats.csv`, `{ROUND}_prover_ntt_gkr_kernel_stats.md`, `{ROUND}_world1_rccl_final.json` "
          "(the builder's boxes, one MI355X each).  The driver's own record of the round before is printed beside them further down: every "
          "box is a different machine, and the same library has measured up to 7 % apart on two of them.", ""]
    L += ["| Quantity | Value | Of the roof |", "|---|---|---|"]
    L.append(f"| **`k_fold_msb`, 2^24 → 2^23 BN254 Fr** (the metric; {b['steps']} timed launches, HIP events on the launch stream) | "
             f"{rf['kernel_ms'] * 1e3:.1f} µs mean, {rf['kernel_ms_median'] * 1e3:.1f} median, {rf['kernel_ms_min'] * 1e3:.1f} min; "
             f"{b['value']:.3e} field-ops/s; step {b['ms_per_step'] * 1e3:.1f} µs | **{rf['frac']:.3f}** of 8 TB/s ({rf['achieved']:.0f} GB/s algorithmic; "
             f"{rf['frac_at_median']:.3f} at the median) |")
    L.append(f"| HBM traffic per launch (live `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE`, separate passes, gfx950 correction) | "
             f"{rf['traffic'] / 1e6:.2f} MB for {rf['algorithmic_bytes'] / 1e6:.2f} MB algorithmic | {rf['traffic_over_algorithmic']:.5f} × |")
    if os.path.exists(P("bench_fold_kernel_stats.csv")):
        for r in csv.DictReader(open(P("bench_fold_kernel_stats.csv"))):
            if "k_fold_msb" in r["Name"]:
                avg = float(r["TotalDurationNs"]) / int(r["Calls"]) / 1e3
                own = ""
                if os.path.exists(P("bench_fold_under_rocprof.json")):
                    u = bench_line(P("bench_fold_under_rocprof.json"))["roofline"]
                    own = f"; the bench's own HIP events in that run: {u['kernel_ms'] * 1e3:.2f} µs"
                L.append(f"| the bench command under `rocprofv3 --kernel-trace --stats` | `k_fold_msb` {r['Calls']} calls, average **{avg:.2f} µs** "
                         f"(min {float(r['MinNs']) / 1e3:.1f}){own} | {frac(rf['algorithmic_bytes'], avg):.3f} |")
    pw = b["sumcheck_prover_wall_clock_ms"]
    L.append(f"| `prove_partial` k = 2, D = 2 (`std::chrono` around the whole call inside the library, median / min of 11) | "
             f"n = 20: **{pw['n20_k2_d2']:.3f}** / {pw['n20_k2_d2_min']:.3f} ms; n = 24: **{pw['n24_k2_d2']:.3f}** / {pw['n24_k2_d2_min']:.3f} ms | "
             f"{frac(2 * 96 * 2 ** 20, pw['n20_k2_d2'] * 1e3):.3f}; {frac(2 * 96 * 2 ** 24, pw['n24_k2_d2'] * 1e3):.3f} (k·96·2^n B) |")
    if "sumcheck_prove_absorbing_ms_n24_k2_d2" in ex:
        L.append(f"| `prove` (tables absorbed first, `prover.rs:15-20`: a serial host Keccak over k·2^n·32 B) | n = 20: {ex['sumcheck_prove_absorbing_ms_n20_k2_d2']:.0f} ms; "
                 f"n = 24: {ex['sumcheck_prove_absorbing_ms_n24_k2_d2']:.0f} ms | host-bound |")
    if ex.get("sumcheck_prove_partial_ms_n22_k3_d3"):
        L.append(f"| `prove_partial` k = 3, D = 3, n = 22 (fused rounds of three tables on the LDS-DMA kernels; gated against the oracle's proof) | "
                 f"**{ex['sumcheck_prove_partial_ms_n22_k3_d3']:.3f}** / {ex['sumcheck_prove_partial_ms_n22_k3_d3_min']:.3f} ms | "
                 f"{frac(3 * 96 * 2 ** 22, ex['sumcheck_prove_partial_ms_n22_k3_d3'] * 1e3):.3f} (k·96·2^n B) |")
    L.append(f"| eight layers of `prove_partial` on 3 factors of 2^20, D = 3 / GKR driver depth 8 × 2^20 (random add/mul wiring) | "
             f"{ex['gkr_shaped_depth8_width2p20_k3_d3_ms']:.2f} ms / prove **{ex['gkr_depth8_width2p20_addmul_prove_ms']:.2f} ms**, verify "
             f"{ex['gkr_depth8_width2p20_addmul_verify_ms']:.2f} ms, proof {ex['gkr_proof_bytes']} B | latency-bound |")
    if ex.get("gkr_shaped_depth8_width2p20_k3_d3_concurrent_ms"):
        L.append(f"| the same eight INDEPENDENT proofs in flight at once (`zk_sumcheck_prove_batch`: one launch per round for all eight; every proof bit-identical "
                 f"to its back-to-back twin, gated) | k = 3: **{ex['gkr_shaped_depth8_width2p20_k3_d3_concurrent_ms']:.2f} ms** (back to back "
                 f"{ex['gkr_shaped_depth8_width2p20_k3_d3_ms']:.2f}); k = 2: **{ex['eight_independent_proofs_n20_k2_d2_concurrent_ms']:.2f} ms** (back to back "
                 f"{ex['eight_independent_proofs_n20_k2_d2_back_to_back_ms']:.2f}) | "
                 f"{frac(8 * 3 * 96 * 2 ** 20, ex['gkr_shaped_depth8_width2p20_k3_d3_concurrent_ms'] * 1e3):.2f}; "
                 f"{frac(8 * 2 * 96 * 2 ** 20, ex['eight_independent_proofs_n20_k2_d2_concurrent_ms'] * 1e3):.2f} (B·k·96·2^n B) |")
    rn = b["roofline_ntt"]
    L.append(f"| NTT 2^24 forward / inverse (3 passes, `zk_bench_ntt`) | **{ex['ntt_2p24_ms']:.3f}** / {ex['intt_2p24_ms']:.3f} ms | "
             f"{rn['frac']:.2f} of the measured {rn['peak']:.3e} modmul/s; {rn['hbm_frac_one_pass_bytes']:.3f} of HBM on the one-pass bytes (P = 3 caps it at 0.33) |")
    re_ = b["roofline_evaluate"]
    ev = " / ".join(f"{ex[f'evaluate_us_n{n}']:.1f}" for n in (18, 19, 20, 21))
    ev2 = " / ".join(f"{ex[f'evaluate_us_n{n}_bn254']:.1f}" for n in (22, 23, 24))
    evb = " / ".join(f"{ex[f'evaluate_us_n{n}_bls12_381']:.1f}" for n in (18, 19, 20, 21))
    L.append(f"| `evaluate` (the reference's criterion bench, `polynomial_evaluation.rs:85-105`), whole call, median | BN254 n = 18 / 19 / 20 / 21: "
             f"**{ev} µs**; n = 22 / 23 / 24: {ev2} µs; BLS12-381 Fr n = 18..21: {evb} µs | 2^24 device time {re_['device_us']:.1f} µs = "
             f"**{re_['frac']:.3f}** of HBM (whole call {re_['frac_of_the_whole_call']:.3f}) |")
    rows = ex.get("rows_2p24", {})
    if rows:
        def rr(k):
            return f"{rows[k]['us']:.0f} µs = {rows[k]['hbm_frac']:.2f}" if k in rows else "—"
        names = [k for k in rows]
        pe = [k for k in names if k.startswith("partial_evaluate")]
        tail = ""
        if "coeff_to_evaluation_2p24_1k_terms" in rows:
            z = rows["coeff_to_evaluation_2p24_1k_terms"]
            tail += f"; `to_evaluation_form` (1 k terms): {z['us'] / 1e3:.2f} ms"
            if "coeff_to_evaluation_2p24_64k_terms" in rows:
                tail += f", (64 k terms, ordered on the device): {rows['coeff_to_evaluation_2p24_64k_terms']['us'] / 1e3:.2f} ms"
        if "to_bytes_2p24" in rows:
            z = rows["to_bytes_2p24"]
            tail += f"; `to_bytes`: {z['ms_fresh_destination']:.1f} ms into a fresh destination, {z['ms_mapped_destination']:.1f} ms into a mapped one"
        settled = " (single calls; 0.8 s of untimed folds first -- right after idle seconds the same call measures up to 40 % longer, `profiles/r06_a_rows_probe.log`)" if ex.get("rows_2p24_note") else ""
        L.append("| other (a)-rows at 2^24 through the allocating calls" + settled + " | `prod_reduce` k = 2 / 3: " + rr("prod_reduce_k2_2p24") + " / " + rr("prod_reduce_k3_2p24") +
                 "; `partial_evaluate` at " + ", ".join(f"{k.split('_')[-1]}: {rr(k)}" for k in pe) + tail + " | (of 8 TB/s) |")
    L.append(f"| multiplier cores, register resident (`zk_bench_modmul`) | `fe_mul` {ex['modmul_per_s_register_resident']:.3e} /s, `fe_mul29` "
             f"{ex['modmul29_per_s_register_resident']:.3e} /s; 1-GiB streaming copy {ex['copy_gbps_1GiB']:.0f} GB/s | the integer roof; the copy ceiling |")
    cb = b["cpu_baseline"]
    L.append(f"| CPU beside it (`cpu_baseline`: the oracle on the GPU box's host, `kind` = port) | faithful, {cb['cores']} thread: fold {cb['value']:.2e} field-ops/s "
             f"({cb.get('fold_2p24_faithful_ms', 0):.0f} ms per 2^24 fold); `prove_partial` {cb['sumcheck_prove_partial_ms_n20_k2_d2']:.0f} ms (n = 20), "
             f"**{cb['sumcheck_prove_partial_ms_n24_k2_d2'] / 1e3:.2f} s** (n = 24); recursive `fft` 2^20: {cb['fft_faithful_recursive_ms']['2p20'] / 1e3:.1f} s.  "
             f"Fused + OpenMP, {cb['optimised']['cores']} threads: fold {cb['optimised']['value']:.2e} field-ops/s, n = 24 proof "
             f"{cb['optimised']['sumcheck_prove_partial_ms_n24_k2_d2']:.0f} ms | context, not the target |")
    if os.path.exists(P("world1_rccl_final.json")):
        w = bench_line(P("world1_rccl_final.json"))
        wx = w["extra"]
        per = ", ".join(f"gather_below {g}: {wx[f'sharded_sumcheck_ms_n24_k2_d2_world1_gather_below{g}']:.3f} ms" for g in (10, 13, 16)
                        if f"sharded_sumcheck_ms_n24_k2_d2_world1_gather_below{g}" in wx)
        plain = w["sumcheck_prover_wall_clock_ms"]["n24_k2_d2"]
        best = wx["sharded_sumcheck_ms_n24_k2_d2_world1"]
        L.append(f"| sharded prover through RCCL {wx.get('rccl_version', '')} at ONE rank (`torch.distributed.run --nproc-per-node 1`), n = 24 | {per}; plain prover in the same run "
                 f"{plain:.3f} ms | best +{(best / plain - 1) * 100:.1f} % over plain; all-reduce of 24 lanes {wx['allreduce_24_lanes_latency_us']:.1f} µs back to back |")
    L.append("")
    # the driver's last record (a different box): the same keys side by side, so the spread between machines is on the page
    drv_path = os.path.join(ROOT, f"BENCH_r{int(ROUND[1:]) - 1:02d}.json")
    if os.path.exists(drv_path):
        try:
            drv = json.load(open(drv_path))["parsed"]
            dx = drv.get("extra", {})
            if not dx:   # the driver keeps the head of the line only: read what it kept
                dx = {}
            tail_txt = json.load(open(drv_path))["run"]["stdout_tail"]

            def grab(key):
                m = re.search(r'"%s": ([0-9.eE+-]+)' % re.escape(key), tail_txt)
                return float(m.group(1)) if m else None
            pairs = [("`k_fold_msb` µs", drv["roofline"]["kernel_ms"] * 1e3, rf["kernel_ms"] * 1e3),
                     ("`prove_partial` n = 20 ms", grab("sumcheck_prove_partial_ms_n20_k2_d2"), pw["n20_k2_d2"]),
                     ("`prove_partial` n = 24 ms", grab("sumcheck_prove_partial_ms_n24_k2_d2"), pw["n24_k2_d2"]),
                     ("8 × (k = 3) back to back ms", grab("gkr_shaped_depth8_width2p20_k3_d3_ms"), ex["gkr_shaped_depth8_width2p20_k3_d3_ms"]),
                     ("GKR prove ms", grab("gkr_depth8_width2p20_addmul_prove_ms"), ex["gkr_depth8_width2p20_addmul_prove_ms"]),
                     ("NTT 2^24 ms", grab("ntt_2p24_ms"), ex["ntt_2p24_ms"]),
                     ("`evaluate` n = 21 µs", grab("evaluate_us_n21"), ex["evaluate_us_n21"]),
                     ("`evaluate` 2^24 device µs", grab("evaluate_device_us_n24_bn254"), re_["device_us"]),
                     ("copy GB/s", grab("copy_gbps_1GiB"), ex["copy_gbps_1GiB"])]
            L += [f"The driver's record of the round before (`BENCH_r{int(ROUND[1:]) - 1:02d}.json`, its own box, that round's library) beside this round's builder-box numbers -- "
                  "rows whose code did not change between the two show what a change of machine alone does:", "",
                  "| quantity | driver, round before | builder box, this round | ratio |", "|---|---|---|---|"]
            for name, a, c in pairs:
                if a and c:
                    L.append(f"| {name} | {a:.4g} | {c:.4g} | {c / a:.3f} |")
            L.append("")
        except Exception as e:   # noqa: BLE001 -- an unreadable driver record must not break the generator
            L += [f"(driver record {os.path.basename(drv_path)} unreadable: {e})", ""]
    if os.path.exists(P("prover_ntt_gkr_kernel_stats.md")):
        ks = kernel_rows(P("prover_ntt_gkr_kernel_stats.md"))
        L += [f"Kernels that carry the time (rocprofv3 kernel trace, `profiles/{ROUND}_prover_ntt_gkr_kernel_stats.md`; bytes = algorithmic bytes of the launch):", "",
              "| kernel (workload) | calls | avg µs (min) | share of the run | of the HBM peak at the average |", "|---|---|---|---|---|"]

        def row(sec, kern, what, nbytes=None, pick="avg"):
            if sec in ks and kern in ks[sec]:
                c, a, mn, mx, pct = ks[sec][kern]
                t = mx if pick == "max" else a
                fr = f"{frac(nbytes, t):.2f}" + (" (largest launch)" if pick == "max" else "") if nbytes else "—"
                L.append(f"| `{kern}` ({what}) | {c} | {a:.1f} ({mn:.1f}; max {mx:.1f}) | {pct:.1f} % | {fr} |")
        row("sumcheck_n24", "k_round0_dot29<0>", "n = 24 round 0: 2 × 2^24 elements read", 2 * 32 * 2 ** 24)
        row("sumcheck_n24", "k_round0_glds<0>", "n = 24 round 0 on the LDS-DMA kernel: 2 × 2^24 elements read", 2 * 32 * 2 ** 24)
        row("sumcheck_n24", "k_round_kd<2, 2, true, 0, true, true>", "n = 24 fused rounds; largest: 2 × (2^24 read + 2^23 written)", 2 * 48 * 2 ** 24, "max")
        row("sumcheck_n24", "k_round_tail", "n = 24: second-stage reduction + transcript step of the classic rounds")
        row("sumcheck_n20", "k_round_pipe<2, 2, 0, true>", "n = 20: pipelined rounds, ≤ 2^12 pairs")
        row("sumcheck_n20", "k_round_tail", "n = 20: classic tails")
        row("sumcheck_n20", "k_finish_pipe<2, 2, 0>", "n = 20: the last 8 rounds in one launch")
        row("sumcheck_k3_n20", "k_round_fused_glds<3, 0, false>", "k = 3, n = 20: fused rounds of three tables on the LDS-DMA kernel, 2^18 .. 2^16 pairs")
        row("sumcheck_k3_n20", "k_round_kd<3, 3, false, 0, false, true>", "k = 3, n = 20 round 0: 3 × 2^20 elements read", 3 * 32 * 2 ** 20)
        row("batch8_k3_n20", "k_round_fused_glds_b<3, 0, false>", "batched fused rounds of B = 2 / 4 / 8 proofs, k = 3, n = 20, LDS-DMA kernel: grid (x, B)")
        row("batch8_k3_n20", "k_round_kd_b<3, 3, true, 0, true, true>", "batched fused rounds of B = 2 / 4 / 8 proofs, k = 3, n = 20: grid (x, B)")
        row("batch8_k3_n20", "k_round_pipe_b<3, 3, 0, true>", "batched pipelined rounds: B transcript blocks + B sets of work blocks per launch")
        row("batch8_k3_n20", "k_finish_pipe_b<3, 3, 0>", "batched finisher: B workgroups, the last rounds of all proofs")
        row("evaluate", "k_eval_stream", "evaluate at 21 and 24 variables; largest: 2^24 elements read", 32 * 2 ** 24, "max")
        row("ntt", "k_ntt_pass<8, false>", "2^24 points, passes 0 and 1: 2 × 2^24 × 32 B each", 2 * 32 * 2 ** 24)
        row("ntt", "k_ntt_pass<8, true>", "2^24 points, last pass", 2 * 32 * 2 ** 24)
        row("gkr", "k_round_kd<2, 2, true, 1, true, true>", "GKR layer polynomial W·H + B, fused rounds")
        row("gkr", "k_round_tail", "GKR: classic tails")
        L.append("")
    L.append(END)
    return "\n".join(L)


def main():
    block = build()
    path = os.path.join(ROOT, "DESIGN.md")
    if "--write" in sys.argv or "--check" in sys.argv:
        s = open(path).read()
        a, z = s.index(BEGIN), s.index(END) + len(END)
        if "--check" in sys.argv:
            if s[a:z] != block:
                sys.stderr.write("DESIGN.md section 5 differs from the artefacts under profiles/: run tools/gen_design_tables.py --write\n")
                sys.exit(1)
            return
        open(path, "w").write(s[:a] + block + s[z:])
        return
    print(block)


if __name__ == "__main__":
    main()
