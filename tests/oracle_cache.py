"""Oracle answers for the forced-kernel-path sweeps, computed ONCE per test session and kept on disk.

tests/skip1_check.py and tests/shard_skip_check.py run in child processes (the library reads its ZK_* switches once per process);
every child used to re-prove the same (field, k, D, n) grid with the CPU oracle before touching the GPU.  The parent now fills a
directory ($ZK_ORACLE_CACHE, a pytest tmp dir) with those proofs -- on several CPU-only worker processes, `python oracle_cache.py
--prefill <spec.json>` -- and the children read them back.  A missing entry is simply computed (and stored) on the spot, so the
children stay runnable by hand.  Inputs are regenerated from their seeds (orc.fill_random); only the oracle's OUTPUTS are cached.
Test infrastructure only: nothing under zk_amd/ imports this.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from oracle import binding as orc  # noqa: E402
from oracle import gkr_ref  # noqa: E402


def _path(name):
    d = os.environ.get("ZK_ORACLE_CACHE")
    return os.path.join(d, name + ".npz") if d else None


def _load(name):
    p = _path(name)
    if p and os.path.exists(p):
        with np.load(p) as z:
            return {k: z[k] for k in z.files}
    return None


def _store(name, **arrays):
    p = _path(name)
    if p:
        tmp = f"{p}.{os.getpid()}.tmp.npz"
        np.savez(tmp, **arrays)
        os.replace(tmp, p)   # atomic: a concurrent reader sees the whole file or none
    return arrays


def sumcheck_tables(field, k, n, seed):
    return [orc.fill_random(field, seed + f, 1 << n) for f in range(k)]


def sumcheck_case(field, k, D, n, seed, wrong=0, tabs=None):
    """tables (regenerated), the claimed sum (true sum + `wrong`), and the faithful oracle's prove_partial of them
    -> (tabs, s, round_polys, challenges)"""
    name = f"sc_f{field}_k{k}_d{D}_n{n}_s{seed}_w{wrong}"
    tabs = tabs if tabs is not None else sumcheck_tables(field, k, n, seed)
    hit = _load(name)
    if hit is None:
        s = orc.sum_elems(field, orc.prod_reduce(field, n, tabs))
        if wrong:
            s = orc.add(field, s, orc.from_int(field, wrong))
        rp, ch = orc.sumcheck_prove(field, n, tabs, D, s, False)
        hit = _store(name, s=s, rp=rp, ch=ch)
    return tabs, hit["s"], hit["rp"], hit["ch"]


def terms_case(field, n, seed=77000):
    """the two-term GKR layer shape A.B + C (gkr_ref.prove_partial_terms, the big-int definition) on seeded tables
    -> (tables [[A, B], [C]] as element arrays, claimed sum element, round polys (n, 3, 4), challenges (n, 4), finals (3, 4))"""
    name = f"terms_f{field}_n{n}_s{seed}"
    tabs = [[orc.fill_random(field, seed + 10 * n + f, 1 << n) for f in range(2)], [orc.fill_random(field, seed + 10 * n + 2, 1 << n)]]
    hit = _load(name)
    if hit is None:
        p = orc.modulus(field)
        ints = [[orc.to_ints(field, t) for t in term] for term in tabs]
        s = sum(a * b + c for a, b, c in zip(ints[0][0], ints[0][1], ints[1][0])) % p
        rp, ch, fin = gkr_ref.prove_partial_terms(field, ints, 2, s)
        hit = _store(name, s=orc.from_int(field, s), rp=np.stack([orc.from_ints(field, r) for r in rp]), ch=orc.from_ints(field, ch),
                     fin=orc.from_ints(field, fin))
    return tabs, hit["s"], hit["rp"], hit["ch"], hit["fin"]


def run_spec(spec):
    for item in spec:
        if item[0] == "sc":
            sumcheck_case(*item[1:])
        elif item[0] == "terms":
            terms_case(*item[1:])
        else:
            raise ValueError(item)


def prefill(spec, workers=8):
    """compute every entry of `spec` (list of ["sc", field, k, D, n, seed, wrong] / ["terms", field, n]) that the cache lacks, on
    `workers` CPU-only child processes (fresh interpreters: the caller may hold a GPU context, which must not be forked)"""
    import subprocess
    import tempfile

    uniq = sorted({json.dumps(i) for i in spec}, key=lambda j: -json.loads(j)[4 if json.loads(j)[0] == "sc" else 2])
    items = [json.loads(j) for j in uniq]
    workers = max(1, min(workers, len(items)))
    procs = []
    for w in range(workers):
        part = items[w::workers]   # sorted by size: round-robin balances the big cases
        f = tempfile.NamedTemporaryFile("w", suffix=".json", delete=False)
        json.dump(part, f)
        f.close()
        procs.append((subprocess.Popen([sys.executable, os.path.abspath(__file__), "--prefill", f.name], env=dict(os.environ, OMP_NUM_THREADS="1")), f.name))
    for pr, fn in procs:
        rc = pr.wait()
        os.unlink(fn)
        if rc != 0:
            raise RuntimeError(f"oracle prefill worker failed with {rc}")
    return len(items)


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--prefill":
        with open(sys.argv[2]) as fh:
            run_spec(json.load(fh))
    else:
        sys.exit("usage: oracle_cache.py --prefill spec.json")
