"""Sumcheck prover over a table sharded across the GPUs of one node.

The reference (sumcheck/src/prover.rs:33-73) is a single-process loop; this module is its multi-GPU form: one process
per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI) for the one exchange per round.

Sharding.  The reference folds variable 0 = the index MSB first (pairing_index.rs:61-65), pairing j with
j + 2^(m-1).  Rank g of W = 2^w holds {idx : idx mod W == g} (the LAST w variables select the rank) as an
(n-w)-variable table with local index idx >> w.  Then every pair of rounds 0 .. n-w-1 is local, every rank runs the
same rounds with the same challenges, and the global round sums are the field sums of the local ones:

    per round:  local (fold +) sums  ->  ONE all-reduce of (D+1)*8 int64 lanes  ->  transcript step on every rank

(the lanes are the 32-bit digits of the Montgomery representatives; integer lane sums are exact and cannot overflow,
RCCL has no mod-p reduction).  After n-w rounds every rank holds one element per factor; one all-gather of W*k
elements gives the w-variable remainder (indexed by rank = the low index bits, i.e. in the reference's own order)
and every rank finishes the last w rounds redundantly -- W elements, no further collective.

Two drivers of the same protocol:
  * `GpuShardBackend.run(comm, gather_below)` -- the product path: the WHOLE loop inside the library
    (zk_shard_prover_run), collectives issued by the library itself on the context's stream (`RcclComm`: RCCL over xGMI, no
    host synchronisation between rounds; `HostComm`: the same control flow with the transport called back into the host,
    used by the multi-process tests that put several ranks on one GPU, where RCCL cannot run).
  * `ShardedSumcheckProver` -- the stepwise orchestration in Python over torch.distributed (backend-agnostic: tests drive
    it over gloo on CPU with a checker backend, and over gloo with `GpuShardBackend`, staging the lanes through the host).
"""
import numpy as np

from ._lib import HOST_ALLGATHER, HOST_ALLREDUCE, HOST_ALLTOALL, c, check, lib, u64p


def shard_of(table, rank, world):
    """Rows {idx : idx mod world == rank} of a (2^n, 4) table, in local-index order (idx >> log2 world)."""
    t = np.ascontiguousarray(table, dtype=np.uint64).reshape(-1, 4)
    return np.ascontiguousarray(t[rank::world])


def sliced_shard_of(vector, rank, world):
    """ShardedNtt's "sliced" layout of an (N, 4) vector: the k with k mod M in the rank's slice, as W rows of M/W
    (local position k2*(M/W) + c  <->  k = rank*M/W + c + M*k2)."""
    v = np.ascontiguousarray(vector, dtype=np.uint64).reshape(-1, 4)
    m = v.shape[0] // world
    c = m // world
    return np.ascontiguousarray(v.reshape(world, m, 4)[:, rank * c:(rank + 1) * c].reshape(-1, 4))


class _DevArray:
    """Zero-copy view of a device buffer for torch.as_tensor (__cuda_array_interface__ v2)."""

    def __init__(self, ptr, n_int64):
        self.__cuda_array_interface__ = {"shape": (n_int64,), "typestr": "<i8", "data": (ptr, False), "version": 2}


class RcclComm:
    """zk_comm over RCCL (xGMI): one per rank.  The 128-byte unique id is created on rank 0 and broadcast through the
    already-initialised torch.distributed group (any backend) -- a Rust host would use whatever channel it has."""

    def __init__(self, ctx, group=None):
        import torch
        import torch.distributed as dist

        self.ctx = ctx
        if dist.is_available() and dist.is_initialized():
            self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        else:
            self.world, self.rank = 1, 0
        buf = c.create_string_buffer(128)
        if self.rank == 0:
            check(lib.zk_comm_unique_id(buf))
        if self.world > 1:
            obj = [buf.raw]
            dist.broadcast_object_list(obj, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            buf = c.create_string_buffer(obj[0], 128)
        h = c.c_void_p()
        check(lib.zk_comm_create_rccl(ctx._h, buf, self.world, self.rank, c.byref(h)))
        self._h = h
        del torch


    def info(self):
        """what the transport reports: {"rccl_version", "ranks", "rank"} (zk_comm_info: ncclGetVersion / ncclCommCount / ncclCommUserRank;
        version 0 = host-callback transport)"""
        v, n, r = c.c_int32(), c.c_uint32(), c.c_uint32()
        check(lib.zk_comm_info(self._h, c.byref(v), c.byref(n), c.byref(r)))
        return {"rccl_version": v.value, "ranks": n.value, "rank": r.value}

    def close(self):
        if getattr(self, "_h", None):
            lib.zk_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


class HostComm:
    """zk_comm whose transport is called back into this process: the library stages each buffer through pinned memory and
    the callbacks run the collective over a torch.distributed group on CPU tensors (gloo).  Same control flow as RcclComm;
    exists so that several ranks can share ONE GPU in the multi-process tests (RCCL refuses two ranks on a device)."""

    def __init__(self, ctx, group=None):
        import torch
        import torch.distributed as dist

        self.ctx, self.group = ctx, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        world = self.world

        def _view(ptr, n):
            return torch.from_numpy(np.ctypeslib.as_array(ptr, shape=(n,)).view(np.int64))

        def allreduce(_user, buf, n):
            try:
                dist.all_reduce(_view(buf, n), op=dist.ReduceOp.SUM, group=group)
                return 0
            except Exception:   # never let an exception cross the C boundary
                return 1

        def allgather(_user, send, n, recv):
            try:
                dist.all_gather(list(_view(recv, n * world).chunk(world)), _view(send, n), group=group)
                return 0
            except Exception:
                return 1

        def alltoall(_user, send, recv, n):
            try:
                dist.all_to_all_single(_view(recv, n * world), _view(send, n * world), group=group)
                return 0
            except Exception:
                return 1

        self._cbs = (HOST_ALLREDUCE(allreduce), HOST_ALLGATHER(allgather), HOST_ALLTOALL(alltoall))   # keep alive
        h = c.c_void_p()
        check(lib.zk_comm_create_host(ctx._h, self.world, self.rank, c.cast(self._cbs[0], c.c_void_p), c.cast(self._cbs[1], c.c_void_p),
                                      c.cast(self._cbs[2], c.c_void_p), None, c.byref(h)))
        self._h = h

    def info(self):
        """what the transport reports: {"rccl_version", "ranks", "rank"} (zk_comm_info: ncclGetVersion / ncclCommCount / ncclCommUserRank;
        version 0 = host-callback transport)"""
        v, n, r = c.c_int32(), c.c_uint32(), c.c_uint32()
        check(lib.zk_comm_info(self._h, c.byref(v), c.byref(n), c.byref(r)))
        return {"rccl_version": v.value, "ranks": n.value, "rank": r.value}

    def close(self):
        if getattr(self, "_h", None):
            lib.zk_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


def ntt_sharded(comm, shard, inverse=False):
    """fft / ifft of the vector held as index-mod-world shards, inside the library (zk_ntt_sharded): returns a new table."""
    from .api import MultiLinearPolynomial

    out = MultiLinearPolynomial.alloc(shard.ctx, shard.n_vars())
    check(lib.zk_ntt_sharded(shard.ctx._h, comm._h, shard._h, int(inverse), out._h))
    return out


class GpuShardBackend:
    """One rank's share of the prover on its GPU (zk_shard_prover_* in include/zk_amd.h).  The factor tables of `poly` are
    CONSUMED (folded in place; only a product that lists one table twice is proved out of place)."""

    def __init__(self, poly, max_var_degree, claimed_sum, world, torch_stream=True):
        import torch

        self.ctx, self.poly = poly.ctx, poly
        if torch_stream:   # stepwise driver: kernels and torch's collectives are ordered on torch's current stream;
            self.ctx.use_torch_stream()   # run(comm) needs no torch stream (the library enqueues its own collectives)
        self.k, self.D, self.world = len(poly.polynomials), max_var_degree, world
        s = np.ascontiguousarray(claimed_sum, dtype=np.uint64).reshape(4)
        arr = (c.c_void_p * self.k)(*[q._h for q in poly.polynomials])
        h = c.c_void_p()
        check(lib.zk_shard_prover_create(self.ctx._h, c.cast(arr, c.POINTER(c.c_void_p)), self.k, self.D,
                                         s.ctypes.data_as(u64p), world, c.byref(h)))
        self._h = h
        loc, tot = c.c_uint64(), c.c_uint64()
        check(lib.zk_shard_prover_rounds(h, c.byref(loc), c.byref(tot), None))
        self.local_rounds, self.total_rounds = loc.value, tot.value
        p, n = c.c_void_p(), c.c_uint64()
        check(lib.zk_shard_prover_lanes_ptr(h, c.byref(p), c.byref(n)))
        self.lanes = torch.as_tensor(_DevArray(p.value, n.value), device=f"cuda:{self.ctx.device}")
        self._torch = torch

    def close(self):
        if getattr(self, "_h", None):
            lib.zk_shard_prover_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def run(self, comm, gather_below=10):
        """the whole prover inside the library (zk_shard_prover_run): per-round all-reduce, tail all-gather, tail rounds"""
        check(lib.zk_shard_prover_run(self._h, comm._h, gather_below))
        return self.results()

    def run_phases(self, comm, gather_below=10):
        """the same run with the library's HIP-event breakdown -> (round_polys, challenges, {phase: ms})"""
        ms = (c.c_double * 4)()
        check(lib.zk_shard_prover_run_phases(self._h, comm._h, gather_below, ms))
        rp, ch = self.results()
        return rp, ch, {"local_kernels_ms": ms[0], "allreduce_ms": ms[1], "gather_ms": ms[2], "tail_rounds_ms": ms[3]}

    def local_vars_left(self):
        """variables of the local shard tables that no completed round has consumed yet"""
        loc, done = c.c_uint64(), c.c_uint64()
        check(lib.zk_shard_prover_rounds(self._h, c.byref(loc), None, c.byref(done)))
        return loc.value - done.value

    def round_begin(self):
        check(lib.zk_shard_prover_round_begin(self._h))
        return self.lanes

    def round_finish(self):
        check(lib.zk_shard_prover_round_finish(self._h))

    def tail(self):
        p, n = c.c_void_p(), c.c_uint64()
        check(lib.zk_shard_prover_tail_ptr(self._h, c.byref(p), c.byref(n)))
        return self._torch.as_tensor(_DevArray(p.value, n.value * 4), device=f"cuda:{self.ctx.device}")

    def tail_rounds(self, gathered):
        self._gathered = gathered.contiguous()   # keep alive until the stream has consumed it
        check(lib.zk_shard_prover_tail_rounds(self._h, c.c_void_p(self._gathered.data_ptr())))

    def results(self):
        rp = np.zeros((self.total_rounds, self.D + 1, 4), dtype=np.uint64)
        ch = np.zeros((max(self.total_rounds, 1), 4), dtype=np.uint64)
        check(lib.zk_shard_prover_results(self._h, rp.ctypes.data_as(u64p), ch.ctypes.data_as(u64p)))
        return rp, ch[: self.total_rounds]


class ShardedSumcheckProver:
    """SumcheckProver::prove_partial (prover.rs:24-30) for a table sharded over the ranks of `group`.

    backend: an object with local_rounds, total_rounds, local_vars_left(), round_begin() -> int64 lane tensor,
    round_finish(), tail() -> int64 tensor (k * 2^s elements), tail_rounds(gathered), results().  Every rank returns the
    same (round_polys, challenges) as the single-process prover on the unsharded table.

    gather_below: once the local shard tables have at most 2^gather_below elements, stop exchanging per round: one
    all-gather of the shard tables, then every rank finishes redundantly (a collective per round costs tens of
    microseconds, a round on a 2^10-element table less than that -- SURVEY 8e).
    """

    def __init__(self, backend, group=None, gather_below=10):
        self.backend, self.group, self.gather_below = backend, group, gather_below

    def prove_partial(self):
        import torch.distributed as dist

        b = self.backend
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1
        # a CPU-only group (gloo) under device tensors: stage the lanes / tail through the host
        stage = multi and dist.get_backend(self.group) == "gloo"
        while b.local_vars_left() > self.gather_below:
            lanes = b.round_begin()
            if multi and stage and lanes.is_cuda:
                h = lanes.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                lanes.copy_(h)
            elif multi:
                dist.all_reduce(lanes, op=dist.ReduceOp.SUM, group=self.group)   # the round's one collective
            b.round_finish()
        tail = b.tail()
        if multi:
            import torch

            src = tail.cpu() if (stage and tail.is_cuda) else tail
            parts = [torch.empty_like(src) for _ in range(dist.get_world_size(self.group))]
            dist.all_gather(parts, src, group=self.group)                        # once, W*k elements, rank-major
            gathered = torch.cat(parts).to(tail.device)
        else:
            gathered = tail
        b.tail_rounds(gathered)
        return b.results()


# ---------------------------------------------------------------------------------------------------------------------
# Four-step NTT across the ranks (SURVEY.md 8e "NTT does not shard without an all-to-all", 8 f4)
# ---------------------------------------------------------------------------------------------------------------------
class ShardedNtt:
    """fft / ifft (fft/src/lib.rs:4-19: out[k] = sum_i in[i] * w^(i*k), w = F::get_root_of_unity(N), natural order both
    sides) of an N = W * M point vector held by W = 2^w ranks, ONE all-to-all per transform.

    Layouts.  "strided": rank r holds x[r + W*j], j in [0, M) -- the index-mod-W shard the sharded prover uses.
    "sliced": rank s holds X[(s*M/W + c) + M*k2] at local position k2*(M/W) + c, k2 in [0, W), c in [0, M/W) -- the k with
    k mod M in the rank's slice.  forward: strided -> sliced; inverse: sliced -> strided (so forward . pointwise . inverse
    needs no re-layout).  With k = k1 + M*k2:

        X[k1 + M*k2] = sum_r w_W^(r*k2) * ( w_N^(r*k1) * NTT_M(shard r)[k1] )

    forward: local M-point NTT -> twiddle w_N^(r*k1) -> all-to-all (rank s receives the k1 of its slice from every r) ->
    W-point transforms across the received rows.  inverse: the same steps backwards with inverse roots, scaled by 1/N.
    backend: local_ntt(inverse), twiddle(inverse), across(inverse), send_tensor(), recv_tensor() (int64 views, M*4
    words each), result().  Host logic only; tests drive it over gloo with a checker backend.
    """

    def __init__(self, backend, group=None):
        self.backend, self.group = backend, group

    def _exchange(self):
        import torch.distributed as dist

        b = self.backend
        send, recv = b.send_tensor(), b.recv_tensor()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            if dist.get_backend(self.group) == "gloo" and send.is_cuda:   # CPU-only group under device tensors
                import torch

                hs = send.cpu()
                hr = torch.empty_like(hs)
                dist.all_to_all_single(hr, hs, group=self.group)
                recv.copy_(hr)
                return
            dist.all_to_all_single(recv, send, group=self.group)   # the transform's one collective: M/W elements per pair
        else:
            recv.copy_(send)

    def forward(self):
        b = self.backend
        b.local_ntt(False)
        b.twiddle(False)
        self._exchange()
        b.across(False)
        return b.result()

    def inverse(self):
        b = self.backend
        b.across(True)
        self._exchange()
        b.twiddle(True)
        b.local_ntt(True)
        return b.result()


class GpuNttBackend:
    """One rank's share of ShardedNtt on its GPU (zk_ntt, zk_mle_mul_powers, zk_dft_across in include/zk_amd.h).
    shard: MultiLinearPolynomial of m variables (M = 2^m elements in this rank's layout); it is not modified."""

    def __init__(self, shard, rank, world):
        import torch

        from .api import MultiLinearPolynomial, fe_from_int, fe_to_int, modulus, root_of_unity

        self.ctx, self.rank, self.world = shard.ctx, rank, world
        self.log_w = world.bit_length() - 1
        if world != 1 << self.log_w:
            raise ValueError("world size must be a power of two")
        self.m = shard.n_vars()
        if self.m < self.log_w:
            raise ValueError("a shard needs at least `world` elements")
        self.ctx.use_torch_stream()   # kernels and the all-to-all are ordered on one stream
        self.field = self.ctx.field
        p = modulus(self.field)
        w_n = fe_to_int(self.field, root_of_unity(self.field, self.m + self.log_w))
        self._tw = {False: fe_from_int(self.field, pow(w_n, rank, p)),
                    True: fe_from_int(self.field, pow(pow(w_n, -1, p), rank, p))}
        self._scale = {False: fe_from_int(self.field, 1), True: fe_from_int(self.field, pow(world, -1, p))}
        self.src = shard
        self.a = MultiLinearPolynomial.alloc(self.ctx, self.m)
        self.b = MultiLinearPolynomial.alloc(self.ctx, self.m)
        dev = f"cuda:{self.ctx.device}"
        self._ta = torch.as_tensor(_DevArray(self.a.device_ptr(), (1 << self.m) * 4), device=dev)
        self._tb = torch.as_tensor(_DevArray(self.b.device_ptr(), (1 << self.m) * 4), device=dev)
        self._out = None

    # forward: src -NTT-> a -twiddle-> a =exchange=> b -across-> a        inverse: src -across-> a =exchange=> b -twiddle-> b -iNTT-> a
    def local_ntt(self, inverse):
        check(lib.zk_ntt(self.ctx._h, (self.b if inverse else self.src)._h, int(inverse), self.a._h))
        self._out = self.a

    def twiddle(self, inverse):
        t = self.b if inverse else self.a
        check(lib.zk_mle_mul_powers(self.ctx._h, t._h, self._tw[inverse].ctypes.data_as(u64p),
                                    self._scale[inverse].ctypes.data_as(u64p)))

    def across(self, inverse):
        check(lib.zk_dft_across(self.ctx._h, (self.src if inverse else self.b)._h, self.a._h, self.log_w, int(inverse)))
        self._out = self.a

    def send_tensor(self):
        return self._ta

    def recv_tensor(self):
        return self._tb

    def result(self):
        return self._out
