"""ctypes loader for libzk_amd.so (the gfx950 library behind include/zk_amd.h).

The library is built in-tree by `make -C zk_amd/csrc` (see __graft_entry__.build).  There is no fallback of any
kind: if the shared object is missing this module raises, and creating a context without a gfx950 device fails
with ZK_ERR_NO_DEVICE.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# ZK_AMD_LIB: another build of the SAME sources (the host-sanitizer build of `make -C zk_amd/csrc asan`, tools/run_sanitized.sh; the
# A/B scripts under tools/); never a different implementation -- there is no fallback path
_DEFAULT_LIB = os.path.join(_HERE, "libzk_amd.so")
LIB_PATH = os.environ.get("ZK_AMD_LIB") or _DEFAULT_LIB
LIB_OVERRIDDEN = os.path.realpath(LIB_PATH) != os.path.realpath(_DEFAULT_LIB)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "zk_amd.h")
if LIB_OVERRIDDEN:
    # an override is for builds of THIS tree (sanitizer build, A/B copies under ab_tmp/): anything else is refused, and the redirect is
    # never silent -- a stale variable would otherwise make the tests and bench.py measure a different build than the sources
    import sys as _sys

    _root = os.path.realpath(os.path.dirname(_HERE))
    if not os.path.realpath(LIB_PATH).startswith(_root + os.sep):
        raise ImportError(f"ZK_AMD_LIB={LIB_PATH} is outside {_root}: refused (the override is for builds of this tree only)")
    print(f"zk_amd: ZK_AMD_LIB override active: loading {os.path.realpath(LIB_PATH)}", file=_sys.stderr, flush=True)

c = ctypes
u64p = c.POINTER(c.c_uint64)
u8p = c.POINTER(c.c_uint8)
vpp = c.POINTER(c.c_void_p)


class ZkError(Exception):
    """A negative zk_status; str() is the reference's own Err text where one exists (zk_strerror)."""

    def __init__(self, code, text):
        super().__init__(text)
        self.code = code


def load():
    # ONE HIP runtime per process: the PyTorch-ROCm wheel bundles its own libamdhip64/libhsa-runtime64, and a process
    # that maps both that copy and /opt/rocm's loses the GPU in whichever initialises second (observed on the
    # MI355X box: torch.cuda.is_available() -> False after libzk_amd had initialised ROCm's copy, and vice versa).
    # Importing torch first makes the dynamic linker resolve libzk_amd.so's NEEDED libamdhip64.so.7 to the copy that
    # is already mapped.  Hosts without torch (the C/C++/Rust side) simply get /opt/rocm's runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make -C zk_amd/csrc` (or __graft_entry__.build()); "
            "zk_amd has no CPU fallback"
        )
    return c.CDLL(LIB_PATH)


lib = load()
lib.zk_strerror.restype = c.c_char_p
lib.zk_last_hip_error.restype = c.c_char_p


def declared_symbols():
    """Every function name include/zk_amd.h declares (used by the export test)."""
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(zk_[a-z0-9_]+)\s*\(", text)))


def check(rc):
    if rc < 0:
        msg = lib.zk_strerror(rc).decode()
        if rc in (-23, -28):   # ZK_ERR_HIP / ZK_ERR_COMM: the runtime's (HIP's, RCCL's) own text
            msg += ": " + lib.zk_last_hip_error().decode()
        raise ZkError(rc, msg)
    return rc


# host-transport callbacks of zk_comm_create_host (include/zk_amd.h)
HOST_ALLREDUCE = c.CFUNCTYPE(c.c_int32, c.c_void_p, u64p, c.c_uint64)
HOST_ALLGATHER = c.CFUNCTYPE(c.c_int32, c.c_void_p, u64p, c.c_uint64, u64p)
HOST_ALLTOALL = c.CFUNCTYPE(c.c_int32, c.c_void_p, u64p, u64p, c.c_uint64)

# explicit signatures (everything returns int32 status unless set above)
_sig = {
    "zk_device_count": [c.POINTER(c.c_int32)],
    "zk_ctx_create": [c.c_int32, c.c_int32, vpp],
    "zk_ctx_destroy": [c.c_void_p],
    "zk_ctx_synchronize": [c.c_void_p],
    "zk_ctx_set_stream": [c.c_void_p, c.c_void_p],
    "zk_ctx_use_own_stream": [c.c_void_p],
    "zk_ctx_field": [c.c_void_p, c.POINTER(c.c_int32)],
    "zk_ctx_trim": [c.c_void_p],
    "zk_field_modulus": [c.c_int32, u64p],
    "zk_field_two_adicity": [c.c_int32, c.POINTER(c.c_int32)],
    "zk_field_root_of_unity": [c.c_int32, c.c_uint64, u64p],
    "zk_fe_from_u64": [c.c_int32, c.c_uint64, u64p],
    "zk_fe_from_canonical": [c.c_int32, u64p, u64p],
    "zk_fe_to_canonical": [c.c_int32, u64p, u64p],
    "zk_fe_from_be_bytes_mod_order": [c.c_int32, c.c_char_p, c.c_size_t, u64p],
    "zk_mle_upload": [c.c_void_p, c.c_uint64, u64p, c.c_uint64, vpp],
    "zk_mle_alloc": [c.c_void_p, c.c_uint64, vpp],
    "zk_mle_fill_random": [c.c_void_p, c.c_void_p, c.c_uint64, c.c_uint64],
    "zk_mle_clone": [c.c_void_p, c.c_void_p, vpp],
    "zk_mle_free": [c.c_void_p, c.c_void_p],
    "zk_mle_n_vars": [c.c_void_p, u64p],
    "zk_mle_download": [c.c_void_p, c.c_void_p, u64p],
    "zk_mle_device_ptr": [c.c_void_p, vpp],
    "zk_mle_equal": [c.c_void_p, c.c_void_p, c.c_void_p, c.POINTER(c.c_int32)],
    "zk_mle_partial_evaluate": [c.c_void_p, c.c_void_p, c.c_uint64, u64p, c.c_uint64, vpp],
    "zk_mle_fold_into": [c.c_void_p, c.c_void_p, u64p, c.c_void_p],
    "zk_mle_evaluate": [c.c_void_p, c.c_void_p, u64p, c.c_uint64, u64p],
    "zk_mle_to_bytes": [c.c_void_p, c.c_void_p, u8p],
    "zk_mle_partial_evaluate_host": [c.c_void_p, c.c_uint64, u64p, c.c_uint64, c.c_uint64, u64p, c.c_uint64, u64p],
    "zk_coeff_to_evaluation": [c.c_void_p, c.c_uint64, u64p, u64p, c.c_uint64, vpp],
    "zk_product_check": [vpp, c.c_uint64],
    "zk_prod_reduce": [c.c_void_p, vpp, c.c_uint64, vpp],
    "zk_product_evaluate": [c.c_void_p, vpp, c.c_uint64, u64p, c.c_uint64, u64p],
    "zk_round_sums": [c.c_void_p, vpp, c.c_uint64, c.c_uint32, u64p],
    "zk_transcript_new": [vpp],
    "zk_transcript_free": [c.c_void_p],
    "zk_transcript_append": [c.c_void_p, c.c_char_p, c.c_size_t],
    "zk_transcript_sample_field_element": [c.c_void_p, c.c_int32, u64p],
    "zk_transcript_sample_n_field_elements": [c.c_void_p, c.c_int32, c.c_uint64, u64p],
    "zk_transcript_sample_challenge": [c.c_void_p, c.c_char_p],
    "zk_keccak256": [c.c_char_p, c.c_size_t, c.c_char_p],
    "zk_sumcheck_prove": [c.c_void_p, vpp, c.c_uint64, c.c_uint32, u64p, c.c_int32, c.c_int32, u64p, u64p],
    "zk_sumcheck_prove_batch": [c.c_void_p, c.c_uint64, vpp, c.c_uint64, c.c_uint32, u64p, c.c_int32, u64p, u64p],
    "zk_batch_last_stats": [u64p, u64p],
    "zk_sumcheck_prove_host": [c.c_void_p, c.POINTER(u64p), c.c_uint64, c.c_uint64, c.c_uint32, u64p, c.c_int32, u64p, u64p],
    "zk_shard_prover_create": [c.c_void_p, vpp, c.c_uint64, c.c_uint32, u64p, c.c_uint32, vpp],
    "zk_shard_prover_destroy": [c.c_void_p],
    "zk_shard_prover_rounds": [c.c_void_p, u64p, u64p, u64p],
    "zk_shard_prover_lanes_ptr": [c.c_void_p, vpp, u64p],
    "zk_shard_prover_round_begin": [c.c_void_p],
    "zk_shard_prover_round_finish": [c.c_void_p],
    "zk_shard_prover_tail_ptr": [c.c_void_p, vpp, u64p],
    "zk_shard_prover_tail_rounds": [c.c_void_p, c.c_void_p],
    "zk_shard_prover_results": [c.c_void_p, u64p, u64p],
    "zk_comm_unique_id": [c.c_char_p],
    "zk_comm_create_rccl": [c.c_void_p, c.c_char_p, c.c_uint32, c.c_uint32, vpp],
    "zk_comm_wrap_rccl": [c.c_void_p, c.c_void_p, c.c_uint32, c.c_uint32, vpp],
    "zk_comm_create_host": [c.c_void_p, c.c_uint32, c.c_uint32, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, vpp],
    "zk_comm_destroy": [c.c_void_p],
    "zk_comm_info": [c.c_void_p, c.POINTER(c.c_int32), c.POINTER(c.c_uint32), c.POINTER(c.c_uint32)],
    "zk_shard_prover_run": [c.c_void_p, c.c_void_p, c.c_uint32],
    "zk_shard_prover_run_phases": [c.c_void_p, c.c_void_p, c.c_uint32, c.POINTER(c.c_double)],
    "zk_ntt_sharded": [c.c_void_p, c.c_void_p, c.c_void_p, c.c_int32, c.c_void_p],
    "zk_ctx_device_alloc": [c.c_void_p, c.c_uint64, vpp],
    "zk_ctx_device_free": [c.c_void_p, c.c_void_p, c.c_uint64],
    "zk_ctx_memcpy_dtoh": [c.c_void_p, c.c_void_p, c.c_void_p, c.c_uint64],
    "zk_ctx_memcpy_htod": [c.c_void_p, c.c_void_p, c.c_void_p, c.c_uint64],
    "zk_sumcheck_verify_partial": [c.c_int32, c.c_uint64, c.c_uint32, u64p, u64p, u64p, u64p],
    "zk_sumcheck_verify": [c.c_void_p, vpp, c.c_uint64, c.c_uint64, c.c_uint32, u64p, u64p, c.POINTER(c.c_int32)],
    "zk_sumcheck_verify_partial_lengths": [c.c_int32, c.c_uint64, c.POINTER(c.c_uint32), u64p, u64p, u64p, u64p],
    "zk_sumcheck_verify_lengths": [c.c_void_p, vpp, c.c_uint64, c.c_uint64, c.POINTER(c.c_uint32), u64p, u64p,
                                   c.POINTER(c.c_int32)],
    "zk_sumcheck_prove_terms": [c.c_void_p, vpp, u64p, c.c_uint64, c.c_uint32, u64p, c.c_int32, u64p, u64p, u64p],
    "zk_eq_table": [c.c_void_p, u64p, c.c_uint64, vpp],
    "zk_circuit_create": [c.c_void_p, vpp],
    "zk_circuit_add_layer": [c.c_void_p, c.c_uint64, c.c_uint64, u8p, c.POINTER(c.c_uint32), c.POINTER(c.c_uint32)],
    "zk_circuit_free": [c.c_void_p],
    "zk_circuit_depth": [c.c_void_p, u64p],
    "zk_circuit_layer_dims": [c.c_void_p, c.c_uint64, u64p, u64p],
    "zk_circuit_proof_elems": [c.c_void_p, u64p],
    "zk_gkr_evaluate": [c.c_void_p, c.c_void_p, vpp],
    "zk_gkr_prove": [c.c_void_p, c.c_void_p, u8p, vpp, u64p],
    "zk_gkr_verify": [c.c_void_p, c.c_void_p, c.c_void_p, u8p, u64p],
    "zk_mle_mul_powers": [c.c_void_p, c.c_void_p, u64p, u64p],
    "zk_dft_across": [c.c_void_p, c.c_void_p, c.c_void_p, c.c_uint64, c.c_int32],
    "zk_ntt": [c.c_void_p, c.c_void_p, c.c_int32, c.c_void_p],
    "zk_fft_host": [c.c_void_p, u64p, c.c_uint64, u64p],
    "zk_ifft_host": [c.c_void_p, u64p, c.c_uint64, u64p],
    "zk_fft_internal_host": [c.c_void_p, u64p, c.c_uint64, u64p, u64p],
    "zk_bench_fold": [c.c_void_p, c.c_void_p, u64p, c.c_void_p, c.c_int32, c.POINTER(c.c_double)],
    "zk_bench_fold_samples": [c.c_void_p, c.c_void_p, u64p, c.c_void_p, c.c_int32, c.c_int32, c.POINTER(c.c_double)],
    "zk_bench_ntt": [c.c_void_p, c.c_void_p, c.c_int32, c.c_void_p, c.c_int32, c.POINTER(c.c_double)],
    "zk_bench_prove_partial": [c.c_void_p, c.POINTER(c.c_void_p), c.c_uint64, c.c_uint32, u64p, c.c_int32, c.POINTER(c.c_double)],
    "zk_bench_evaluate": [c.c_void_p, c.c_void_p, u64p, c.c_uint64, c.c_int32, c.POINTER(c.c_double)],
    "zk_bench_evaluate_device": [c.c_void_p, c.c_void_p, u64p, c.c_uint64, c.c_int32, c.POINTER(c.c_double)],
    "zk_bench_modmul": [c.c_void_p, c.c_int32, c.c_int32, c.POINTER(c.c_double)],
    "zk_bench_copy": [c.c_void_p, c.c_uint64, c.c_int32, c.POINTER(c.c_double)],
}
for _name, _args in _sig.items():
    if LIB_OVERRIDDEN and not hasattr(lib, _name):
        continue   # an older A/B build without this entry point: calling it raises AttributeError; the shipped library must have them all
    _f = getattr(lib, _name)
    _f.argtypes = _args
    _f.restype = c.c_int32


def lib_info():
    """which shared object this process computes with (bench.py records it in its JSON line)"""
    return {"path": os.path.realpath(LIB_PATH), "overridden_by_ZK_AMD_LIB": bool(LIB_OVERRIDDEN), "abi": int(lib.zk_abi_version())}
