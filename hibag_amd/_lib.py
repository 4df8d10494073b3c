"""ctypes binding of ``libhibag_hip.so`` (C ABI: ``include/hibag_hip.h``).

The library is the product; this module only declares its prototypes and turns
its error codes into exceptions.  There is no fallback: if the shared object is
missing, or there is no MI355X, compute entry points raise.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# HIBAG_HIP_LIBRARY selects another build of the same library (tuning experiments)
LIB_PATH = os.environ.get("HIBAG_HIP_LIBRARY") or os.path.join(CSRC, "libhibag_hip.so")

K_PACK, K_TOTAL, K_ACCUM, K_FINISH, K_COUNT = 0, 1, 2, 3, 4
KERNEL_NAMES = {K_PACK: "pack", K_TOTAL: "total", K_ACCUM: "accum", K_FINISH: "finish"}

EXPORTS = [
    "hibag_hip_abi_version", "hibag_hip_last_error", "hibag_hip_device_count", "hibag_hip_set_device", "hibag_hip_get_device",
    "hibag_hip_set_kernel_target", "hibag_hip_model_new", "hibag_hip_model_add_classifier",
    "hibag_hip_model_add_classifier_packed", "hibag_hip_model_finalize", "hibag_hip_model_free",
    "hibag_hip_model_n_hla", "hibag_hip_model_n_snp", "hibag_hip_model_n_classifier",
    "hibag_hip_model_pair_evals", "hibag_hip_model_stored_cells", "hibag_hip_model_second_pass_pairs", "hibag_hip_model_mutation_table", "hibag_hip_predict",
    "hibag_hip_predict_device", "hibag_hip_model_set_snp_weights", "hibag_hip_predict_partial_device",
    "hibag_hip_finish_device", "hibag_hip_set_timing", "hibag_hip_get_timing", "hibag_hip_reset_timing",
    "hibag_hip_gpu_ext_proc", "hibag_hip_bed_flag", "hibag_hip_conv_bed", "hibag_hip_predict_bed", "hibag_hip_predict_mapped", "hibag_hip_predict_mapped_device",
    "hibag_hip_predict_snp_major", "hibag_hip_predict_snp_major_device",
    "hibag_hip_trainer_new", "hibag_hip_trainer_free", "hibag_hip_trainer_set_rng", "hibag_hip_trainer_set_seed",
    "hibag_hip_trainer_new_classifiers", "hibag_hip_trainer_n_classifier", "hibag_hip_trainer_classifier_dims",
    "hibag_hip_trainer_classifier_get", "hibag_hip_trainer_set_threads", "hibag_hip_trainer_threads", "hibag_hip_trainer_set_em_mode",
    "hibag_hip_trainer_set_shared", "hibag_hip_train_set_thread_budget", "hibag_hip_train_combine_stats", "hibag_hip_train_combine_times",
    "hibag_hip_model_status", "hibag_hip_model_clear_status", "hibag_hip_model_handover_faults",
    "hibag_hip_test_inject_handover_fault", "hibag_hip_model_engine", "hibag_hip_model_replicate",
    "hibag_hip_multi_slice", "hibag_hip_predict_multi", "hibag_hip_model_device",
    "hibag_hip_shard_bounds", "hibag_hip_model_shard", "hibag_hip_model_batch_limit", "hibag_hip_shard_group_new",
    "hibag_hip_shard_group_free", "hibag_hip_shard_group_ranks", "hibag_hip_shard_group_allreduces", "hibag_hip_rccl_version",
    "hibag_hip_shard_group_predict", "hibag_hip_predict_multi_sharded", "hibag_hip_measure_issue_costs",
    "hibag_hip_test_time_avg_prob", "hibag_hip_test_read_diag", "hibag_hip_plugin_degraded_calls",
]


class HibagHipError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(message or f"libhibag_hip error {code}")
        self.code = code


def build(force: bool = False) -> str:
    """Compile the HIP library for gfx950 (``make -C hibag_amd/csrc``)."""
    cmd = ["make", "-C", CSRC, "-s", "-j2"] + (["-B"] if force else [])
    subprocess.check_call(cmd)
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    """Load the library; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C hibag_amd/csrc`.  hibag_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, dbl = C.c_void_p, C.c_int, C.c_int64, C.c_double
    L.hibag_hip_abi_version.restype = i32
    L.hibag_hip_last_error.restype = C.c_char_p
    L.hibag_hip_device_count.restype = i32
    L.hibag_hip_set_device.argtypes = [i32]
    L.hibag_hip_set_kernel_target.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
    L.hibag_hip_model_new.argtypes = [i32, i32]
    L.hibag_hip_model_new.restype = vp
    L.hibag_hip_model_add_classifier.argtypes = [vp, i32, vp, i32, vp, vp, C.POINTER(C.c_char_p)]
    L.hibag_hip_model_add_classifier_packed.argtypes = [vp, i32, vp, i32, vp, vp, vp]
    L.hibag_hip_model_set_snp_weights.argtypes = [vp, vp]
    L.hibag_hip_model_finalize.argtypes = [vp]
    L.hibag_hip_model_free.argtypes = [vp]
    L.hibag_hip_model_free.restype = None
    for f in (L.hibag_hip_model_n_hla, L.hibag_hip_model_n_snp, L.hibag_hip_model_n_classifier, L.hibag_hip_model_device):
        f.argtypes = [vp]
        f.restype = i32
    L.hibag_hip_model_pair_evals.argtypes = [vp]
    L.hibag_hip_model_pair_evals.restype = i64
    L.hibag_hip_model_stored_cells.argtypes = [vp]
    L.hibag_hip_model_stored_cells.restype = i64
    L.hibag_hip_model_second_pass_pairs.argtypes = [vp]
    L.hibag_hip_model_second_pass_pairs.restype = i64
    L.hibag_hip_model_mutation_table.argtypes = [vp, vp]
    L.hibag_hip_predict.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp, vp, vp]
    L.hibag_hip_predict_device.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp]
    L.hibag_hip_predict_partial_device.argtypes = [vp, vp, i32, vp, vp]
    L.hibag_hip_finish_device.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp, vp, vp]
    L.hibag_hip_set_timing.argtypes = [vp, i32]
    L.hibag_hip_get_timing.argtypes = [vp, i32, C.POINTER(dbl), C.POINTER(i64)]
    L.hibag_hip_reset_timing.argtypes = [vp]
    L.hibag_hip_gpu_ext_proc.restype = vp
    L.hibag_hip_trainer_new.argtypes = [i32, i32, vp, i32, vp, vp]
    L.hibag_hip_trainer_new.restype = vp
    L.hibag_hip_trainer_free.argtypes = [vp]
    L.hibag_hip_trainer_free.restype = None
    L.hibag_hip_trainer_set_rng.argtypes = [vp, vp, vp]
    L.hibag_hip_trainer_set_seed.argtypes = [vp, C.c_uint32]
    L.hibag_hip_trainer_new_classifiers.argtypes = [vp, i32, i32, i32, i32, i32]
    L.hibag_hip_trainer_n_classifier.argtypes = [vp]
    L.hibag_hip_trainer_set_threads.argtypes = [vp, i32]
    L.hibag_hip_trainer_threads.argtypes = [vp]
    L.hibag_hip_trainer_set_em_mode.argtypes = [vp, i32]
    L.hibag_hip_trainer_set_shared.argtypes = [vp, i32]
    L.hibag_hip_train_set_thread_budget.argtypes = [i32]
    L.hibag_hip_train_combine_stats.argtypes = [C.POINTER(i64), C.POINTER(i64), i32]
    L.hibag_hip_train_combine_times.argtypes = [C.POINTER(dbl), i32]
    L.hibag_hip_trainer_classifier_dims.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(i32)]
    L.hibag_hip_trainer_classifier_get.argtypes = [vp, i32, vp, vp, vp, vp, vp, C.POINTER(dbl)]
    L.hibag_hip_bed_flag.argtypes = [C.c_char_p]
    L.hibag_hip_conv_bed.argtypes = [C.c_char_p, i32, i32, i32, vp, vp]
    L.hibag_hip_predict_bed.argtypes = [vp, C.c_char_p, i32, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp]
    L.hibag_hip_predict_mapped.argtypes = [vp, vp, i32, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp]
    L.hibag_hip_predict_mapped_device.argtypes = [vp, vp, i32, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp]
    L.hibag_hip_predict_snp_major.argtypes = [vp, vp, C.c_size_t, i32, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp]
    L.hibag_hip_predict_snp_major_device.argtypes = [vp, vp, C.c_size_t, i32, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp]
    L.hibag_hip_model_status.argtypes = [vp]
    L.hibag_hip_model_clear_status.argtypes = [vp]
    L.hibag_hip_model_handover_faults.argtypes = [vp]
    L.hibag_hip_model_handover_faults.restype = i64
    L.hibag_hip_test_inject_handover_fault.argtypes = [vp, i32]
    L.hibag_hip_model_engine.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(i32)]
    L.hibag_hip_model_replicate.argtypes = [vp, i32]
    L.hibag_hip_model_replicate.restype = vp
    L.hibag_hip_multi_slice.argtypes = [i32, i32, i32, C.POINTER(i32), C.POINTER(i32)]
    L.hibag_hip_predict_multi.argtypes = [C.POINTER(vp), i32, vp, i32, i32, vp, vp, vp, vp, vp, vp]
    L.hibag_hip_shard_bounds.argtypes = [i32, i32, i32, C.POINTER(i32), C.POINTER(i32)]
    L.hibag_hip_model_shard.argtypes = [vp, i32, i32, i32]
    L.hibag_hip_model_shard.restype = vp
    L.hibag_hip_model_batch_limit.argtypes = [vp]
    L.hibag_hip_shard_group_new.argtypes = [C.POINTER(vp), i32]
    L.hibag_hip_shard_group_new.restype = vp
    L.hibag_hip_shard_group_free.argtypes = [vp]
    L.hibag_hip_shard_group_free.restype = None
    L.hibag_hip_shard_group_ranks.argtypes = [vp]
    L.hibag_hip_shard_group_allreduces.argtypes = [vp]
    L.hibag_hip_shard_group_allreduces.restype = i64
    L.hibag_hip_rccl_version.restype = i32
    L.hibag_hip_shard_group_predict.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp, vp]
    L.hibag_hip_predict_multi_sharded.argtypes = [C.POINTER(vp), i32, vp, i32, vp, vp, vp, vp, vp, vp]
    L.hibag_hip_measure_issue_costs.argtypes = [C.POINTER(dbl)] * 4
    L.hibag_hip_test_time_avg_prob.argtypes = [vp, vp, i32, i32, i32, vp, vp, C.POINTER(dbl)]
    if hasattr(L, "hibag_hip_test_read_diag"):          # (absent from older builds loaded through HIBAG_HIP_LIBRARY for A/B timings)
        L.hibag_hip_test_read_diag.argtypes = [vp, vp, i32]
        L.hibag_hip_plugin_degraded_calls.restype = i64
    _lib = L
    return L


def check(rc: int) -> None:
    if rc != 0:
        raise HibagHipError(rc, lib().hibag_hip_last_error().decode("utf-8", "replace"))
