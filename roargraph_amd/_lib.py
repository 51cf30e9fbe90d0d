"""ctypes binding of librg_hip.so (include/rg.h).

The product has no CPU fallback: if the shared library is missing, or no GPU
is visible when a compute entry point is called, this raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# RG_HIP_LIB selects another BUILD of the same library (the instrumented `make prof` build used by scripts/exp/); it is
# still the HIP path -- there is no CPU fallback behind this switch
LIB_PATH = os.environ.get("RG_HIP_LIB") or os.path.join(HERE, "librg_hip.so")

RG_OK = 0
RG_ERR_IO, RG_ERR_FORMAT, RG_ERR_ARG, RG_ERR_DEVICE, RG_ERR_NOT_ENOUGH, RG_ERR_OOM = -1, -2, -3, -4, -5, -6
METRIC = {"l2": 0, "ip": 1, "mips": 1, "cosine": 4}

# every symbol include/rg.h declares (tests/test_abi.py checks the header against this list and the .so)
SYMBOLS = [
    "rg_last_error", "rg_version", "rg_device_count", "rg_free",
    "rg_fbin_meta", "rg_fbin_load", "rg_fbin_save", "rg_gt_meta", "rg_gt_load", "rg_gt_save", "rg_knn_ids_load",
    "rg_graph_load", "rg_graph_save", "rg_recall", "rg_normalize_rows",
    "rg_index_open", "rg_index_open_multi", "rg_index_open_mem", "rg_index_open_dev", "rg_index_close", "rg_index_info", "rg_index_set", "rg_index_stat", "rg_mem_stats", "rg_mem_stats_ex", "rg_mem_release", "rg_mem_fault_report", "rg_mem_journal_dump", "rg_mem_walk_stress", "rg_index_debug_ell",
    "rg_score_batch", "rg_score_batch_dev", "rg_search", "rg_search_sharded", "rg_search_dev", "rg_search_wait", "rg_search_prepare", "rg_search_reuse_stats",
    "rg_gt_shard_dev", "rg_gt_merge_dev", "rg_groundtruth_mem", "rg_groundtruth",
    "rg_comm_unique_id", "rg_comm_init_rank", "rg_comm_init_local", "rg_comm_uses_rccl", "rg_comm_destroy", "rg_gt_exchange_plan", "rg_groundtruth_rank", "rg_build_roargraph", "rg_build_roargraph_gpu", "rg_build_schedule", "rg_build_prune_debug", "rg_projection_ep", "rg_projection_ep_dev",
]


class RgError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "librg_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` or "
                "`make -C roargraph_amd/csrc`. There is no CPU fallback." % LIB_PATH)
        # One HIP/HSA runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7.  If librg_hip.so were
        # loaded first it would pull in /opt/rocm's copy and torch would then fail to see the GPU, so when torch is
        # installed load it first and let librg_hip.so bind to the runtime that is already resident.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.rg_last_error.restype = C.c_char_p
        L.rg_version.restype = C.c_char_p
        L.rg_recall.restype = C.c_float
        L.rg_free.argtypes = [C.c_void_p]
        L.rg_index_close.argtypes = [C.c_void_p]
        L.rg_comm_destroy.argtypes = [C.c_void_p]
        L.rg_comm_uses_rccl.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def check(rc):
    if rc != RG_OK:
        raise RgError(rc, lib().rg_last_error().decode(errors="replace"))
