"""Loader for libcu2rec_amd.so (the C ABI in include/cu2rec_amd.h).

The library is built in-tree by `make -C cu2rec_amd/csrc` (also by __graft_entry__.build()).
There is no Python or CPU fallback for the hot path: if the shared object is missing, importing
fails with instructions, and if no GPU is present the compute entry points return
CU2REC_ENODEVICE, surfaced here as Cu2recError.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# CU2REC_AMD_LIB: load another build of the same library (ablation / tuning builds)
LIB_PATH = os.environ.get("CU2REC_AMD_LIB") or os.path.join(HERE, "libcu2rec_amd.so")


class Cu2recError(RuntimeError):
    """A non-zero cu2rec_status; the text is cu2rec_last_error() (the reference throws
    std::runtime_error from CHECK_CUDA, util.h:27-34)."""

    def __init__(self, status, message):
        super().__init__("cu2rec_amd status %d: %s" % (status, message))
        self.status = status


class Config(C.Structure):
    """config::Config (config.h:20-58); the first nine fields are the config file's fields."""
    _fields_ = [("cur_iterations", C.c_int), ("total_iterations", C.c_int), ("n_factors", C.c_int),
                ("learning_rate", C.c_float), ("seed", C.c_int), ("P_reg", C.c_float), ("Q_reg", C.c_float),
                ("user_bias_reg", C.c_float), ("item_bias_reg", C.c_float), ("is_train", C.c_int),
                ("n_threads", C.c_int), ("check_error", C.c_int), ("patience", C.c_float),
                ("learning_rate_decay", C.c_float)]

    def hyper(self):
        return Hyper(self.learning_rate, self.P_reg, self.Q_reg, self.user_bias_reg, self.item_bias_reg)


class Hyper(C.Structure):
    _fields_ = [("learning_rate", C.c_float), ("P_reg", C.c_float), ("Q_reg", C.c_float),
                ("user_bias_reg", C.c_float), ("item_bias_reg", C.c_float)]


class TrainStats(C.Structure):
    _fields_ = [("seconds_total", C.c_double), ("seconds_sgd", C.c_double), ("updates", C.c_double),
                ("n_checks", C.c_int), ("last_train_mae", C.c_float), ("last_train_rmse", C.c_float),
                ("last_test_mae", C.c_float), ("last_test_rmse", C.c_float)]


class CommInfo(C.Structure):
    """cu2rec_comm_info_t: what RCCL itself reports about the attached communicator."""
    _fields_ = [("rank", C.c_int), ("nranks", C.c_int), ("rccl_nranks", C.c_int), ("rccl_rank", C.c_int), ("rccl_device", C.c_int),
                ("rccl_version", C.c_int), ("is_callback", C.c_int)]


_P = C.c_void_p
_ip, _fp, _dp = C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_double)

# name -> (restype, argtypes); every symbol include/cu2rec_amd.h declares
SIGNATURES = {
    "cu2rec_last_error": (C.c_char_p, []),
    "cu2rec_version": (C.c_int, []),
    "cu2rec_device_count": (C.c_int, []),
    "cu2rec_set_device": (C.c_int, [C.c_int]),
    "cu2rec_config_default": (C.c_int, [C.POINTER(Config)]),
    "cu2rec_config_read": (C.c_int, [C.c_char_p, C.POINTER(Config)]),
    "cu2rec_config_write": (C.c_int, [C.c_char_p, C.POINTER(Config)]),
    "cu2rec_config_print": (C.c_int, [C.POINTER(Config)]),
    "cu2rec_ratings_read_csv": (C.c_int, [C.c_char_p, C.POINTER(_P)]),
    "cu2rec_ratings_save_binary": (C.c_int, [_P, C.c_char_p]),
    "cu2rec_ratings_load_binary": (C.c_int, [C.c_char_p, C.POINTER(_P)]),
    "cu2rec_ratings_info": (C.c_int, [_P, _ip, _ip, _ip, _fp]),
    "cu2rec_ratings_view": (C.c_int, [_P, C.POINTER(_ip), C.POINTER(_ip), C.POINTER(_fp)]),
    "cu2rec_ratings_free": (None, [_P]),
    "cu2rec_csr_build": (C.c_int, [_P, C.c_int, _P, _P, _P]),
    "cu2rec_init_normal": (C.c_int, [_P, C.c_size_t, C.c_int, C.c_float, C.c_float, C.c_int]),
    "cu2rec_write_csv": (C.c_int, [C.c_char_p, _P, C.c_int, C.c_int]),
    "cu2rec_write_component": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, _P, C.c_int, C.c_int, C.c_int]),
    "cu2rec_read_array": (C.c_int, [C.c_char_p, C.POINTER(_fp), _ip, _ip]),
    "cu2rec_free": (None, [_P]),
    "cu2rec_sampler_draw": (C.c_uint32, [C.c_uint64, C.c_uint64, C.c_uint64]),
    "cu2rec_sampler_index": (C.c_int, [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int]),
    "cu2rec_sgd_update": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, C.c_int, _P, C.c_int, _P, _P, C.c_float,
                                    C.c_int, C.POINTER(Hyper), C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "cu2rec_model_item_stride": (C.c_int, [_P]),
    "cu2rec_hogwild_resident_geometry": (C.c_int, [C.c_int, C.c_int, C.c_int, _ip, _ip, _ip]),
    "cu2rec_hogwild_resident_streamed_rows": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "cu2rec_sgd_update_ex": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, C.c_int, _P, C.c_int, _P, _P, C.c_float,
                                       C.c_int, C.POINTER(Hyper), C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, _P,
                                       _P]),
    "cu2rec_sgd_update_pingpong": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, C.c_int, _P, _P, C.c_int, _P, _P, _P, _P,
                                             C.c_float, C.c_int, C.POINTER(Hyper), C.c_uint64, C.c_uint64, C.c_int, C.c_int,
                                             C.c_int, C.c_int, _ip, _P]),
    "cu2rec_sample_pairs_bytes": (C.c_size_t, [C.c_int]),
    "cu2rec_sample_pairs_build": (C.c_int, [_P, _P, C.c_int, _P, _P]),
    "cu2rec_hogwild_resident": (C.c_int, [C.c_int]),
    "cu2rec_check_faults": (C.c_int, []),
    "cu2rec_hogwild_resident_refusals": (C.c_int, []),
    "cu2rec_hogwild_resident_plan": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "cu2rec_schedule_create": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "cu2rec_schedule_destroy": (None, [_P]),
    "cu2rec_sgd_update_ordered": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, _P, C.c_int, _P, C.c_int, _P, _P, C.c_float,
                                            C.c_int, C.POINTER(Hyper), C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, _P]),
    "cu2rec_sgd_update_blocksolve": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, _P, C.c_int, _P, C.c_int, _P, _P, C.c_float,
                                               C.c_int, C.POINTER(Hyper), C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, _P]),
    "cu2rec_blocksolve_min_rate": (C.c_float, [C.c_float]),
    "cu2rec_blocksolve_lookahead_blocks": (C.c_int, [C.c_int]),
    "cu2rec_blocksolve_topology": (C.c_int, [C.c_char_p, C.c_size_t]),
    "cu2rec_csr_blocksolve_items": (C.c_int, [_P]),
    "cu2rec_debug_blocksolve_stamps": (C.c_int, [_P, C.c_int]),
    "cu2rec_loss_workspace_bytes": (C.c_size_t, []),
    "cu2rec_loss": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, C.c_int, _P, C.c_int, _P, _P, C.c_float, C.c_int, _P,
                              _P, _dp, _dp, _fp, _fp, _P]),
    "cu2rec_error_metrics": (C.c_int, [_P, C.c_int, _P, _fp, _fp, _P]),
    "cu2rec_csr_create": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, _P, _P, C.POINTER(_P)]),
    "cu2rec_csr_info": (C.c_int, [_P, _ip, _ip, _ip]),
    "cu2rec_csr_device_ptrs": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P)]),
    "cu2rec_csr_destroy": (None, [_P]),
    "cu2rec_model_create": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, C.c_float, C.POINTER(_P)]),
    "cu2rec_model_info": (C.c_int, [_P, _ip, _ip, _ip, _ip, _fp]),
    "cu2rec_model_device_ptrs": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P)]),
    "cu2rec_model_download": (C.c_int, [_P, _P, _P, _P, _P]),
    "cu2rec_model_destroy": (None, [_P]),
    "cu2rec_model_sgd": (C.c_int, [_P, _P, C.POINTER(Hyper), C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int]),
    "cu2rec_model_loss": (C.c_int, [_P, _P, _dp, _dp, _fp, _fp]),
    "cu2rec_model_scores": (C.c_int, [_P, _P, _P]),
    "cu2rec_model_scores_host": (C.c_int, [_P, _P]),
    "cu2rec_model_recommend": (C.c_int, [_P, _P, C.c_int, _P, _P]),
    "cu2rec_train": (C.c_int, [_P, _P, C.POINTER(Config), _P, C.c_int, C.c_int, _P, C.POINTER(TrainStats)]),
    "cu2rec_shard_plan": (C.c_int, [C.c_int, C.c_int, _P]),
    "cu2rec_comm_unique_id": (C.c_int, [_P]),
    "cu2rec_comm_create": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(_P)]),
    "cu2rec_comm_from_nccl": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(_P)]),
    "cu2rec_comm_from_callback": (C.c_int, [_P, _P, C.c_int, C.c_int, C.POINTER(_P)]),
    "cu2rec_comm_destroy": (None, [_P]),
    "cu2rec_comm_info": (C.c_int, [_P, C.POINTER(CommInfo)]),
    "cu2rec_shard_job_create": (C.c_int, [_P, _P, _P, C.c_int, _P, C.POINTER(_P)]),
    "cu2rec_shard_job_destroy": (None, [_P]),
    "cu2rec_shard_job_run": (C.c_int, [_P, C.POINTER(Hyper), C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int]),
    "cu2rec_shard_job_exchange": (C.c_int, [_P]),
    "cu2rec_shard_job_loss": (C.c_int, [_P, _P, _dp, _dp, _dp, _fp, _fp]),
    "cu2rec_shard_job_info": (C.c_int, [_P, _ip, _ip, _dp, _dp, C.POINTER(C.c_size_t)]),
    "cu2rec_shard_job_exchange_stats": (C.c_int, [_P, _ip, _dp, _dp]),
    "cu2rec_train_sharded": (C.c_int, [_P, _P, C.POINTER(Config), C.c_int, C.c_int, _P, C.POINTER(TrainStats)]),
    "cu2rec_csr_slice": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _ip, _ip]),
    "cu2rec_items_delta_pack": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, _P, _P]),
    "cu2rec_items_delta_apply_overlapped": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, _P, C.c_float, _P]),
    "cu2rec_items_delta_pack_weighted": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, _P, _P]),
    "cu2rec_item_update_rates": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "cu2rec_items_delta_apply": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, _P, C.c_float, _P]),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "cu2rec_amd: %s is missing. Build it with `make -C cu2rec_amd/csrc` (or "
                "`python -c 'import __graft_entry__ as g; g.build()'`). There is no fallback." % LIB_PATH)
        # PyTorch wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's).  If torch is going
        # to be used in this process it must be loaded FIRST, so that this library's DT_NEEDED
        # libamdhip64.so.7 resolves to the runtime torch already holds; loading /opt/rocm's copy first
        # and torch's afterwards puts two HIP/HSA runtimes in one process (torch then sees no GPU).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            if os.environ.get("CU2REC_AMD_LIB_LENIENT") and not hasattr(L, name):
                continue  # A/B timing against an older build of the library (tools/): its newer entry points stay unbound
            fn = getattr(L, name)  # AttributeError here = header / library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status):
    if status != 0:
        raise Cu2recError(status, lib().cu2rec_last_error().decode(errors="replace"))
