"""ctypes binding of libmla_hip.so (C ABI: include/mla_hip.h).

The library is loaded on first use and NEVER substituted: if it is missing or a call
fails, the product path raises -- there is no CPU / PyTorch fallback behind it.
"""

import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MLA_HIP_LIB") or os.path.join(HERE, "libmla_hip.so")   # MLA_HIP_LIB: an alternative build (A/B measurements)
HEADER = os.path.join(os.path.dirname(HERE), "include", "mla_hip.h")

F32, BF16, I16, BF16X3, F64, I32 = 0, 1, 2, 3, 4, 5
E_SHORT = -3

_lib = None


class MlaError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libmla_hip error %d: %s" % (code, msg))
        self.code = code


def declared_symbols():
    """Function names declared in include/mla_hip.h."""
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mla_[a-z0-9_]+)\s*\(", text)))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "%s is missing: build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950). "
                "There is no fallback path." % LIB_PATH)
        # PyTorch ships its own HIP runtime (torch/lib/libamdhip64.so); libmla_hip.so is linked against the same SONAME. torch must
        # be in the process FIRST so that both use one runtime: loaded the other way round the system runtime comes in with this
        # library, torch brings its own, and the second one finds no device ("no ROCm-capable device is detected").
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        L.mla_last_error.restype = ctypes.c_char_p
        L.mla_logmel_table_floats.restype = ctypes.c_int64
        i64, vp, ci = ctypes.c_int64, ctypes.c_void_p, ctypes.c_int
        L.mla_logmel_counts.argtypes = [i64, ctypes.POINTER(i64), ctypes.POINTER(i64)]
        L.mla_logmel_build_tables.argtypes = [vp]
        L.mla_logmel_reference_tables.argtypes = [vp, vp]
        L.mla_logmel_examples.argtypes = [vp, ci, i64, i64, i64, vp, vp, ci, vp]
        cf = ctypes.c_float
        L.mla_stft_magnitude.argtypes = [vp, i64, vp, vp, i64, i64, i64, vp, vp]
        L.mla_mel_log.argtypes = [vp, vp, i64, i64, i64, cf, vp, vp]
        L.mla_dataset_frames.argtypes = [vp, i64, ci, ci, ci, ci, vp, vp]
        L.mla_postprocess.argtypes = [vp, vp, vp, i64, vp, vp]
        L.mla_mono_mix.argtypes = [vp, ci, i64, ci, vp, vp]
        L.mla_split_bf16x3.argtypes = [vp, i64, i64, i64, vp, i64, i64, ci, vp]
        L.mla_merge_bf16x3.argtypes = [vp, i64, i64, i64, i64, vp, vp]
        L.mla_linear_bf16x3.argtypes = [vp, i64, vp, i64, vp, vp, i64, i64, i64, i64, i64, ci, ci, vp]
        L.mla_conv_repack_weights.argtypes = [vp, i64, i64, vp, ci, vp]
        L.mla_convert_f32.argtypes = [vp, vp, i64, ci, vp]
        L.mla_convert_bf16_to_f32.argtypes = [vp, vp, i64, vp]
        L.mla_vggish_conv1.argtypes = [vp, ci, i64, vp, vp, vp, ci, vp]
        L.mla_vggish_conv.argtypes = [ci, vp, vp, vp, vp, i64, ci, vp]
        L.mla_linear.argtypes = [vp, i64, vp, i64, vp, vp, i64, i64, i64, i64, ci, ci, ci, vp]
        L.mla_linear_small.argtypes = [vp, i64, vp, i64, vp, vp, i64, i64, i64, i64, vp]
        L.mla_linear_narrow.argtypes = [vp, i64, vp, i64, vp, vp, i64, i64, i64, i64, vp]
        L.mla_linear_splitk.argtypes = [vp, i64, vp, i64, vp, vp, i64, i64, i64, i64, ci, ci, vp, i64, vp]
        L.mla_linear_ksplit.argtypes = [vp, i64, vp, i64, vp, vp, i64, i64, i64, i64, ci, ci, ci, ci, vp, i64, vp]
        L.mla_bn_stats_workspace_bytes.restype = i64
        L.mla_bn_stats.argtypes = [vp, i64, i64, i64, ci, ci, vp, vp, vp, vp, vp, cf, vp]
        L.mla_bn_apply.argtypes = [vp, i64, vp, i64, i64, i64, ci, ci, vp, vp, vp, vp, cf, ci, vp, cf, vp]
        L.mla_attention_pool.argtypes = [vp, i64, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, cf, vp, i64, vp, vp, vp]
        cd = ctypes.c_double
        L.mla_bn_stats_sums.argtypes = [vp, i64, i64, i64, ci, ci, vp, vp, vp]
        L.mla_bn_stats_finish.argtypes = [vp, ci, cd, vp, vp, vp, vp, cf, vp, vp]
        L.mla_bn_stats_fused.argtypes = [vp, i64, i64, i64, ci, ci, vp, vp, vp, vp, vp, vp, cf, vp, vp]
        L.mla_bn_bwd_sums.argtypes = [vp, i64, vp, i64, vp, i64, ci, cf, i64, i64, ci, ci, vp, vp, cf, vp, vp, vp]
        L.mla_bn_bwd_apply.argtypes = [vp, i64, vp, i64, vp, i64, ci, cf, i64, i64, ci, ci, vp, vp, vp, cf, vp, vp, cd,
                                       vp, i64, ci, vp, vp, vp]
        L.mla_attention_pool_bwd.argtypes = [vp, i64, vp, vp, i64, ci, ci, vp, vp, vp]
        L.mla_linear_small_bwd.argtypes = [vp, i64, vp, i64, vp, i64, i64, i64, i64, vp, i64, vp, vp, vp]
        L.mla_transpose_f32.argtypes = [vp, i64, vp, i64, i64, i64, vp]
        L.mla_col_sum.argtypes = [vp, i64, i64, i64, vp, vp, vp]
        L.mla_axpy.argtypes = [cf, vp, vp, i64, vp]
        L.mla_cross_entropy.argtypes = [vp, i64, vp, i64, ci, cf, vp, vp, i64, vp, vp]
        L.mla_adam_step.argtypes = [vp, vp, vp, vp, i64, cf, cf, cf, cf, i64, vp]
        L.mla_conv3x3.argtypes = [vp, vp, vp, vp, i64, ci, ci, ci, ci, ci, ci, ci, vp]
        L.mla_conv3x3_train.argtypes = [vp, vp, vp, vp, vp, i64, ci, ci, ci, ci, ci, vp]
        L.mla_conv_repack_dgrad.argtypes = [vp, i64, i64, vp, vp]
        L.mla_maxpool2x2.argtypes = [vp, vp, i64, ci, ci, ci, vp]
        L.mla_relu_pool_bwd.argtypes = [vp, vp, vp, i64, ci, ci, ci, ci, vp]
        L.mla_relu_pool_bwd_bias.argtypes = [vp, vp, vp, i64, ci, ci, ci, ci, vp, vp, vp]
        L.mla_relu_pool_bwd_bias_workspace_bytes.restype = ctypes.c_int64
        L.mla_conv_wgrad_workspace_floats.restype = i64
        L.mla_conv_wgrad.argtypes = [vp, vp, i64, ci, ci, ci, ci, vp, i64, vp, vp]
        L.mla_conv1_bwd.argtypes = [vp, vp, vp, vp, i64, vp, vp, vp, vp]
        L.mla_conv_repack_dgrad_bf16.argtypes = [vp, i64, i64, vp, vp]
        L.mla_maxpool2x2_bf16.argtypes = [vp, vp, i64, ci, ci, ci, vp]
        L.mla_relu_pool_bwd_bf16_workspace_bytes.restype = i64
        L.mla_relu_pool_bwd_bf16.argtypes = [vp, ci, vp, ci, vp, i64, ci, ci, ci, ci, vp, vp, vp]
        L.mla_conv_wgrad_bf16.argtypes = [vp, vp, i64, ci, ci, ci, ci, vp, i64, vp, vp]
        L.mla_conv1_bwd_bf16.argtypes = [vp, vp, vp, vp, i64, vp, vp, vp, vp]
        L.mla_conv3x3_train_codes.argtypes = [vp, vp, vp, vp, vp, i64, ci, ci, ci, ci, ci, vp]
        L.mla_pool_bwd_codes_bf16.argtypes = [vp, vp, vp, i64, ci, ci, ci, vp, vp, vp]
        L.mla_conv1_bwd_workspace_floats.argtypes = []
        L.mla_conv1_bwd_workspace_floats.restype = i64
        L.mla_transpose_bf16.argtypes = [vp, i64, vp, i64, i64, i64, vp]
        L.mla_col_sum_bf16.argtypes = [vp, i64, i64, i64, vp, vp, vp]
        L.mla_resample_length.restype = i64
        L.mla_resample_length.argtypes = [i64, cd, cd]
        L.mla_resample.argtypes = [vp, i64, cd, cd, vp, vp, ci, ci, vp, i64, vp]
        u64 = ctypes.c_uint64
        L.mla_dropout_mask.argtypes = [vp, i64, u64, u64, u64, cf, vp]
        L.mla_dropout_mask_dev.argtypes = [vp, i64, u64, u64, vp, u64, cf, vp]
        L.mla_adam_step_dev.argtypes = [vp, vp, vp, vp, i64, cf, cf, cf, vp, vp]
        L.mla_adam_prepare.argtypes = [vp, cf, cf, cf, i64, vp]
        L.mla_counter_add.argtypes = [vp, i64, vp]
        L.mla_comm_unique_id.argtypes = [vp]
        L.mla_comm_init_rank.argtypes = [ctypes.POINTER(vp), ci, vp, ci]
        L.mla_comm_destroy.argtypes = [vp]
        L.mla_comm_count.argtypes = [vp, ctypes.POINTER(ci)]
        L.mla_comm_library_origin.restype = ctypes.c_char_p
        L.mla_allreduce_flat.argtypes = [vp, i64, ci, vp, vp]
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise MlaError(rc, lib().mla_last_error().decode("utf-8", "replace"))


def stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
