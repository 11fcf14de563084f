"""ctypes binding of libquarkaudio_hip.so (declarations mirror include/quarkaudio.h one to one)."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = "libquarkaudio_hip.so"
_lib: Optional[C.CDLL] = None


class QuarkAudioError(RuntimeError):
    """Raised for every non-zero qa_status; carries qa_last_error()."""

    def __init__(self, status: int, message: str):
        super().__init__(f"[qa_status {status}] {message}")
        self.status = status


class qa_tensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("numel", C.c_int64)]


class qa_hcodec_spec(C.Structure):
    _fields_ = [
        ("n_filters", C.c_int32), ("n_ratios", C.c_int32), ("ratios", C.c_int32 * 8), ("dimension", C.c_int32),
        ("enc_heads", C.c_int32), ("enc_layers", C.c_int32), ("sem_in", C.c_int32), ("sem_ch", C.c_int32),
        ("n_sem_strides", C.c_int32), ("sem_strides", C.c_int32 * 4), ("code_dim", C.c_int32),
        ("codebook_size", C.c_int32), ("num_quantizers", C.c_int32), ("dec_dim", C.c_int32), ("dec_inter", C.c_int32),
        ("dec_heads", C.c_int32), ("dec_layers", C.c_int32), ("convnext_layers", C.c_int32), ("n_fft", C.c_int32),
        ("hop", C.c_int32), ("gn_groups", C.c_int32),
        ("adaptive", C.c_int32), ("agg_layers", C.c_int32), ("agg_heads", C.c_int32), ("agg_ff", C.c_int32),
        ("bt_layers", C.c_int32), ("bt_heads", C.c_int32), ("bt_ff", C.c_int32), ("max_tokens_per_group", C.c_int32),
        ("threshold", C.c_float),
        ("version", C.c_int32), ("enc_dim", C.c_int32), ("enc_inter", C.c_int32), ("enc_convnext_layers", C.c_int32),
        ("frame_stride", C.c_int32), ("tr_inter_cap", C.c_int32), ("causal", C.c_int32),
        ("agg_causal", C.c_int32), ("agg_context", C.c_int32), ("bt_causal", C.c_int32), ("bt_context", C.c_int32),
    ]


class qa_conv_args(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("gamma", C.c_void_p), ("residual", C.c_void_p),
        ("gate", C.c_void_p), ("y", C.c_void_p),
        ("B", C.c_int64), ("T_in", C.c_int64), ("C_in", C.c_int64), ("T_out", C.c_int64), ("N", C.c_int64),
        ("ldx", C.c_int64), ("ldy", C.c_int64), ("ldr", C.c_int64), ("ldg", C.c_int64),
        ("ksize", C.c_int32), ("stride", C.c_int32), ("pad_left", C.c_int32), ("pad_right", C.c_int32),
        ("pad_mode", C.c_int32), ("prologue", C.c_int32), ("act", C.c_int32), ("post_act", C.c_int32),
        ("in_rep", C.c_int32),
    ]


class qa_lm_spec(C.Structure):
    _fields_ = [
        ("hidden", C.c_int32), ("n_layers", C.c_int32), ("n_heads", C.c_int32), ("intermediate", C.c_int32),
        ("global_size", C.c_int32), ("semantic_size", C.c_int32), ("feats_dim", C.c_int32), ("num_tasks", C.c_int32),
        ("rope_theta", C.c_float), ("rms_eps", C.c_float),
    ]


class qa_mimi_spec(C.Structure):
    _fields_ = [("d_model", C.c_int32), ("num_heads", C.c_int32), ("num_layers", C.c_int32), ("dim_feedforward", C.c_int32),
                ("causal", C.c_int32), ("context", C.c_int32)]


class qa_bicodec_spec(C.Structure):
    _fields_ = [
        ("latent_dim", C.c_int32), ("codebook_size", C.c_int32), ("codebook_dim", C.c_int32), ("spk_latent_dim", C.c_int32),
        ("token_num", C.c_int32), ("n_levels", C.c_int32), ("levels", C.c_int32 * 8), ("vocos_dim", C.c_int32),
        ("vocos_inter", C.c_int32), ("vocos_layers", C.c_int32), ("gen_channels", C.c_int32), ("n_rates", C.c_int32),
        ("rates", C.c_int32 * 8), ("kernel_sizes", C.c_int32 * 8),
    ]


class qa_ssl_spec(C.Structure):
    _fields_ = [
        ("n_conv", C.c_int32), ("conv_dim", C.c_int32 * 8), ("conv_kernel", C.c_int32 * 8), ("conv_stride", C.c_int32 * 8),
        ("conv_bias", C.c_int32), ("feat_norm_layer", C.c_int32), ("hidden", C.c_int32), ("n_layers", C.c_int32),
        ("n_heads", C.c_int32), ("intermediate", C.c_int32), ("stable_layer_norm", C.c_int32), ("pos_kernel", C.c_int32),
        ("pos_groups", C.c_int32), ("pad", C.c_int32), ("n_select", C.c_int32), ("select", C.c_int32 * 32),
        ("layer_norm_eps", C.c_float), ("compress_exponent", C.c_float),
        ("rel_pos_buckets", C.c_int32), ("rel_pos_max_distance", C.c_int32),
    ]


# every symbol include/quarkaudio.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "qa_version": (C.c_int, []),
    "qa_last_error": (C.c_char_p, []),
    "qa_device_count": (C.c_int, []),
    "qa_hcodec_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(qa_hcodec_spec), C.POINTER(qa_tensor), C.c_int64, C.c_int]),
    "qa_hcodec_destroy": (None, [C.c_void_p]),
    "qa_hcodec_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64,
                                   C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "qa_hcodec_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "qa_hcodec_encode_adaptive": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64,
                                            C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_float,
                                            C.c_void_p]),
    "qa_codes_check": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_int64), C.c_void_p]),
    "qa_codes_check_async": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "qa_resample_length": (C.c_int64, [C.c_int64, C.c_int32, C.c_int32]),
    "qa_resample": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "qa_hcodec_adaptive_frames": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_int64), C.c_void_p]),
    "qa_hcodec_decode_adaptive": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p,
                                            C.c_void_p]),
    "qa_hcodec_enable_taps": (C.c_int, [C.c_void_p, C.c_int]),
    "qa_hcodec_tap": (C.c_int64, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "qa_rvq_search": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                C.c_void_p, C.c_void_p]),
    "qa_rvq_lookup": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                C.c_void_p]),
    "qa_conv1d_cl": (C.c_int, [C.POINTER(qa_conv_args), C.c_void_p]),
    "qa_rownorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_int32, C.c_void_p]),
    "qa_dwconv_cl": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                             C.c_float, C.c_void_p]),
    "qa_sconv_geometry": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "qa_resolve_frame": (C.c_int64, [C.c_int64, C.c_int64, C.c_int32, C.c_int32]),
    "qa_profile_begin": (C.c_int, []),
    "qa_profile_end": (C.c_int, [C.POINTER(C.c_double), C.c_int32]),
    "qa_profile_begin_ex": (C.c_int, [C.c_int32]),
    "qa_profile_hbm_kinds": (C.c_int, []),
    "qa_profile_hbm_name": (C.c_char_p, [C.c_int32]),
    "qa_debug_lstm_stats": (C.c_int, [C.c_int32, C.POINTER(C.c_int64)]),
    "qa_profile_end_hbm": (C.c_int, [C.POINTER(C.c_double), C.c_int32]),
    "qa_set_serial": (C.c_int, [C.c_int32]),
    "qa_knob_count": (C.c_int, []),
    "qa_knob_info": (C.c_int, [C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_char_p)]),
    "qa_set_knob": (C.c_int, [C.c_char_p, C.c_int64]),
    "qa_get_knob": (C.c_int, [C.c_char_p, C.POINTER(C.c_int64)]),
    "qa_mimi_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(qa_mimi_spec), C.POINTER(qa_tensor), C.c_int64, C.c_char_p, C.c_int]),
    "qa_mimi_destroy": (None, [C.c_void_p]),
    "qa_mimi_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "qa_mimi_stream_begin": (C.c_int, [C.c_void_p, C.c_int64]),
    "qa_mimi_stream_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "qa_mimi_stream_reset": (C.c_int, [C.c_void_p]),
    "qa_mimi_stream_end": (C.c_int, [C.c_void_p]),
    "qa_mimi_stream_offset": (C.c_int64, [C.c_void_p]),
    "qa_ssl_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(qa_ssl_spec), C.POINTER(qa_tensor), C.c_int64, C.c_int]),
    "qa_ssl_destroy": (None, [C.c_void_p]),
    "qa_ssl_frames": (C.c_int64, [C.c_void_p, C.c_int64]),
    "qa_ssl_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "qa_bicodec_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(qa_bicodec_spec), C.POINTER(qa_tensor), C.c_int64, C.c_int]),
    "qa_bicodec_destroy": (None, [C.c_void_p]),
    "qa_bicodec_hop": (C.c_int64, [C.c_void_p]),
    "qa_bicodec_detokenize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "qa_bicodec_enable_taps": (C.c_int, [C.c_void_p, C.c_int]),
    "qa_bicodec_tap": (C.c_int64, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "qa_lm_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(qa_lm_spec), C.POINTER(qa_tensor), C.c_int64, C.c_int]),
    "qa_lm_destroy": (None, [C.c_void_p]),
    "qa_lm_generate": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64,
                                 C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_float, C.c_void_p, C.c_void_p,
                                 C.c_void_p]),
    "qa_lm_generate_sampled": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64,
                                         C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_float, C.c_uint64, C.c_void_p,
                                         C.c_void_p, C.c_void_p]),
    "qa_sample_logits": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_float, C.c_float, C.c_int32,
                                   C.c_uint64, C.c_void_p, C.c_void_p]),
}


def lib_path() -> str:
    # QA_LIBRARY: kernel-tuning experiments load an alternative BUILD of this same library (tools/variants.py)
    return os.environ.get("QA_LIBRARY") or os.path.join(_HERE, _LIB_NAME)


def load_library() -> C.CDLL:
    """Load the shared library (once).  Raises - never falls back - when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise QuarkAudioError(-2, f"{path} not found: build it with `python -m unified_audio_amd.build` "
                                  "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the ABI drifted
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int) -> None:
    if status != 0:
        raise QuarkAudioError(status, load_library().qa_last_error().decode("utf-8", "replace"))


def set_knob(name: str, value: int) -> int:
    """Set a tuning knob (csrc/knobs.h) for this process; returns the previous value."""
    lib = load_library()
    old = C.c_int64()
    check(lib.qa_get_knob(name.encode(), C.byref(old)))
    check(lib.qa_set_knob(name.encode(), int(value)))
    return int(old.value)


def get_knob(name: str) -> int:
    v = C.c_int64()
    check(load_library().qa_get_knob(name.encode(), C.byref(v)))
    return int(v.value)


def knobs() -> dict:
    """name -> (value, default, doc) of every knob the library has."""
    lib = load_library()
    out = {}
    for i in range(lib.qa_knob_count()):
        n, d = C.c_char_p(), C.c_char_p()
        v, dv = C.c_int64(), C.c_int64()
        check(lib.qa_knob_info(i, C.byref(n), C.byref(v), C.byref(dv), C.byref(d)))
        out[n.value.decode()] = (int(v.value), int(dv.value), d.value.decode())
    return out


def require_device() -> None:
    if load_library().qa_device_count() <= 0:
        raise QuarkAudioError(-2, "no HIP device visible: the quarkaudio hot path only runs on an MI355X (gfx950)")


def tensor_table(state_dict):
    """state_dict (name -> torch tensor) -> (qa_tensor array, keep-alive list).  Floating-point tensors of any precision
    (fp16 / bf16 / fp64 checkpoints) are converted to fp32 on the host - the path computes in fp32 like the reference;
    integer / bool buffers (e.g. `initted`, `num_batches_tracked`) are not weights and are skipped."""
    import torch

    keep, names = [], []
    items = [(k, v) for k, v in state_dict.items() if torch.is_tensor(v) and v.is_floating_point()]
    arr = (qa_tensor * len(items))()
    for i, (k, v) in enumerate(items):
        t = v.detach().to(device="cpu", dtype=torch.float32).contiguous()
        nm = k.encode()
        keep.append(t)
        names.append(nm)
        arr[i].name = nm
        arr[i].data = t.data_ptr()
        arr[i].numel = t.numel()
    return arr, len(items), (keep, names)
