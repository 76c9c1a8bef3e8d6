"""ctypes binding of libfakebob_hip.so (include/fakebob_hip.h).

There is no CPU fallback: if the HIP library is missing or cannot be loaded,
importing the native layer raises.  (oracle/ is test infrastructure and is
never imported from here.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FAKEBOB_HIP_LIB: a library built elsewhere (an install prefix, a profiling build); default: the in-tree build
LIB_PATH = os.environ.get("FAKEBOB_HIP_LIB") or os.path.join(_HERE, "lib", "libfakebob_hip.so")

FB_OK = 0
FB_E_ARG, FB_E_HIP, FB_E_STATE, FB_E_NO_VOICED, FB_E_NOMEM, FB_E_LIMIT, FB_E_CALLBACK = -1, -2, -3, -4, -5, -6, -7
TASK = {"OSI": 0, "CSI": 1, "SV": 2}
ATTACK = {"untargeted": 0, "targeted": 1}


class FrontendCfg(C.Structure):
    """fb_frontend_cfg"""
    _fields_ = [
        ("sample_freq", C.c_double), ("frame_length", C.c_int), ("frame_shift", C.c_int),
        ("padded_length", C.c_int), ("num_mel_bins", C.c_int), ("num_ceps", C.c_int),
        ("low_freq", C.c_double), ("high_freq", C.c_double), ("preemph", C.c_double),
        ("cepstral_lifter", C.c_double), ("snip_edges", C.c_int), ("remove_dc", C.c_int),
        ("use_energy", C.c_int), ("raw_energy", C.c_int), ("energy_floor", C.c_double),
        ("vad_energy_threshold", C.c_double), ("vad_energy_mean_scale", C.c_double),
        ("vad_proportion_threshold", C.c_double), ("vad_frames_context", C.c_int),
        ("delta_window", C.c_int), ("delta_order", C.c_int), ("cmn_window", C.c_int), ("text_scores", C.c_int), ("compress_feats", C.c_int),
        ("mfcc_f32", C.c_int),
    ]


class NesParams(C.Structure):
    """fb_nes_params"""
    _fields_ = [
        ("task", C.c_int), ("attack_type", C.c_int), ("adver_thresh", C.c_double),
        ("epsilon", C.c_double), ("max_iter", C.c_int), ("max_lr", C.c_double),
        ("min_lr", C.c_double), ("samples_per_draw", C.c_int), ("sigma", C.c_double),
        ("momentum", C.c_double), ("plateau_length", C.c_int), ("plateau_drop", C.c_double),
        ("threshold", C.c_double), ("target", C.c_int), ("true_label", C.c_int),
        ("seed", C.c_uint64), ("stream", C.c_uint32), ("bits_per_sample", C.c_int),
    ]


class IvectorSystem(C.Structure):
    """fb_ivector_system"""
    _fields_ = [
        ("C", C.c_int), ("D", C.c_int), ("R", C.c_int), ("L", C.c_int), ("S", C.c_int),
        ("lda_cols", C.c_int), ("num_gselect", C.c_int), ("min_post", C.c_double),
        ("prior_offset", C.c_double), ("fg_weights", C.c_void_p), ("fg_means_invcovars", C.c_void_p),
        ("fg_inv_covars", C.c_void_p), ("ie_M", C.c_void_p), ("ie_sigma_inv", C.c_void_p),
        ("mean_vec", C.c_void_p), ("lda", C.c_void_p), ("plda_mean", C.c_void_p),
        ("plda_transform", C.c_void_p), ("plda_psi", C.c_void_p), ("enrolled", C.c_void_p),
        ("z_mean", C.c_void_p), ("z_std", C.c_void_p),
    ]


# fb_score_cb: int (*)(void *ctx, const double *audios, int64_t N, int B, double *scores)
SCORE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int64, C.c_int, C.POINTER(C.c_double))

EXPORTS = [
    "fb_last_error", "fb_version", "fb_device_count", "fb_engine_create", "fb_engine_destroy",
    "fb_default_frontend", "fb_set_frontend", "fb_load_gmm", "fb_load_ivector", "fb_set_system", "fb_num_speakers",
    "fb_score_i16", "fb_score_f64", "fb_system_scores", "fb_get_grad", "fb_attack", "fb_attack_iter_seconds", "fb_get_grad_ext", "fb_attack_ext",
    "fb_estimate_threshold", "fb_debug_noise", "fb_debug_quantize", "fb_debug_mfcc", "fb_debug_feats", "fb_debug_gmm_frames", "fb_debug_iv_active", "fb_debug_iv_gselect", "fb_stats", "fb_gmm_acc_stats", "fb_last_ivectors", "fb_gmm_kernel_mode", "fb_gmm_kernel_variant", "fb_gmm_delta_tiles", "fb_gmm_delta_tiles_f6", "fb_set_fused_chain",
    "fb_bench_gmm_kernel", "fb_bench_nes",
]

_lib = None


class NativeError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "libfakebob_hip: %s (code %d)" % (msg, code))
        self.code = code


def lib():
    """Load libfakebob_hip.so; raise loudly when it is absent (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libfakebob_hip.so not built: run `python -m fakebob_amd.build` "
            "(or __graft_entry__.build()); there is no CPU fallback for the hot path")
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    L.fb_last_error.restype = C.c_char_p
    for name in EXPORTS:
        getattr(L, name)  # AttributeError if the ABI is incomplete
    _lib = L
    return L


def check(rc):
    if rc != FB_OK:
        raise NativeError(rc, lib().fb_last_error().decode("utf-8", "replace"))


def ptr(a, t=C.c_void_p):
    return a.ctypes.data_as(t)
