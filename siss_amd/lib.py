"""ctypes binding of libsiss_hip.so -- the C-ABI declared in include/siss_hip.h.

The product path has NO CPU fallback: if the library is missing or a launcher returns a
non-zero status, a RuntimeError is raised.
"""
import ctypes as C
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SISS_LIB_PATH") or os.path.join(_HERE, "libsiss_hip.so")   # override: A/B two builds

P, I, L, F = C.c_void_p, C.c_int, C.c_long, C.c_float
IP = C.POINTER(C.c_int)

# name -> argtypes (all launchers return int status unless listed in _RET_LONG)
SIGNATURES = {
    "siss_loss_partials_words": [I, L],
    "siss_mixture_fwd": [P, P, P, I, P, P, P, P, P, F, I, L, P, P, P, P, P, P, P, P, P],
    "siss_mixture_select": [P, P, P, P, I, P, P, P, P, F, I, L, P, P, P, P, P, P, P, P, P],
    "siss_ddpm_step": [P, P, P, P, L, F, F, F, F, F, I, P],
    "siss_loss_bwd_seed": [P, P, P, P, I, P, P, P, P, F, I, L, P, P, P, P, P, P, P, P],
    "siss_mse_bwd_seed": [P, P, I, F, I, L, P, P, P, P, P],
    "siss_opt_partials_words": [],
    "siss_opt_scalars_words": [],
    "siss_grad_norms_scale": [P, P, L, I, F, F, F, F, P, P, P],
    "siss_grad_norm_partials": [P, P, L, P, IP, P],
    "siss_grad_scalars": [P, I, I, F, F, F, F, P, P],
    "siss_recombine_clip_adamw": [P, P, P, P, P, P, P, L, F, F, F, F, F, P, P],
    "siss_cast_f32_bf16": [P, P, L, P],
    "siss_conv_weight_dgrad_layout": [P, P, I, I, I, P],
    "siss_conv_weight_dgrad_multi": [P, P, P, I, I, P],
    "siss_conv_weight_dgrad_multi_bf16": [P, P, P, I, I, P],
    "siss_gemm_nt": [P, L, P, P, L, P, P, L, P, L, I, I, I, I, IP, IP, I, I, I, F, I, L, L, L, P],
    "siss_gemm_nt_qstats": [P, L, P, P, L, P, P, L, P, L, I, I, I, I, IP, IP, I, I, I, F, P, IP, P],
    "siss_gemm_nt_alpha_cols": [P, L, P, P, L, P, P, L, I, I, I, F, I, P],
    "siss_gemm_nt_geglu_bwd": [P, L, P, P, P, L, I, I, I, P],
    "siss_gemm_nt_geglu_fwd": [P, L, P, P, P, P, I, I, I, P],
    "siss_abi_version": [],
    "siss_conv3x3_sc": [P, L, P, P, L, P, P, L, P, L, P, I, P, I, I, I, IP, IP, I, I, I, P, IP, P],
    "siss_conv3x3_sc_takes": [I, I, I, I, I, I, L, L, L],
    "siss_conv3x3_dgrad_sc": [P, L, P, P, L, P, L, P, P, L, I, I, I, I, IP, IP, I, I, I, P],
    "siss_conv3x3_dgrad_sc_takes": [I, I, I, I, I, I, L, L, L, L],
    "siss_conv_qstats_words": [L, I],
    "siss_gemm_nt_d2s": [P, L, P, P, L, P, L, I, I, I, I, IP, IP, I, I, I, I, P],
    "siss_gemm_nt_d2s_bias": [P, L, P, P, L, P, I, I, I, I, IP, IP, I, I, I, I, P],
    "siss_gemm_nt_d2s_phases": [P, L, P, P, L, P, P, L, I, I, I, IP, IP, IP, I, I, I, P],
    "siss_upsample_phase_weights": [P, P, P, I, I, P],
    "siss_upsample_phase_wgrad_fold": [P, P, L, I, I, I, P],
    "siss_gemm_nt_set_workspace": [P, L],
    "siss_gemm_nt_set_c3p_blocks": [I],
    "siss_dispatch_count": [I],
    "siss_dispatch_reset": [],
    "siss_gemm_nt_mulsub": [P, L, P, P, L, P, L, P, I, I, I, F, I, L, L, L, P],
    "siss_rowdot": [P, P, P, L, L, I, P],
    "siss_gemm_tn": [P, L, P, L, P, L, I, I, I, IP, IP, I, I, L, I, I, I, P, P, P, P],
    "siss_gemm_tn_bs": [P, L, P, L, P, L, I, I, I, IP, IP, I, I, L, I, I, I, P, P, P, L, P],
    "siss_gemm_tn_grouped": [P, I, P],
    "siss_gemm_tn_grouped_capped": [P, I, I, P],
    "siss_gemm_tn_pair": [P, P, I, P],
    "siss_gemm_tn_set_pair_cost": [I],
    "siss_gemm_tn_overwrite_log": [P, L],
    "siss_zero_ranges": [P, P, I, L, P],
    "siss_gn_partial_words": [I, I, I, I, I],
    "siss_groupnorm_set_slab": [I],
    "siss_groupnorm_fwd": [P, P, P, P, P, P, P, I, I, I, I, I, F, I, I, P],
    "siss_groupnorm_fwd_ld": [P, P, P, P, P, P, P, I, I, I, I, I, F, I, I, I, P],
    "siss_quad_stats": [P, L, I, I, I, P, P],
    "siss_groupnorm_fwd_qs": [P, P, P, P, P, P, P, P, I, P, I, I, I, I, I, F, I, I, I, P],
    "siss_groupnorm_bwd": [P, P, P, P, P, P, P, P, P, P, I, I, P, P, P, L, P, I, I, I, L, I, I, I, I, I, I, P],
    "siss_groupnorm_bwd_ld": [P, P, P, P, P, P, P, P, P, P, I, I, P, P, P, L, P, I, I, I, L, I, I, I, I, I, I, I, P],
    "siss_groupnorm_bwd_ld_s2d": [P, P, P, P, P, P, P, P, P, P, I, I, P, P, P, L, P, I, I, I, L, I, I, I, I, I, I, I, P],
    "siss_upsample2x": [P, P, I, I, I, I, P],
    "siss_upsample2x_bwd": [P, P, I, I, I, I, P],
    "siss_concat": [P, P, P, I, I, I, I, I, P],
    "siss_concat_tail": [P, P, I, I, I, I, I, P],
    "siss_concat_bwd": [P, P, P, I, I, I, I, I, I, P],
    "siss_add_inplace": [P, P, I, I, I, I, P],
    "siss_space_to_depth": [P, P, I, I, I, I, P],
    "siss_space_to_depth_ld": [P, P, I, I, I, I, I, P],
    "siss_depth_to_space": [P, P, I, I, I, I, I, P],
    "siss_pad_to_compact": [P, P, I, I, I, I, P],
    "siss_compact_add_to_pad": [P, P, P, I, I, I, I, P],
    "siss_transpose_bf16": [P, P, I, I, I, P],
    "siss_colsum": [P, L, I, I, L, P, P, P],
    "siss_im2col3x3": [P, I, P, I, I, I, I, I, I, P],
    "siss_nchw_channel_sums": [P, I, I, I, L, L, P, P],
    "siss_conv_out_fprop": [P, P, P, P, I, I, I, I, I, P],
    "siss_conv_out_dgrad": [P, P, P, I, I, I, I, I, P],
    "siss_conv_out_wgrad": [P, P, P, P, I, I, I, L, L, I, I, I, I, P],
    "siss_mha_small_fwd": [P, P, P, P, P, I, I, I, I, F, P],
    "siss_mha_small_bwd": [P, P, P, P, P, P, P, P, P, I, I, I, I, I, F, P],
    "siss_softmax_fwd": [P, P, L, I, P],
    "siss_softmax_bwd": [P, P, P, L, L, I, F, P],
    "siss_layernorm_fwd": [P, P, P, P, P, P, L, I, F, P],
    "siss_layernorm_bwd": [P, P, P, P, P, P, P, P, P, L, L, L, L, I, P],
    "siss_geglu_fwd": [P, P, L, I, P],
    "siss_geglu_bwd": [P, P, P, L, L, I, P],
    "siss_head_split": [P, P, I, I, I, I, I, I, P],
    "siss_head_merge": [P, P, I, I, I, I, I, I, P],
    "siss_flash_attn_fwd": [P, P, P, P, P, I, I, I, I, I, F, P],
    "siss_flash_attn_bwd": [P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, F, P],
    "siss_flash_attn_fwd_merged": [P, L, P, L, P, L, P, L, P, I, I, I, I, I, F, I, P],
    "siss_flash_attn_bwd_merged": [P, L, P, L, P, L, P, L, P, L, P, P, P, L, P, L, P, L, I, I, I, I, I, I, F, I, P],
    "siss_attn1h_takes": [I, I],
    "siss_attn1h_fwd": [P, P, P, L, P, L, I, P, I, I, I, F, P],
    "siss_attn1h_bwd": [P, P, P, L, P, L, P, L, I, P, P, P, P, P, L, I, I, I, I, F, P],
    "siss_softmax_rows_fwd": [P, P, L, I, I, I, P],
    "siss_quick_gelu": [P, P, L, P],
    "siss_softmax_rows_bwd": [P, P, P, L, L, I, I, F, P],
    "siss_timestep_sincos": [P, P, I, I, I, F, P],
    "siss_linear_small_fwd": [P, P, P, P, I, I, I, I, P],
    "siss_linear_multi_fwd": [P, P, P, P, P, I, I, I, P],
    "siss_linear_multi_bwd": [P, P, P, P, P, P, P, P, I, I, I, L, I, I, P],
    "siss_linear_small_bwd": [P, P, P, P, P, I, P, P, P, I, I, I, L, L, I, I, I, P],
}
# ---- the f32 parity mode (csrc/f32_path.hip): same argument lists as the bf16 entry points, f32 tensors.  F32_ENTRY maps a bf16
# entry point to its f32 form; F32_SAME lists the ones that never see an activation (f32 / index tensors only) and serve both
# modes.  In f32 mode (f32_mode(True): set by an f32 engine around its launches) call() takes the f32 form and REFUSES an entry
# point that has none -- a bf16 kernel handed f32 bytes would produce numbers, not an error.
F32_ENTRY = {n: n + "_f32" for n in (
    "siss_gemm_nt", "siss_gemm_tn", "siss_groupnorm_fwd_ld", "siss_groupnorm_bwd_ld", "siss_upsample2x", "siss_upsample2x_bwd",
    "siss_concat", "siss_concat_tail", "siss_concat_bwd", "siss_add_inplace", "siss_space_to_depth_ld", "siss_depth_to_space",
    "siss_pad_to_compact", "siss_compact_add_to_pad", "siss_im2col3x3", "siss_conv_out_fprop", "siss_mha_small_fwd",
    "siss_mha_small_bwd", "siss_softmax_fwd", "siss_softmax_bwd", "siss_softmax_rows_fwd", "siss_softmax_rows_bwd",
    "siss_layernorm_fwd", "siss_layernorm_bwd", "siss_geglu_fwd", "siss_geglu_bwd", "siss_head_split", "siss_head_merge",
    "siss_rowdot", "siss_gemm_nt_mulsub")}
F32_ENTRY.update({"siss_transpose_bf16": "siss_transpose_f32", "siss_cast_f32_bf16": "siss_copy_f32",
                  "siss_conv_weight_dgrad_multi_bf16": "siss_conv_weight_dgrad_multi_f32"})
# round 5: f32 forms of the engine's SCHEDULE SWITCHES (folded shortcut, depth-to-space epilogue, sub-pixel upsample, grouped wgrads),
# so that UNetEngine(dtype=float32, f32_fused=True) runs the fused schedule against the fp32 oracle
F32_ENTRY.update({n: n + "_f32" for n in ("siss_gemm_nt_d2s", "siss_gemm_nt_d2s_bias", "siss_gemm_nt_d2s_phases", "siss_conv3x3_sc", "siss_conv3x3_dgrad_sc",
                                          "siss_gemm_tn_bs", "siss_gemm_tn_grouped", "siss_groupnorm_bwd_ld_s2d",
                                          "siss_upsample_phase_weights")})
F32_SAME = {"siss_zero_ranges", "siss_upsample_phase_wgrad_fold", "siss_timestep_sincos", "siss_linear_small_fwd", "siss_linear_small_bwd", "siss_linear_multi_fwd", "siss_linear_multi_bwd",
            "siss_nchw_channel_sums", "siss_mixture_fwd", "siss_mixture_select", "siss_loss_bwd_seed", "siss_mse_bwd_seed",
            "siss_ddpm_step", "siss_grad_norms_scale", "siss_grad_norm_partials", "siss_grad_scalars", "siss_recombine_clip_adamw"}
for _b, _f in F32_ENTRY.items():
    SIGNATURES[_f] = SIGNATURES[_b]
_MODE = threading.local()        # per thread: autograd runs an engine's backward on its own device thread (siss_amd/model.py)


def in_f32_mode():
    return bool(getattr(_MODE, "f32", False))


class f32_mode:
    """`with lib.f32_mode(engine.f32):` -- the launches of THIS thread inside the block go to the f32 entry points when the flag is set."""

    def __init__(self, on):
        self.on = bool(on)

    def __enter__(self):
        self.prev, _MODE.f32 = getattr(_MODE, "f32", False), self.on
        return self

    def __exit__(self, *exc):
        _MODE.f32 = self.prev
        return False


_RET_LONG = {"siss_gemm_tn_overwrite_log", "siss_loss_partials_words", "siss_opt_partials_words", "siss_opt_scalars_words",
             "siss_gn_partial_words", "siss_dispatch_count", "siss_conv_qstats_words"}

# siss_dispatch_count() ids (common.h SissKernelId): which device kernel a launcher call landed on
KERNEL_IDS = {"gemm_nt_kernel": 0, "gemm_nt_c3p_kernel": 1, "flash_attn_fwd": 2, "flash_attn_bwd": 3,
              "gemm_nt_kernel/splitk": 4, "gemm_tn_kernel<1>": 5, "gemm_tn_kernel<3>": 6, "gn_slab": 7, "gn_qstats": 8, "flash_dkdv_qsplit": 9,
              "attn1h_fwd": 10, "attn1h_bwd": 11, "gemm_tn_pair": 12, "flash32_bwd": 13, "flash32_fwd": 14, "gemm_nt_kernel/wide": 15}


def dispatch_counts(reset=False):
    """{kernel symbol: launches since the last reset} from the library's own dispatch counters."""
    lib = load()
    out = {k: int(lib.siss_dispatch_count(i)) for k, i in KERNEL_IDS.items()}
    if reset:
        lib.siss_dispatch_reset()
    return out



class TNJob(C.Structure):
    """siss_tn_job of include/siss_hip.h: one problem of siss_gemm_tn_grouped."""
    _fields_ = [("Y", P), ("ldy", L), ("X", P), ("ldx", L), ("dW", P), ("set_stride", L),
                ("N", I), ("C", I), ("npanels", I), ("nsets", I), ("rows_per_set", I), ("row_begin", I), ("row_end", I),
                ("nsplits", I), ("x_set_rows", L), ("zero_page", P), ("dbias", P), ("dbias2", P),
                ("shifts", I * 9), ("coffs", I * 9), ("bias_set_stride", L)]


_lib = None
MIN_ABI = 5          # oldest build an A/B may load: round 5's final (siss_tn_job with bias_set_stride, attn1h / quad_stats / grouped_capped)


def load():
    """Load the HIP library; raise loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -m siss_amd.build` "
            "(there is no CPU fallback for the SISS hot path)")
    lib = C.CDLL(LIB_PATH)
    override = os.path.abspath(LIB_PATH) != os.path.join(_HERE, "libsiss_hip.so")
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)       # AttributeError if the symbol is missing
        except AttributeError:
            if override:                  # an OLDER build loaded for an A/B (bench.py --lib, tools/ab_bench.sh): newer entry points absent
                continue
            raise
        fn.argtypes = argtypes
        fn.restype = C.c_long if name in _RET_LONG else C.c_int
    _lib = lib
    if override and abi_version() < MIN_ABI:
        _lib = None
        raise RuntimeError(f"{LIB_PATH}: ABI version {abi_version()} < {MIN_ABI} (siss_tn_job layout / entry points the engines call "
                           "unconditionally): rebuild that tree, or A/B against a newer build")
    return lib


def has(name):
    """Whether the loaded library exports `name` (an older build loaded through SISS_LIB_PATH / bench.py --lib may lack newer entry
    points: callers switch the feature off instead of failing in the middle of a step)."""
    return hasattr(load(), name)


def abi_version():
    """siss_abi_version() of the loaded library (struct layouts + the entry points the engines call unconditionally); builds
    before round 6 have no such export and count as 5."""
    lib = load()
    return int(lib.siss_abi_version()) if hasattr(lib, "siss_abi_version") else 5


_WORKSPACE = {}
WORKSPACE_BYTES = 64 << 20      # split-K partial tiles of siss_gemm_nt (at most 32 MiB) / query-chunk partials of the attention dK, dV kernel


def ensure_workspace(device):
    """Hand the library its split-K scratch for `device` (one buffer per device, kept alive here; the C side never
    allocates and keeps one pointer PER DEVICE).  Launches on one stream at a time use it -- the one-compute-stream
    schedule of this package."""
    device = torch.device(device)
    index = device.index if device.index is not None else torch.cuda.current_device()
    key = (device.type, index)
    if key not in _WORKSPACE:
        with torch.cuda.device(index):           # the library files the pointer under hipGetDevice()
            buf = torch.zeros(WORKSPACE_BYTES, dtype=torch.uint8, device=torch.device("cuda", index))    # zero-filled once: arrival counters
            rc = load().siss_gemm_nt_set_workspace(C.c_void_p(buf.data_ptr()), WORKSPACE_BYTES)
        if rc != 0:
            raise RuntimeError(f"siss_gemm_nt_set_workspace failed with status {rc}")
        _WORKSPACE[key] = buf
    return _WORKSPACE[key]


def ptr(t):
    """Device (or host) pointer of a tensor; None -> NULL."""
    if t is None:
        return None
    if torch.is_tensor(t):
        return C.c_void_p(t.data_ptr())
    return t


def int_array(xs):
    return (C.c_int * len(xs))(*xs)



def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# Optional per-launch timing (bench.py): when PROF is a list, every call is bracketed by events on
# the launch stream and (name, start, end, work) is appended.  work = algorithmic flops of the launch
# for the two MFMA GEMM kernels, 0 otherwise.
PROF = None


def _work(name, a):
    """ALGORITHMIC flops of a GEMM launch: the rows that count are the images' TRUE pixels (B * H * W), not the rows of
    the padded-NHWC layout the kernel walks (halo pixels are layout overhead, 3.3 % at 256 x 256, 13 % at 32 x 32)."""
    if name == "siss_gemm_nt":      # 2 * M * N * Kp * npanels * batch
        M, rpi, hp, wp = a[10], a[16], a[17], a[18]
        if hp > 2 and wp > 2 and M % rpi == 0:
            M = (M // rpi) * (hp - 2) * (wp - 2)
        return 2.0 * M * a[11] * a[12] * a[13] * a[20]
    if name == "siss_conv3x3_sc":   # 2 * M * N * (9 Kp + K2): the 3x3 filter and the folded 1x1 shortcut
        M, rpi, hp, wp = a[13], a[18], a[19], a[20]
        if hp > 2 and wp > 2 and M % rpi == 0:
            M = (M // rpi) * (hp - 2) * (wp - 2)
        return 2.0 * M * a[14] * (9 * a[15] + a[11])
    if name == "siss_conv3x3_dgrad_sc":   # 2 * M * Kp * (9 N + Nx): the 3x3 dgrad and the 1x1 shortcut dgrad over the same cotangent
        M, rpi, hp, wp = a[11], a[16], a[17], a[18]
        if hp > 2 and wp > 2 and M % rpi == 0:
            M = (M // rpi) * (hp - 2) * (wp - 2)
        return 2.0 * M * a[13] * (9 * a[12] + a[10])
    if name == "siss_gemm_tn":      # 2 * N * C * npanels * nsets * rows
        rows, rps, rb = a[15] - a[14], a[12], a[14]
        wp = rb - 1                 # padded layouts reduce over rows [wp + 1, rows_per_set - (wp + 1)); images are square
        if wp > 2 and rps % (wp * wp) == 0 and rows == rps - 2 * rb:
            rows = (rps // (wp * wp)) * (wp - 2) * (wp - 2)
        return 2.0 * a[6] * a[7] * a[8] * a[11] * rows
    if name in ("siss_gemm_tn_grouped", "siss_gemm_tn_grouped_capped"):
        return sum(_work("siss_gemm_tn", [None] * 6 + [j.N, j.C, j.npanels, None, None, j.nsets, j.rows_per_set, None,
                                                       j.row_begin, j.row_end]) for j in a[0])
    if name == "siss_gemm_tn_pair":     # (byref(job3), byref(job1), max_blocks): both products
        return sum(_work("siss_gemm_tn", [None] * 6 + [j.N, j.C, j.npanels, None, None, j.nsets, j.rows_per_set, None,
                                                       j.row_begin, j.row_end]) for j in (a[0]._obj, a[1]._obj))
    if name == "siss_attn1h_fwd":       # QK^T and PV: 2 products of 2 B S S D
        return 2.0 * 2 * a[8] * a[9] * a[9] * a[10]
    if name == "siss_attn1h_bwd":       # algorithmic: S, dP, dQ, dK, dV over the nb cotangent images (7 products run: S and dP twice)
        return 2.0 * 5 * a[15] * a[17] * a[17] * a[18]
    if name == "siss_gemm_nt_mulsub":   # 2 * M * N * Kp * batch
        return 2.0 * a[8] * a[9] * a[10] * a[12]
    if name == "siss_flash_attn_fwd":   # QK^T and PV over the VALID keys (padded queries / head dim counted as laid out)
        return 2.0 * 2 * a[5] * a[6] * a[9] * a[8]
    if name == "siss_flash_attn_bwd":   # algorithmic: S, dP, dQ, dK, dV (the two-kernel form recomputes S and dP: 7 products run)
        return 2.0 * 5 * a[9] * a[11] * a[14] * a[13]
    if name == "siss_flash_attn_fwd_merged":   # true head dim, true rows: 2 products over B * H heads
        return 2.0 * 2 * a[9] * a[10] * a[11] * a[12] * a[13]
    if name == "siss_flash_attn_bwd_merged":
        return 2.0 * 5 * a[18] * a[20] * a[21] * a[22] * a[23]
    return 0.0


def hbm_bytes(name, a):
    """ALGORITHMIC HBM bytes of one launch of the HBM-bound launchers (every operand read once, every
    result written once), for bench.py's GB/s-vs-HBM-peak figures.  None for the others."""
    if name in ("siss_groupnorm_fwd", "siss_groupnorm_fwd_ld"):            # read x + write y (bf16)
        return 2.0 * 2 * a[7] * a[8] * a[9] * a[10]
    if name in ("siss_groupnorm_bwd", "siss_groupnorm_bwd_ld", "siss_groupnorm_bwd_ld_s2d"):   # read x (nx samples), read dy + write dx (n2 samples) (+ accum reads)
        px = a[21] * a[22] * a[23]
        n2, nx = a[17], a[18]
        extra = (1 if a[7] is not None else 0) + (1 if a[8] is not None else 0)
        return 2.0 * px * (nx + (2 + extra) * n2)
    if name == "siss_recombine_clip_adamw":     # read g_x, g_a, theta, m, v; write theta, m, v (f32)
        return 32.0 * a[7]
    if name == "siss_mixture_fwd":              # read x0, a0, noise; write x_mix
        return 4.0 * (2 if a[3] else 4) * a[10] * a[11]
    if name == "siss_loss_bwd_seed":            # read pred (f32), x_mix, x0, a0; write c_x, c_a (f32)
        return (4 + 3 * (2 if a[4] else 4) + 8.0) * a[10] * a[11]
    return None


def _shape_key(name, a):
    """Problem shape of a launch, for per-layer breakdowns (tools/step_breakdown.py)."""
    if name == "siss_gemm_nt":
        return ("M", a[10], "N", a[11], "K", a[12], "panels", a[13], "batch", a[20]) + ((a[24],) if len(a) > 24 else ())
    if name == "siss_conv3x3_sc":
        return ("M", a[13], "N", a[14], "K", a[15], "panels", 9, "+1x1 K", a[11])
    if name == "siss_conv3x3_dgrad_sc":
        return ("M", a[11], "N", a[12], "K", a[13], "panels", 9, "+1x1 N", a[10])
    if name == "siss_gemm_tn":
        return ("N", a[6], "C", a[7], "panels", a[8], "sets", a[11], "rows", a[15] - a[14], "splits", a[16])
    if name == "siss_gemm_tn_pair":
        j3, j1 = a[0]._obj, a[1]._obj
        return ("N", j3.N, "C", j3.C, "panels", j3.npanels, "+ N", j1.N, "C", j1.C, "panels", j1.npanels, "rows", j3.row_end - j3.row_begin)
    if name in ("siss_attn1h_fwd", "siss_attn1h_bwd"):
        return ("B", a[8] if name.endswith("fwd") else a[15], "S", a[9] if name.endswith("fwd") else a[17], "D", a[10] if name.endswith("fwd") else a[18])
    if name in ("siss_groupnorm_fwd", "siss_groupnorm_fwd_ld"):
        return ("n", a[7], "H", a[8], "C", a[10])
    if name in ("siss_groupnorm_bwd", "siss_groupnorm_bwd_ld", "siss_groupnorm_bwd_ld_s2d"):
        return ("n2", a[17], "H", a[21], "C", a[23])
    if name == "siss_flash_attn_fwd":
        return ("BH", a[5], "Sq", a[6], "Sk", a[9], "D", a[8])
    if name == "siss_flash_attn_bwd":
        return ("BH", a[9], "Sq", a[11], "Sk", a[14], "D", a[13])
    if name == "siss_flash_attn_fwd_merged":
        return ("BH", a[9] * a[10], "Sq", a[11], "Sk", a[12], "D", a[13])
    if name == "siss_flash_attn_bwd_merged":
        return ("BH", a[18] * a[20], "Sq", a[21], "Sk", a[22], "D", a[23])
    return ()


def kernel_symbol(name, a):
    """Which device kernel a GEMM launcher call lands on (mirrors the dispatch in gemm_nt.hip / gemm_tn.hip), so
    that bench.py can report the roofline of the dominant KERNEL under the name rocprofv3 lists it by."""
    def triples(shifts, coffs, n):
        return n % 3 == 0 and all(shifts[3 * g + 1] == shifts[3 * g] + 1 and shifts[3 * g + 2] == shifts[3 * g] + 2
                                  and coffs[3 * g] == coffs[3 * g + 1] == coffs[3 * g + 2] for g in range(n // 3))
    if name == "siss_gemm_nt":
        M, N, Kp, npan, batch, rpi = a[10], a[11], a[12], a[13], a[20], a[16]
        tiles = -(-M // 128) * -(-N // 128)
        if (npan == 9 and batch == 1 and len(a) <= 24 and Kp % 64 == 0 and N % 128 == 0 and rpi >= 256
                and tiles >= 256
                and triples(a[14], a[15], 9)):
            return "gemm_nt_c3p_kernel"
        return "gemm_nt_kernel"
    if name in ("siss_conv3x3_sc", "siss_conv3x3_dgrad_sc"):
        return "gemm_nt_c3p_kernel"
    if name == "siss_gemm_nt_mulsub":
        return "gemm_nt_kernel"
    if name == "siss_gemm_tn_grouped":
        return "gemm_tn_grouped_kernel"
    if name == "siss_gemm_tn_grouped_capped":
        return "gemm_tn_grouped_capped_kernel"
    if name == "siss_gemm_tn_pair":
        return "gemm_tn_mixed_kernel"
    if name == "siss_gemm_tn":
        rows = a[15] - a[14]
        return "gemm_tn_kernel<3>" if triples(a[9], a[10], a[8]) and (a[16] > 0 or rows >= 8192) else "gemm_tn_kernel<1>"
    return name


def call(name, *args, refusable=False):
    """Call a launcher on torch's current stream; tensors are passed as raw pointers.  refusable: status 1 (a shape the launcher
    does not take) is RETURNED instead of raised -- for entry points documented to refuse shapes the caller then runs another way."""
    lib = load()
    if getattr(_MODE, "f32", False):
        if name in F32_ENTRY:
            name = F32_ENTRY[name]
        elif name not in F32_SAME:
            raise RuntimeError(f"{name} has no f32 form: the f32 parity mode runs the engines with the fused bf16 kernels "
                               "switched off (csrc/f32_path.hip)")
    fn = getattr(lib, name)
    conv = [ptr(a) if (torch.is_tensor(a) or a is None) else
            (C.cast(a, C.c_void_p) if isinstance(a, C.Array) and a._type_ is not C.c_int else a) for a in args]   # job tables
    if PROF is not None:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        rc = fn(*conv, stream_ptr())
        e.record()
        # variants that only add operands are booked under their plain form (same work, same shape key)
        if name == "siss_gemm_nt_qstats":
            name, args = "siss_gemm_nt", list(args[:20]) + [1, 0, 0, 0]
        elif name == "siss_gemm_nt_alpha_cols":             # (A, lda, W, C, ldc, bias, R, ldr, M, N, Kp, alpha, alpha_cols)
            a = args
            name, args = "siss_gemm_nt", [a[0], a[1], a[2], a[3], a[4], a[5], None, a[9], a[6], a[7], a[8], a[9], a[10], 1,
                                          int_array([0]), int_array([0]), 1, 0, 0, a[11], 1, 0, 0, 0]
        elif name == "siss_gemm_nt_geglu_bwd":              # (A, lda, W, dh, h, rows_x, M, N, Kp)
            a = args
            name, args = "siss_gemm_nt", [a[0], a[1], a[2], a[3], 2 * a[7], None, None, a[7], None, 0, a[6], a[7], a[8], 1,
                                          int_array([0]), int_array([0]), 1, 0, 0, 1.0, 1, 0, 0, 0]
        elif name == "siss_gemm_nt_geglu_fwd":              # (A, lda, W, bias, h, y, M, F, Kp)
            a = args
            name, args = "siss_gemm_nt", [a[0], a[1], a[2], a[4], 2 * a[7], a[3], None, 0, None, 0, a[6], 2 * a[7], a[8], 1,
                                          int_array([0]), int_array([0]), 1, 0, 0, 1.0, 1, 0, 0, 0]
        elif name == "siss_gemm_tn_bs":
            name, args = "siss_gemm_tn", list(args[:20])
        elif name == "siss_gemm_nt_d2s_bias":               # (A, lda, W, C, ldc, bias, M, N, Kp, npanels, shifts, coffs, rpi, Hp, Wp, plane)
            a = args
            name, args = "siss_gemm_nt", [a[0], a[1], a[2], a[3], a[4], a[5], None, a[7], None, 0, a[6], a[7], a[8], a[9],
                                          a[10], a[11], a[12], a[13], a[14], 1.0, 1, 0, 0, 0]
        elif name == "siss_gemm_nt_d2s_phases":             # (A, lda, W, C, ldc, bias, R, ldr, M, N, Kp, phase_p0, shifts, coffs, rpi, Hp, Wp)
            a = args
            name, args = "siss_gemm_nt", [a[0], a[1], a[2], a[3], a[4], a[5], None, a[9], a[6], a[7], a[8], a[9], a[10], a[11][4],
                                          a[12], a[13], a[14], a[15], a[16], 1.0, 1, 0, 0, 0, "4 planes"]
        elif name == "siss_gemm_nt_d2s":
            a = args
            name, args = "siss_gemm_nt", [a[0], a[1], a[2], a[3], a[4], None, None, a[8], a[5], a[6], a[7], a[8], a[9], a[10],
                                          a[11], a[12], a[13], a[14], a[15], 1.0, 1, 0, 0, 0]
        elif name == "siss_groupnorm_fwd_qs":
            name, args = "siss_groupnorm_fwd_ld", list(args[:7]) + list(args[10:])
        if name == "siss_groupnorm_bwd_ld_s2d":
            name = "siss_groupnorm_bwd_ld"
        base = name[:-3] if name.endswith("_ld") else name          # row-stride variants count as their plain form
        if base.endswith("_merged"):
            base = base[:-7]
        if base in ("siss_conv3x3_sc", "siss_conv3x3_dgrad_sc"):
            base = "siss_gemm_nt"
        PROF.append((base, s, e, _work(name, args), _shape_key(name, args), kernel_symbol(name, args), hbm_bytes(name, args)))
    else:
        rc = fn(*conv, stream_ptr())
    if rc == 1 and refusable:
        if PROF is not None:
            PROF.pop()                      # nothing ran
        return 1
    if rc != 0:
        raise RuntimeError(f"{name} failed with status {rc} "
                           f"({'bad argument' if rc == 1 else 'launch error'})")
    return 0


def query(name, *args):
    return getattr(load(), name)(*args)


def overwrite_log(max_records=8192):
    """Drain siss_gemm_tn_overwrite_log: [(address, floats), ...] of the weight-gradient products that overwrote their output since
    the last drain (host-side bookkeeping, no device work), or None when the log is unusable (it had filled up, or held more)."""
    buf = (C.c_long * (2 * max_records))()
    n = load().siss_gemm_tn_overwrite_log(C.cast(buf, C.c_void_p), max_records)
    if n < 0 or n > max_records:
        return None
    return [(buf[2 * i], buf[2 * i + 1]) for i in range(n)]
