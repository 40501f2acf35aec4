"""Thin Python launch wrappers over the C-ABI (raw device pointers on torch's current stream).

Native parameter layouts (what the flat master buffer holds):
  conv3x3 weight  [9][Cout][Cin]   (tap = ky*3+kx; the reference/diffusers layout is [Cout][Cin][3][3])
  conv1x1 / linear weight [Cout][Cin]
"""
import torch

from . import lib
from .layout import Act, conv3x3_panels

_ZERO = {}


def zero_page(device):
    z = _ZERO.get(device)
    if z is None:
        z = torch.zeros(4096, dtype=torch.bfloat16, device=device)
        _ZERO[device] = z
    return z


# ---------------------------------------------------------------- weight layout helpers
def conv_w_to_native(w):
    """[Co, Ci, kh, kw] -> [kh*kw, Co, Ci]"""
    co, ci, kh, kw = w.shape
    return w.permute(2, 3, 0, 1).reshape(kh * kw, co, ci).contiguous()


def conv_w_from_native(wn, kh=3, kw=3):
    t, co, ci = wn.shape
    return wn.reshape(kh, kw, co, ci).permute(2, 3, 0, 1).contiguous()


def dgrad_weight(w_native_f32):
    """[T][Co][Ci] f32 -> [T][Ci][Co] bf16 with the tap order reversed (device kernel)."""
    t, co, ci = w_native_f32.shape
    out = torch.empty(t, ci, co, dtype=torch.bfloat16, device=w_native_f32.device)
    lib.call("siss_conv_weight_dgrad_layout", w_native_f32, out, t, co, ci)
    return out


# ---------------------------------------------------------------- panelled NT GEMM
def gemm_nt(a_ptr, lda, w, c_ptr, ldc, M, N, Kp, shifts, coffs, *, bias=None, rowbias=None, ldrb=None,
            res_ptr=None, ldr=0, rows_per_image=1, hp=0, wp=0, alpha=1.0, batch=1,
            stride_a=0, stride_w=0, stride_c=0, qstats=None, alpha_cols=0):
    """qstats (f32 tensor of lib siss_conv_qstats_words(M, N) floats): ask the product to leave the GroupNorm statistics
    of its output there; returns True when it did (the launch went to the persistent 3x3 kernel), else False/None."""
    if alpha_cols:
        # alpha scales only the first alpha_cols output columns (one-panel products: the pre-scaled query part of a fused q / k / v
        # projection, siss_flash_attn_*_merged(q_prescaled = 1))
        assert len(shifts) == 1 and shifts[0] == 0 and coffs[0] == 0 and batch == 1 and rowbias is None and qstats is None
        lib.call("siss_gemm_nt_alpha_cols", a_ptr, lda, w, c_ptr, ldc, bias, res_ptr, ldr, M, N, Kp, float(alpha), int(alpha_cols))
        return
    if qstats is not None:
        assert batch == 1
        written = lib.C.c_int(0)
        lib.call("siss_gemm_nt_qstats", a_ptr, lda, w, c_ptr, ldc, bias, rowbias, ldrb if ldrb is not None else N, res_ptr,
                 ldr, M, N, Kp, len(shifts), lib.int_array(shifts), lib.int_array(coffs), rows_per_image, hp, wp,
                 float(alpha), qstats, lib.C.byref(written))
        return bool(written.value)
    lib.call("siss_gemm_nt", a_ptr, lda, w, c_ptr, ldc, bias, rowbias, ldrb if ldrb is not None else N, res_ptr, ldr, M, N, Kp,
             len(shifts), lib.int_array(shifts), lib.int_array(coffs), rows_per_image, hp, wp,
             float(alpha), batch, stride_a, stride_w, stride_c)


def conv_fprop(x: Act, w_bf16, out: Act, bias=None, rowbias=None, residual: Act = None, ksize=3, ldrb=None):
    """out = conv(x) (+bias +rowbias[img] +residual); w_bf16 [T][Co][Ci] bf16."""
    t, co, ci = w_bf16.shape
    assert ci == x.c and co == out.c and (x.n, x.h, x.w) == (out.n, out.h, out.w)
    if ksize == 3:
        shifts, coffs = conv3x3_panels(x.wp, ci)
    else:
        shifts, coffs = [0], [0]
    assert t == len(shifts)
    gemm_nt(lib.ptr(x.data), getattr(x, "ld", x.c), w_bf16, lib.ptr(out.data), getattr(out, "ld", out.c), x.rows, co, ci,
            shifts, coffs,
            bias=bias, rowbias=rowbias, ldrb=ldrb, res_ptr=lib.ptr(residual.data) if residual is not None else None,
            ldr=getattr(residual, "ld", residual.c) if residual is not None else 0,
            rows_per_image=x.rows_per_image, hp=x.hp, wp=x.wp)
    return out


def conv_fprop_qstats(x: Act, w_bf16, out: Act, qstats, **kw):
    """conv_fprop that asks for the GroupNorm statistics of `out` (see gemm_nt); returns whether they were written."""
    t, co, ci = w_bf16.shape
    assert t == 9 and ci == x.c and co == out.c and (x.n, x.h, x.w) == (out.n, out.h, out.w)
    shifts, coffs = conv3x3_panels(x.wp, ci)
    residual = kw.get("residual")
    return gemm_nt(lib.ptr(x.data), getattr(x, "ld", x.c), w_bf16, lib.ptr(out.data), getattr(out, "ld", out.c), x.rows, co, ci,
                   shifts, coffs, bias=kw.get("bias"), rowbias=kw.get("rowbias"), ldrb=kw.get("ldrb"),
                   res_ptr=lib.ptr(residual.data) if residual is not None else None,
                   ldr=getattr(residual, "ld", residual.c) if residual is not None else 0,
                   rows_per_image=x.rows_per_image, hp=x.hp, wp=x.wp, qstats=qstats)


def conv3x3_sc_takes(x: Act, co, out, x2: Act):
    """Whether conv_fprop_sc can fold the 1x1 shortcut over `x2` into the 3x3 product over `x` (the persistent kernel takes it)."""
    if lib.in_f32_mode():                       # the f32 form (two launches, one accumulation) has no shape limits
        return True
    return bool(lib.query("siss_conv3x3_sc_takes", x.rows, co, x.c, x2.c, x.rows_per_image, x.wp, getattr(x, "ld", x.c),
                          getattr(out, "ld", out.c), getattr(x2, "ld", x2.c)))


def conv_fprop_sc(x: Act, w_bf16, out: Act, x2: Act, w2_bf16, bias=None, bias2=None, rowbias=None, ldrb=None, qstats=None):
    """out = conv3x3(x; w) + conv1x1(x2; w2) + bias + bias2 (+ rowbias[img]) in ONE product (siss_conv3x3_sc): a resnet's
    conv2 + conv_shortcut.  Returns whether the GroupNorm statistics were written to `qstats`."""
    t, co, ci = w_bf16.shape
    assert t == 9 and ci == x.c and co == out.c and (x.n, x.h, x.w) == (out.n, out.h, out.w) == (x2.n, x2.h, x2.w)
    assert tuple(w2_bf16.shape[-2:]) == (co, x2.c)
    shifts, coffs = conv3x3_panels(x.wp, ci)
    written = lib.C.c_int(0)
    lib.call("siss_conv3x3_sc", x.data, getattr(x, "ld", x.c), w_bf16, out.data, getattr(out, "ld", out.c), bias, rowbias,
             ldrb if ldrb is not None else co, x2.data, getattr(x2, "ld", x2.c), w2_bf16, x2.c, bias2, x.rows, co, ci,
             lib.int_array(shifts), lib.int_array(coffs), x.rows_per_image, x.hp, x.wp, qstats,
             lib.C.byref(written) if qstats is not None else None)
    return bool(written.value)


def conv_dgrad(dy: Act, wT_bf16, out: Act, residual: Act = None, ksize=3):
    """out = conv_transpose(dy) for a stride-1 'same' conv; wT_bf16 from dgrad_weight()."""
    t, ci, co = wT_bf16.shape
    assert co == dy.c and ci == out.c
    if ksize == 3:
        # dX[q] = sum_tap dY[q - shift(tap)] W[tap]^T ; tap order already reversed in wT
        shifts, coffs = conv3x3_panels(dy.wp, co)
    else:
        shifts, coffs = [0], [0]
    gemm_nt(lib.ptr(dy.data), dy.c, wT_bf16, lib.ptr(out.data), out.c, dy.rows, ci, co, shifts, coffs,
            res_ptr=lib.ptr(residual.data) if residual is not None else None,
            ldr=residual.c if residual is not None else 0,
            rows_per_image=dy.rows_per_image, hp=dy.hp, wp=dy.wp)
    return out


def conv3x3_dgrad_sc_takes(dy: Act, ci, out: Act, out_x, residual: Act = None):
    """Whether conv_dgrad_sc can run the 1x1 shortcut's dgrad (-> out_x) inside the 3x3 dgrad (-> out) over the same cotangent
    (residual: what conv_dgrad_sc will be given -- its size is part of the kernel's 32-bit addressing limit)."""
    if lib.in_f32_mode():
        return True
    return bool(lib.query("siss_conv3x3_dgrad_sc_takes", dy.rows, ci, dy.c, out_x.c, dy.rows_per_image, dy.wp, dy.c,
                          getattr(out, "ld", out.c), getattr(out_x, "ld", out_x.c), residual.c if residual is not None else 0))


def conv_dgrad_sc(dy: Act, wT_bf16, out: Act, wT_sc_bf16, out_x: Act, residual: Act = None):
    """out = conv3x3^T(dy) (+ residual)  AND  out_x = conv1x1^T(dy)  in ONE product (siss_conv3x3_dgrad_sc): the backward of a
    resnet's conv2 + conv_shortcut tail; wT_sc_bf16 [Cin_x][Cout] = the transposed shortcut weight."""
    t, ci, co = wT_bf16.shape
    assert t == 9 and co == dy.c and ci == out.c and tuple(wT_sc_bf16.shape[-2:]) == (out_x.c, co)
    shifts, coffs = conv3x3_panels(dy.wp, co)
    lib.call("siss_conv3x3_dgrad_sc", dy.data, dy.c, wT_bf16, out.data, getattr(out, "ld", out.c),
             residual.data if residual is not None else None, residual.c if residual is not None else 0, wT_sc_bf16, out_x.data,
             getattr(out_x, "ld", out_x.c), out_x.c, dy.rows, ci, co, lib.int_array(shifts), lib.int_array(coffs),
             dy.rows_per_image, dy.hp, dy.wp)
    return out, out_x


def _nsplits(tiles, npanels, nsets, rows, fused3):
    """Split-K factor of the wgrad GEMM: 0 = let siss_gemm_tn choose (kernel variant + split count from its cost
    model: K-steps per block vs the float-atomic traffic every extra split adds; gemm_tn.hip)."""
    return 0


def is_conv3_panels(shifts, coffs):
    return len(shifts) % 3 == 0 and all(
        shifts[3 * g + 1] == shifts[3 * g] + 1 and shifts[3 * g + 2] == shifts[3 * g] + 2
        and coffs[3 * g] == coffs[3 * g + 1] == coffs[3 * g + 2] for g in range(len(shifts) // 3))


def conv_wgrad(dy: Act, x: Act, dW, nsets, ksize=3, dbias=None):
    """dW[set][T][Co][Ci] += dy^T x over each set's images (dy has nsets*B images; x has B or nsets*B)."""
    co, ci = dy.c, x.c
    assert dy.n % nsets == 0
    b = dy.n // nsets
    assert x.n in (b, dy.n) and (x.h, x.w) == (dy.h, dy.w)
    if ksize == 3:
        shifts, coffs = conv3x3_panels(dy.wp, ci)
    else:
        shifts, coffs = [0], [0]
    t = len(shifts)
    assert dW.dtype == torch.float32 and dW.numel() == nsets * t * co * ci
    rows_per_set = b * dy.rows_per_image
    x_set_rows = 0 if x.n == b else rows_per_set
    rb, re = dy.wp + 1, rows_per_set - (dy.wp + 1)
    tiles = (-(-co // 128)) * (-(-ci // 128))
    ns = _nsplits(tiles, t, nsets, re - rb, is_conv3_panels(shifts, coffs))
    lib.call("siss_gemm_tn", dy.data, dy.c, x.data, x.c, dW, t * co * ci, co, ci, t,
             lib.int_array(shifts), lib.int_array(coffs), nsets, rows_per_set, x_set_rows, rb, re, ns,
             zero_page(dy.buf.device), dbias, None)
    return dW
