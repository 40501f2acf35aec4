"""Activation layout in HBM: NHWC bf16 with a one-pixel ZERO halo, flattened to rows.

A tensor of logical shape [N, C, H, W] is stored as rows r = (n, y, x), y in [0, H+2),
x in [0, W+2), of C contiguous bf16 channels.  Interior pixel (y, x) of the image lives at
padded (y+1, x+1).  Invariants every kernel keeps:

* halo rows are zero (3x3 zero padding comes for free; wgrad can reduce over the flat range);
* `guard` zero rows precede and follow the tensor, so row-shifted panel reads
  (shift in [-(W+3), W+3]) of the first/last tiles stay inside the allocation and read zeros.
"""
import torch


class Act:
    """A padded NHWC bf16 activation living in one flat torch buffer."""

    __slots__ = ("buf", "n", "h", "w", "c", "guard", "cat_parts", "cat_done", "qstats", "s2d_cot")

    def __init__(self, n, h, w, c, device="cuda", buf=None, dtype=torch.bfloat16):
        """dtype: bf16 (the product path) or f32 (the f32 parity mode: csrc/f32_path.hip)."""
        self.n, self.h, self.w, self.c = n, h, w, c
        self.cat_parts, self.cat_done = None, False
        self.qstats = None                            # GroupNorm statistics left by the producing conv (engine bookkeeping)
        self.s2d_cot = False                          # its consumer wants its cotangent in space-to-depth layout (sub-pixel upsample)
        self.guard = (w + 2) + 2                      # rows of zero guard on each side
        rows = self.rows + 2 * self.guard
        if buf is None:
            buf = torch.zeros(rows * c, dtype=dtype, device=device)
        assert buf.numel() == rows * c
        self.buf = buf

    @property
    def hp(self):
        return self.h + 2

    @property
    def wp(self):
        return self.w + 2

    @property
    def rows_per_image(self):
        return self.hp * self.wp

    @property
    def rows(self):
        return self.n * self.rows_per_image

    @property
    def data(self):
        """Flat [rows, C] view starting at row 0 of the tensor (after the guard)."""
        g = self.guard * self.c
        return self.buf[g:g + self.rows * self.c].view(self.rows, self.c)

    def data_ptr(self):
        return self.data.data_ptr()

    def padded(self):
        return self.data.view(self.n, self.hp, self.wp, self.c)

    def interior(self):
        return self.padded()[:, 1:-1, 1:-1, :]

    def to_nchw(self):
        return self.interior().permute(0, 3, 1, 2).float().contiguous()

    def set_from_nchw(self, x):
        assert tuple(x.shape) == (self.n, self.c, self.h, self.w), (x.shape, (self.n, self.c, self.h, self.w))
        self.interior().copy_(x.permute(0, 2, 3, 1).to(self.buf.dtype))
        return self

    @staticmethod
    def from_nchw(x, device="cuda", dtype=torch.bfloat16):
        n, c, h, w = x.shape
        return Act(n, h, w, c, device=device, dtype=dtype).set_from_nchw(x.to(device))

    def like(self, c=None, n=None):
        return Act(n or self.n, self.h, self.w, c or self.c, device=self.buf.device, dtype=self.buf.dtype)

    def halo_is_zero(self):
        p = self.padded().float()
        return bool((p[:, 0].abs().sum() + p[:, -1].abs().sum() + p[:, :, 0].abs().sum()
                     + p[:, :, -1].abs().sum()) == 0)


class ActView:
    """Columns [c0, c0 + c) of a wider padded activation `base` (row stride ld = base.c).  A conv whose result is about
    to be concatenated writes straight into the concat buffer through one of these (GEMM epilogue with ldc = ld), so
    the concat only has to copy the other part."""

    __slots__ = ("base", "c0", "n", "h", "w", "c", "ld", "cat_parts", "cat_done", "qstats", "s2d_cot")

    def __init__(self, base, c0, c):
        assert 0 <= c0 and c0 + c <= base.c and c0 % 8 == 0 and c % 8 == 0
        self.base, self.c0, self.c, self.ld = base, c0, c, base.c
        self.qstats = None
        self.s2d_cot = False
        self.n, self.h, self.w = base.n, base.h, base.w
        self.cat_parts, self.cat_done = None, False

    hp = property(lambda self: self.h + 2)
    wp = property(lambda self: self.w + 2)
    rows_per_image = property(lambda self: self.hp * self.wp)
    rows = property(lambda self: self.n * self.rows_per_image)

    @property
    def data(self):
        return self.base.data[:, self.c0:self.c0 + self.c]


def conv3x3_panels(wp, cin):
    """(shifts, coffs) of the nine 3x3 taps in flat padded row space; tap = ky*3 + kx."""
    shifts = [(ky - 1) * wp + (kx - 1) for ky in range(3) for kx in range(3)]
    return shifts, [0] * 9
