"""Trunk convolutions of a TRAINING step on the hand-written Winograd / fp32-MFMA kernel.

At the Othello configuration an iteration's turnaround is 91 % training, and a training step on the stock
library kernels spends most of its time in the 3x3 convolutions of the residual trunk: the library's F(2x2,3x3) kernel needs
~140 us for the forward or the backward-data convolution of a 1024-board batch, this repository's F(4x4,3x3) kernel ~15 us.
`WinoConv3x3` is a `torch.autograd.Function` with the semantics of `conv2d(x, weight, bias, padding=1)` for 64 -> 64 channels on
boards up to 8x8:

  forward        y  = conv(x, w) + b            sprl_wino_conv64 (cnn_wino.hip) on layout W, scale = 1, shift = b, no ReLU
  backward-data  dx = conv(dy, w^T rot180)      the same kernel with the filters transposed over the channels and rotated
  backward-weights, backward-bias               the framework's own (torch.nn.grad.conv2d_weight, a sum)

The filters change with every optimiser step, so they are brought to the Winograd domain on the device per call
(cnn_train.hip: sprl_train_wino_weights, the arithmetic of torch_eval.cpp: wino_transform); tensors cross NCHW <-> layout W
through two small kernels.  Every launch goes to torch's current stream, so the op can be captured in a HIP graph
(trainer._GraphStep).  `fast_trunk(net)` routes the residual blocks' convolutions of a GridResNet through it for the duration of
a `with` block; the module, its parameters and its state_dict are untouched (the exported TorchScript file is the plain network).
"""
import contextlib
import ctypes as C
import os

import torch

_LIB = [None]
_CONST = {}


def _lib():
    if _LIB[0] is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsprl_amd_torch.so")
        L = C.CDLL(path)                      # raises when the gfx950 build is missing: no fallback inside the op
        L.sprl_train_nchw_to_w.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.sprl_train_w_to_nchw.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.sprl_train_wino_weights.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.sprl_wino_conv64.argtypes = [C.c_void_p] * 6 + [C.c_int] * 4 + [C.c_void_p]
        _LIB[0] = L
    return _LIB[0]


def available():
    """The op needs the gfx950 plugin library and a GPU."""
    try:
        return torch.cuda.is_available() and _lib() is not None
    except OSError:
        return False


def _consts(device):
    key = str(device)
    if key not in _CONST:
        _CONST[key] = (torch.ones(64, device=device), torch.zeros(64, device=device))
    return _CONST[key]


def _conv_w(x, weight, shift, flip_transpose):
    """conv3x3(x, filters) + shift[channel] through layout W; filters = weight (forward) or its transposed, rotated form."""
    L = _lib()
    B, Cc, H, W = x.shape
    st = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
    x = x.contiguous()
    xw = torch.empty(B, 4096, device=x.device, dtype=torch.float32)
    yw = torch.empty(B, 4096, device=x.device, dtype=torch.float32)
    u = torch.empty(36 * 64 * 64, device=x.device, dtype=torch.float32)
    y = torch.empty(B, 64, H, W, device=x.device, dtype=torch.float32)
    ones, _ = _consts(x.device)
    rc = L.sprl_train_nchw_to_w(x.data_ptr(), xw.data_ptr(), B, H, W, st)
    rc |= L.sprl_train_wino_weights(weight.data_ptr(), u.data_ptr(), 1 if flip_transpose else 0, st)
    rc |= L.sprl_wino_conv64(xw.data_ptr(), u.data_ptr(), ones.data_ptr(), shift.data_ptr(), None, yw.data_ptr(), B, H, W, 0, st)
    rc |= L.sprl_train_w_to_nchw(yw.data_ptr(), y.data_ptr(), B, H, W, st)
    if rc != 0:
        raise RuntimeError(f"hand-written convolution failed (rc {rc}): batch {B}, board {H}x{W}")
    return y


def supported(x, weight):
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 64 and
            tuple(weight.shape) == (64, 64, 3, 3) and (x.shape[2], x.shape[3]) in ((8, 8), (6, 7), (7, 7)) and
            x.shape[0] * 16384 < 0xFFFFFFFF)


class WinoConv3x3(torch.autograd.Function):
    """y = conv2d(x, weight, bias, stride 1, padding 1) for 64 -> 64 channels, boards 8x8 / 6x7 / 7x7, fp32."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return _conv_w(x, weight.contiguous(), bias.contiguous(), False)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            _, zeros = _consts(dy.device)
            dx = _conv_w(dy, weight.contiguous(), zeros, True)
        if ctx.needs_input_grad[1]:
            dw = torch.nn.grad.conv2d_weight(x, weight.shape, dy, stride=1, padding=1)
        if ctx.needs_input_grad[2]:
            db = dy.sum((0, 2, 3))
        return dx, dw, db


@contextlib.contextmanager
def fast_trunk(net):
    """Inside the block, the 3x3 64 -> 64 convolutions of `net`'s residual blocks run through WinoConv3x3 whenever the input is
    supported (CUDA fp32, board up to 8x8); anything else takes the module's own forward."""
    patched = []
    for block in getattr(net, "residual_blocks", []):
        for conv in (block.conv1, block.conv2):
            if conv.in_channels == 64 and conv.out_channels == 64 and conv.kernel_size == (3, 3) and conv.padding == (1, 1) and \
                    conv.stride == (1, 1) and conv.bias is not None:
                def fwd(x, conv=conv, orig=conv.forward):
                    if supported(x, conv.weight):
                        return WinoConv3x3.apply(x, conv.weight, conv.bias)
                    return orig(x)
                patched.append(conv)
                conv.forward = fwd
    try:
        yield len(patched)
    finally:
        for conv in patched:
            del conv.forward                      # the instance attribute shadows the class method: remove it again
