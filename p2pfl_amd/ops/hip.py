"""ctypes bindings of ``libp2pfl_hip.so`` (the gfx950 kernels under ``p2pfl_amd/csrc``).

Every wrapper takes torch tensors whose *leading* dimension is the client dimension; the
client stride is ``t.stride(0)``, so views into the ``[clients, P]`` slabs
(``slab[:, off:off+n].unflatten(1, shape)``) are passed without copies. Kernels are enqueued
on the current torch HIP stream, so they compose with torch ops and can be captured into a
hipGraph with ``torch.cuda.graph``.

The library is mandatory on a GPU box: if it is missing, calls raise instead of silently
falling back to eager PyTorch.
"""
from __future__ import annotations

import ctypes
import threading
from ctypes import c_float, c_int, c_long, c_void_p
from typing import Optional

import torch

from .build import HIP_LIB

_lib = None


class HipKernelError(RuntimeError):
    pass


def load():
    """Load (once) and return the kernel library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not HIP_LIB.exists():
            raise HipKernelError(
                f"{HIP_LIB} is missing: run `python -m p2pfl_amd.ops.build` (or __graft_entry__.build()) first")
        _lib = ctypes.CDLL(str(HIP_LIB))
        for name in dir(_Sig):
            if name.startswith("p2pfl_"):
                fn = getattr(_lib, name)
                fn.argtypes = getattr(_Sig, name)
                fn.restype = c_long if name.endswith("_scratch") else c_int
    return _lib


def available() -> bool:
    return HIP_LIB.exists() and torch.cuda.is_available()


P, L, I, F = c_void_p, c_long, c_int, c_float

# Precision policy of the matrix products (see csrc/common.h). The reference trains 32-bit
# (fedstellar/learning/pytorch/lightninglearner.py:182-193), so the default is "fp32": fp32 storage, fp32 accumulation,
# fp32 operands on v_mfma_f32_32x32x2_f32 (exact fp32 fma chains) everywhere except the stage-2 convolutions (the
# matrix-core-bound kernels of a step), which form every fp32 product from three bf16 terms per operand - an EXACT
# decomposition (3 x 8 significand bits), six bf16 MFMAs with fp32 accumulation, dropped cross terms <= 2^-24 of the
# product: the same accuracy against a float64 oracle as the fp32 MFMA kernels at 2.7x fewer matrix-core cycles.
# "fp32-mfma" runs those convolutions on v_mfma_f32_32x32x2_f32 too (nothing but fp32 instructions anywhere);
# "bf16" rounds all operands to bf16 with fp32 accumulation (16x the matrix-core rate; a labelled fast mode).
# Every product wrapper takes ``prec`` (name or None = default).
PRECISIONS = {"bf16": 0, "fp32-mfma": 1, "fp32": 2}
SPLIT_TERMS = 3
_default_prec = "fp32"


def set_default_precision(name: str) -> None:
    global _default_prec
    if name not in PRECISIONS:
        raise ValueError(f"unknown precision {name!r}: expected one of {sorted(PRECISIONS)}")
    _default_prec = name


def default_precision() -> str:
    return _default_prec


_PRODUCT_OPS = frozenset((
    "linear_fwd", "linear_dgrad", "linear_wgrad", "linear_wgrad_adam", "linear_wgrad_adam_dgrad", "linear_wgrad_sgdm",
    "conv1_fwd", "conv1_wgrad", "conv1_wgrad_scratch",
    "conv2_prep_alloc", "conv2_prep", "conv2_fwd", "conv2_wgrad_scratch", "conv2_wgrad_scratch_opt", "conv2_wgrad",
    "conv2_dgrad"))


class _Bound:
    """This module with ``prec`` pre-set on every matrix-product wrapper (what an engine holds)."""

    def __init__(self, prec: str):
        _prec(prec)
        self.precision = prec
        self._cache = {}

    def __getattr__(self, name):
        import functools
        import sys
        fn = self._cache.get(name)
        if fn is None:
            fn = getattr(sys.modules[__name__], name)
            if name in _PRODUCT_OPS:
                fn = functools.partial(fn, prec=self.precision)
            self._cache[name] = fn
        return fn


def bound(prec=None) -> "_Bound":
    return _Bound(_default_prec if prec is None else prec)


def _prec(prec) -> int:
    if prec is None:
        prec = _default_prec
    try:
        return PRECISIONS[prec]
    except KeyError:
        raise ValueError(f"unknown precision {prec!r}: expected one of {sorted(PRECISIONS)}") from None


class _Sig:
    p2pfl_linear_fwd = [P, L, P, L, P, L, P, L, I, I, I, I, I, P, L, P, L, I, P]
    p2pfl_linear_fwd_scratch = [I, I, I, I]
    p2pfl_linear_dgrad = [P, L, P, L, P, L, P, L, I, I, I, I, P, L, I, P]
    p2pfl_linear_dgrad_scratch = [I, I, I, I]
    p2pfl_linear_wgrad = [P, L, P, L, P, L, P, L, I, I, I, I, P, L, I, P]
    p2pfl_linear_wgrad_adam = [P, L, P, L, P, P, P, L, P, P, P, L, I, I, I, I, F, F, F, F, F, I, P, I, P, L, I, P]
    p2pfl_linear_wgrad_adam_dgrad = [P, L, P, L, P, P, P, L, P, P, P, L, P, L, I, I, I, I, F, F, F, F, F, I, P, I, P, L, I, P]
    p2pfl_linear_wgrad_sgdm = [P, L, P, L, P, P, L, P, P, L, I, I, I, I, F, F, F, I, P, L, I, P]
    p2pfl_conv1_fwd = [P, L, P, L, P, L, P, L, P, L, P, L, I, I, I, I, I, P]
    p2pfl_conv1_wgrad = [P, L, P, L, P, L, P, L, P, L, P, L, P, L, I, I, I, I, I, P, P]
    p2pfl_conv1_wgrad_scratch = [I, I, I, I]
    p2pfl_conv2_wgrad_opt = [P, L, P, L, P, L, P, L, P, L, P, L, I, I, I, I, I, I, I, P, P]
    p2pfl_linear_head = [P, L, P, L, P, L, P, L, P, L, P, L, P, L, P, L, P, L, P, P, P, P, P, L, I, I, I, I, I, P, P, L, P]
    p2pfl_linear_head_scratch = [I, I, I, I]
    p2pfl_linear_head_accepts = [I, I, I]
    p2pfl_linear_wgrad_adam_dgrad_accepts = [I, I, I, I]
    p2pfl_conv2_prep = [P, L, P, L, I, I, I, I, I, P]
    p2pfl_conv2_fwd = [P, L, P, L, P, L, P, L, P, L, I, I, I, I, I, I, I, P]
    p2pfl_conv2_wgrad = [P, L, P, L, P, L, P, L, P, L, P, L, I, I, I, I, I, I, I, P]
    p2pfl_conv2_wgrad_scratch = [I, I, I, I]
    p2pfl_conv2_dgrad = [P, L, P, L, P, L, P, L, I, I, I, I, I, I, I, P]
    p2pfl_softmax_ce = [P, L, P, L, P, L, P, L, P, P, P, P, P, L, I, I, I, P]
    p2pfl_mse_input = [P, L, P, L, P, L, P, L, P, P, P, L, I, I, I, P]
    p2pfl_zero_rows = [P, L, I, I, P]
    p2pfl_adam_step = [P, P, P, P, L, L, L, I, F, F, F, F, F, I, P, P]
    p2pfl_adam_step2 = [P, P, P, P, L, L, L, L, I, F, F, F, F, F, I, P, P]
    p2pfl_sgdm_step = [P, P, P, L, L, L, I, F, F, F, P]
    p2pfl_fedavg = [P, L, I, P, L, P, L, P, P, P, P, P, I, L, P]
    p2pfl_fedavg_inplace = [P, L, I, P, L, I, P, P, P, P, P, I, L, P]
    p2pfl_fedavg_inplace_chunk = [I]
    p2pfl_aes128_ecb = [P, P, P, L, I, P]


class OptStruct(ctypes.Structure):
    """``P2pflOpt`` of csrc/opt_epi.h: the optimizer step a gradient-producing kernel applies in its epilogue."""
    _fields_ = [("kind", c_int), ("t", c_int),
                ("pw", c_void_p), ("mw", c_void_p), ("vw", c_void_p),
                ("pb", c_void_p), ("mb", c_void_p), ("vb", c_void_p),
                ("cs", c_long),
                ("lr", c_float), ("beta1", c_float), ("beta2", c_float), ("eps", c_float),
                ("weight_decay", c_float), ("momentum", c_float),
                ("bc", c_void_p)]


def fold_opt(kind: str, w, mw, vw, b, mb, vb, *, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, momentum=0.9,
             t: int = 1, bc=None) -> OptStruct:
    """Optimizer epilogue for one layer: ``w/mw/vw`` [K,...] and ``b/mb/vb`` [K,Out] are views of the parameter /
    moment slabs (``vw``/``vb`` None for SGD-momentum) that share ONE row stride. ``kind``: "adam" | "sgd"."""
    cs = w.stride(0)
    views = [w, mw, b, mb] + ([vw, vb] if kind == "adam" else [])
    for v in views:
        _t(v)
        if v.shape[0] > 1 and v.stride(0) != cs:        # (one client: the stride of a size-1 dimension means nothing)
            raise HipKernelError("optimizer epilogue: all parameter / moment views must share one row stride")
    ptr = lambda v: None if v is None else v.data_ptr()
    return OptStruct(1 if kind == "adam" else 2, int(t), ptr(w), ptr(mw), ptr(vw) if kind == "adam" else None,
                     ptr(b), ptr(mb), ptr(vb) if kind == "adam" else None, cs, lr, beta1, beta2, eps, weight_decay,
                     momentum, None if bc is None else _t(bc).data_ptr())


def _optp(opt):
    return None if opt is None else ctypes.byref(opt)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _chk(code: int, what: str) -> None:
    if code != 0:
        raise HipKernelError(f"{what} failed with code {code}")


def _t(t: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
    if not t.is_cuda:
        raise HipKernelError("HIP kernels need tensors on the GPU")
    if t.dtype != dtype:
        raise HipKernelError(f"expected {dtype}, got {t.dtype}")
    # all dims but the first must be densely packed
    exp = 1
    for size, stride in zip(reversed(t.shape[1:]), reversed(t.stride()[1:])):
        if size != 1 and stride != exp:
            raise HipKernelError(f"inner dimensions must be contiguous, got shape {tuple(t.shape)} stride {t.stride()}")
        exp *= size
    return t


def _pc(t: Optional[torch.Tensor], dtype=torch.float32):
    """(pointer, client stride in elements) of a [K, ...] tensor; (None, 0) for None."""
    if t is None:
        return None, 0
    _t(t, dtype)
    return t.data_ptr(), (t.stride(0) if t.dim() > 0 else 0)


# ----------------------------------------------------------------------------- linear
_fwd_scratch = {}


def linear_fwd_scratch(K: int, B: int, In: int, Out: int, device) -> Optional[torch.Tensor]:
    """Scratch buffer :func:`linear_fwd` wants for this problem (None: none) - skinny heads with few clients split
    their reduction over several blocks whose partial tiles go through it."""
    need = int(load().p2pfl_linear_fwd_scratch(K, B, In, Out))
    return torch.empty(need, dtype=torch.float32, device=device) if need else None


def linear_fwd(x, w, b, y, relu: bool, sidx=None, scratch: Optional[torch.Tensor] = None, prec=None) -> None:
    """y[k] = act(x[k] @ w[k].T + b[k]);  x [K,B,In], w [K,Out,In], b [K,Out], y [K,B,Out].

    ``sidx`` [K,B] int32: x is the whole data set [K,n,In] and row r of the product is x[k, sidx[k, r]] (the
    mini-batch gather of a first dense layer happens in the operand loader). ``scratch``: buffer from
    :func:`linear_fwd_scratch` owned by the caller; without it one per (device, stream, thread) is kept here."""
    K, B, Out = y.shape
    In = x.shape[2]
    assert w.shape == (K, Out, In) and b.shape == (K, Out) and (sidx is not None or x.shape[1] == B)
    (xp, xs), (wp, ws), (bp, bs), (yp, ys), (sp, ss) = _pc(x), _pc(w), _pc(b), _pc(y), _pc(sidx, torch.int32)
    assert sidx is None or sidx.shape == (K, B)
    lib = load()
    stream = _stream()
    need = int(lib.p2pfl_linear_fwd_scratch(K, B, In, Out))
    cp = 0
    if need:
        if scratch is not None and scratch.numel() >= need:
            buf = _t(scratch)
        else:
            key = (y.device, int(stream or 0), threading.get_ident())
            buf = _fwd_scratch.get(key)
            if buf is None or buf.numel() < need:
                buf = _fwd_scratch[key] = torch.empty(need, dtype=torch.float32, device=y.device)
        cp = buf.data_ptr()
    _chk(lib.p2pfl_linear_fwd(xp, xs, wp, ws, bp, bs, yp, ys, K, B, In, Out, int(relu), sp, ss, cp, need,
                              _prec(prec), stream), "linear_fwd")


def linear_dgrad_scratch(K: int, B: int, In: int, Out: int, device) -> Optional[torch.Tensor]:
    """Scratch buffer :func:`linear_dgrad` wants for this problem (None: none) - shards with too few tiles to fill the
    chip split the reduction over several workgroups whose partial tiles go through it."""
    need = int(load().p2pfl_linear_dgrad_scratch(K, B, In, Out))
    return torch.empty(need, dtype=torch.float32, device=device) if need else None


def linear_dgrad(dy, w, act, dx, scratch: Optional[torch.Tensor] = None, prec=None) -> None:
    """dx[k] = (dy[k] @ w[k]) * (act[k] > 0)   (act None: no mask). ``scratch``: buffer from
    :func:`linear_dgrad_scratch` owned by the caller; without it one per (device, stream, thread) is kept here."""
    K, B, Out = dy.shape
    In = w.shape[2]
    assert w.shape == (K, Out, In) and dx.shape == (K, B, In)
    (dp, ds), (wp, ws), (ap, as_), (xp, xs) = _pc(dy), _pc(w), _pc(act), _pc(dx)
    lib = load()
    stream = _stream()
    need = int(lib.p2pfl_linear_dgrad_scratch(K, B, In, Out))
    cp = 0
    if need:
        if scratch is not None and scratch.numel() >= need:
            buf = _t(scratch)
        else:
            key = (dx.device, int(stream or 0), threading.get_ident(), "dgrad")
            buf = _fwd_scratch.get(key)
            if buf is None or buf.numel() < need:
                buf = _fwd_scratch[key] = torch.empty(need, dtype=torch.float32, device=dx.device)
        cp = buf.data_ptr()
    _chk(lib.p2pfl_linear_dgrad(dp, ds, wp, ws, ap, as_, xp, xs, K, B, In, Out, cp, need, _prec(prec), stream),
         "linear_dgrad")


def linear_wgrad(dy, x, dw, db, sidx=None, prec=None) -> None:
    """dw[k] = dy[k].T @ x[k];  db[k] = dy[k].sum(0).   (``sidx``: batch rows gathered from x, see linear_fwd)"""
    K, B, Out = dy.shape
    In = x.shape[2]
    assert dw.shape == (K, Out, In) and db.shape == (K, Out) and (sidx is None or sidx.shape == (K, B))
    (dp, ds), (xp, xs), (wp, ws), (bp, bs), (sp, ss) = _pc(dy), _pc(x), _pc(dw), _pc(db), _pc(sidx, torch.int32)
    _chk(load().p2pfl_linear_wgrad(dp, ds, xp, xs, wp, ws, bp, bs, K, B, In, Out, sp, ss, _prec(prec), _stream()),
         "linear_wgrad")


def linear_wgrad_adam(dy, x, w, mw, vw, b, mb, vb, *, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0,
                      t: int, bc=None, fresh: bool = False, sidx=None, prec=None) -> None:
    """Fused ``dw = dy.T @ x`` + Adam update of (w, b) and their moments; the gradient is never
    written to memory. w/mw/vw are [K,Out,In] slab views with one common row stride, b/mb/vb [K,Out].
    ``fresh``: first step of a new optimizer - the moments count as zero and are not read (only written)."""
    K, B, Out = dy.shape
    In = x.shape[2]
    assert w.shape == (K, Out, In) and b.shape == (K, Out)
    assert w.stride(0) == mw.stride(0) == vw.stride(0) and b.stride(0) == mb.stride(0) == vb.stride(0)
    (dp, ds), (xp, xs), (sp, ss) = _pc(dy), _pc(x), _pc(sidx, torch.int32)
    _chk(load().p2pfl_linear_wgrad_adam(dp, ds, xp, xs, _t(w).data_ptr(), _t(mw).data_ptr(), _t(vw).data_ptr(),
                                        w.stride(0), _t(b).data_ptr(), _t(mb).data_ptr(), _t(vb).data_ptr(),
                                        b.stride(0), K, B, In, Out, lr, beta1, beta2, eps, weight_decay, t,
                                        None if bc is None else _t(bc).data_ptr(), int(fresh), sp, ss, _prec(prec),
                                        _stream()), "linear_wgrad_adam")


def linear_wgrad_adam_dgrad(dy, x, w, mw, vw, b, mb, vb, dx, *, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0,
                            t: int, bc=None, fresh: bool = False, sidx=None, prec=None) -> bool:
    """:func:`linear_wgrad_adam` that also writes the layer's data gradient ``dx = dy @ w`` ([K,B,In], no mask) from
    the weights it reads anyway - one pass over the weights less than ``linear_dgrad`` + ``linear_wgrad_adam``.
    fp32 policies only. Returns False (nothing launched) when the shape is not covered: run the two kernels then."""
    K, B, Out = dy.shape
    In = x.shape[2]
    assert w.shape == (K, Out, In) and b.shape == (K, Out) and dx.shape == (K, B, In)
    assert w.stride(0) == mw.stride(0) == vw.stride(0) and b.stride(0) == mb.stride(0) == vb.stride(0)
    if _prec(prec) == PRECISIONS["bf16"]:
        return False
    (dp, ds), (xp, xs), (sp, ss), (gp, gs) = _pc(dy), _pc(x), _pc(sidx, torch.int32), _pc(dx)
    rc = load().p2pfl_linear_wgrad_adam_dgrad(dp, ds, xp, xs, _t(w).data_ptr(), _t(mw).data_ptr(), _t(vw).data_ptr(),
                                              w.stride(0), _t(b).data_ptr(), _t(mb).data_ptr(), _t(vb).data_ptr(),
                                              b.stride(0), gp, gs, K, B, In, Out, lr, beta1, beta2, eps, weight_decay, t,
                                              None if bc is None else _t(bc).data_ptr(), int(fresh), sp, ss,
                                              _prec(prec), _stream())
    if rc == -4:
        return False
    _chk(rc, "linear_wgrad_adam_dgrad")
    return True


def linear_wgrad_sgdm(dy, x, w, mw, b, mb, *, lr, momentum=0.9, weight_decay=0.0, fresh: bool = False,
                      sidx=None, prec=None) -> None:
    """Fused ``dw = dy.T @ x`` + SGD-momentum update of (w, b) and their momentum buffers (same layout rules
    as :func:`linear_wgrad_adam`)."""
    K, B, Out = dy.shape
    In = x.shape[2]
    assert w.shape == (K, Out, In) and b.shape == (K, Out)
    assert w.stride(0) == mw.stride(0) and b.stride(0) == mb.stride(0)
    (dp, ds), (xp, xs), (sp, ss) = _pc(dy), _pc(x), _pc(sidx, torch.int32)
    _chk(load().p2pfl_linear_wgrad_sgdm(dp, ds, xp, xs, _t(w).data_ptr(), _t(mw).data_ptr(), w.stride(0),
                                        _t(b).data_ptr(), _t(mb).data_ptr(), b.stride(0), K, B, In, Out, lr,
                                        momentum, weight_decay, int(fresh), sp, ss, _prec(prec), _stream()),
         "linear_wgrad_sgdm")


# ----------------------------------------------------------------------------- conv
def conv1_fwd(data, sidx, w, bias, a1, code, ks: int, prec=None) -> None:
    """Stage-1 conv(1->F, ks x ks, same) + ReLU + 2x2 maxpool.

    data [K,n,784] fp32, sidx [K,B] int32 or None, w [K,F,ks*ks], bias [K,F],
    a1 [K,B,F,196] fp32 out, code [K,B,F,196] uint8 out.
    """
    K, B, Fm = a1.shape[0], a1.shape[1], a1.shape[2]
    (dp, ds), (sp, ss), (wp, ws), (bp, bs), (ap, as_) = _pc(data), _pc(sidx, torch.int32), _pc(w), _pc(bias), _pc(a1)
    cp, cs = _pc(code, torch.uint8)
    _chk(load().p2pfl_conv1_fwd(dp, ds, sp, ss, wp, ws, bp, bs, ap, as_, cp, cs, K, B, Fm, ks, _prec(prec), _stream()),
         "conv1_fwd")


_c1_scratch = {}


def conv1_wgrad_scratch(K: int, B: int, ks: int, device, prec=None) -> torch.Tensor:
    """Scratch buffer of :func:`conv1_wgrad` (per-wave partial tiles) for this problem size."""
    need = int(load().p2pfl_conv1_wgrad_scratch(K, B, ks, _prec(prec)))
    return torch.empty(max(need, 1), dtype=torch.float32, device=device)


def conv1_wgrad(g, code, data, sidx, dw, db, ks: int, scratch: Optional[torch.Tensor] = None, opt: Optional[OptStruct] = None,
                prec=None) -> None:
    """Stage-1 weight / bias gradient: per-wave partial tiles go through ``scratch`` and a finishing pass adds them up
    in a fixed order (no atomics: bitwise reproducible), then stores dw / db - or, with ``opt`` (:func:`fold_opt`),
    applies the layer's optimizer step instead (dw / db may then be None)."""
    K, B, Fm = g.shape[0], g.shape[1], g.shape[2]
    (gp, gs), (dp, ds), (sp, ss), (wp, ws), (bp, bs) = _pc(g), _pc(data), _pc(sidx, torch.int32), _pc(dw), _pc(db)
    cp, cs = _pc(code, torch.uint8)
    lib = load()
    need = int(lib.p2pfl_conv1_wgrad_scratch(K, B, ks, _prec(prec)))
    if scratch is not None and scratch.numel() >= need:
        buf = _t(scratch)
    else:
        key = (g.device, int(_stream() or 0), threading.get_ident())
        buf = _c1_scratch.get(key)
        if buf is None or buf.numel() < need:
            buf = _c1_scratch[key] = torch.empty(need, dtype=torch.float32, device=g.device)
    _chk(lib.p2pfl_conv1_wgrad(gp, gs, cp, cs, dp, ds, sp, ss, wp, ws, bp, bs, buf.data_ptr(), buf.numel(), K, B, Fm, ks,
                               _prec(prec), _optp(opt), _stream()), "conv1_wgrad")


def _prep_dtype(prec):
    return torch.float32 if _prec(prec) == PRECISIONS["fp32-mfma"] else torch.bfloat16


def _prep_terms(prec) -> int:
    return SPLIT_TERMS if _prec(prec) == PRECISIONS["fp32"] else 1


def conv2_prep_alloc(K: int, ks: int, device, prec=None) -> torch.Tensor:
    """Workspace of :func:`conv2_prep`: per client a tap-major copy (fp32 or bf16, by ``prec``) of the stage-2
    weights for the forward pass followed by the mirrored, transposed copy for the data gradient."""
    return torch.empty(K, _prep_terms(prec) * 2 * ks * ks * 64 * 32, dtype=_prep_dtype(prec), device=device)


def conv2_prep(w, wprep, ks: int, prec=None) -> None:
    """w [K,Cout=64,Cin*ks*ks] fp32 -> wprep (see conv2_prep_alloc). Run once per weight update."""
    K, Cout = w.shape[0], w.shape[1]
    Cin = w.shape[2] // (ks * ks)
    (wp, ws), (pp, ps) = _pc(w), _pc(wprep, _prep_dtype(prec))
    _chk(load().p2pfl_conv2_prep(wp, ws, pp, ps, K, Cin, Cout, ks, _prec(prec), _stream()), "conv2_prep")


def conv2_fwd(a1, w, bias, a2, code, ks: int, wprep=None, prec=None, coresident: bool = False) -> None:
    """Stage-2 conv(Cin->Cout, ks x ks, same) + ReLU + 2x2 maxpool on the matrix cores.
    ``coresident`` (fp32 only): the small-footprint kernel variant that fits on a CU beside the fused wgrad+Adam
    kernel - what the two-stream training step launches on its side stream.

    a1 [K,B,Cin,196], w [K,Cout,Cin*ks*ks], bias [K,Cout], a2 [K,B,Cout,49], code uint8 like a2.
    ``wprep``: workspace already filled by :func:`conv2_prep` for these weights (else made here).
    """
    K, B, Cin = a1.shape[0], a1.shape[1], a1.shape[2]
    Cout = w.shape[1]
    if wprep is None:
        wprep = conv2_prep_alloc(K, ks, a1.device, prec)
        conv2_prep(w, wprep, ks, prec)
    (ap, as_), (pp, ps), (bp, bs), (op, os_) = _pc(a1), _pc(wprep, _prep_dtype(prec)), _pc(bias), _pc(a2)
    cp, cs = _pc(code, torch.uint8)
    _chk(load().p2pfl_conv2_fwd(ap, as_, pp, ps, bp, bs, op, os_, cp, cs, K, B, Cin, Cout, ks, _prec(prec),
                                int(coresident), _stream()), "conv2_fwd")


_wgrad_scratch = {}


def conv2_wgrad_scratch(K: int, B: int, ks: int, device, prec=None) -> Optional[torch.Tensor]:
    """Scratch buffer :func:`conv2_wgrad` wants for this problem size (None: it needs none). With few clients
    the batch is split over more blocks and their partial sums go through this buffer."""
    need = int(load().p2pfl_conv2_wgrad_scratch(K, B, ks, _prec(prec)))
    return torch.empty(need, dtype=torch.float32, device=device) if need else None


def conv2_wgrad_scratch_opt(K: int, B: int, ks: int, device, prec=None) -> torch.Tensor:
    """Scratch of :func:`conv2_wgrad` with an optimizer epilogue: the partial-sum pass is then always taken."""
    need = max(int(load().p2pfl_conv2_wgrad_scratch(K, B, ks, _prec(prec))), K * (ks * ks * 64 * 32 + 64))
    return torch.empty(need, dtype=torch.float32, device=device)


def conv2_wgrad(g, code, a1, dw, db, ks: int, scratch: Optional[torch.Tensor] = None, prec=None,
                coresident: bool = False, opt: Optional[OptStruct] = None) -> None:
    """Stage-2 weight/bias gradient: dw [K,Cout,Cin*ks*ks], db [K,Cout] (overwritten) - or, with ``opt``
    (:func:`fold_opt`), the layer's optimizer step applied by the reduction pass (dw / db may then be None).

    ``scratch``: buffer from :func:`conv2_wgrad_scratch` owned by the caller (one call in flight per buffer);
    without it a module-level buffer per (device, stream, thread) is used."""
    K, B, Cout = g.shape[0], g.shape[1], g.shape[2]
    Cin = a1.shape[2]
    (gp, gs), (ap, as_), (wp, ws), (bp, bs) = _pc(g), _pc(a1), _pc(dw), _pc(db)
    cp, cs = _pc(code, torch.uint8)
    lib = load()
    stream = _stream()
    need = int(lib.p2pfl_conv2_wgrad_scratch(K, B, ks, _prec(prec)))
    if opt is not None:
        need = max(need, K * (ks * ks * Cout * Cin + Cout))
    sp = 0
    if need:
        if scratch is not None and scratch.numel() >= need:
            buf = _t(scratch)
        else:
            key = (g.device, int(stream or 0), threading.get_ident(), _prec(prec))
            buf = _wgrad_scratch.get(key)
            if buf is None or buf.numel() < need:
                buf = _wgrad_scratch[key] = torch.empty(need, dtype=torch.float32, device=g.device)
        sp = buf.data_ptr()
        need = buf.numel()
    _chk(lib.p2pfl_conv2_wgrad_opt(gp, gs, cp, cs, ap, as_, wp, ws, bp, bs, sp, need, K, B, Cin, Cout, ks, _prec(prec),
                                   int(coresident), _optp(opt), stream), "conv2_wgrad")


def conv2_dgrad(g, code, w, da1, ks: int, wprep=None, prec=None, coresident: bool = False) -> None:
    K, B, Cout = g.shape[0], g.shape[1], g.shape[2]
    Cin = da1.shape[2]
    if wprep is None:
        wprep = conv2_prep_alloc(K, ks, g.device, prec)
        conv2_prep(w, wprep, ks, prec)
    (gp, gs), (pp, ps), (dp, ds) = _pc(g), _pc(wprep, _prep_dtype(prec)), _pc(da1)
    cp, cs = _pc(code, torch.uint8)
    _chk(load().p2pfl_conv2_dgrad(gp, gs, cp, cs, pp, ps, dp, ds, K, B, Cin, Cout, ks, _prec(prec), int(coresident),
                                  _stream()), "conv2_dgrad")


# ----------------------------------------------------------------------------- loss / misc
def softmax_ce(logits, labels, sidx, dlogits, loss_sum, correct, count, confusion=None, n_valid=None) -> None:
    """Fused softmax-CE fwd/bwd + metric counters. logits [K,B,C]; labels [K,n] int32."""
    K, B, C = logits.shape
    (lp, ls), (yp, ys), (sp, ss), (dp, ds) = _pc(logits), _pc(labels, torch.int32), _pc(sidx, torch.int32), _pc(dlogits)
    nvp, nvs = _pc(n_valid, torch.int32)
    _chk(load().p2pfl_softmax_ce(lp, ls, yp, ys, sp, ss, dp, ds, _t(loss_sum).data_ptr(),
                                 _t(correct, torch.int32).data_ptr(), _t(count, torch.int32).data_ptr(),
                                 None if confusion is None else _t(confusion, torch.int32).data_ptr(),
                                 nvp, nvs, K, B, C, _stream()), "softmax_ce")


def linear_wgrad_adam_dgrad_accepts(K: int, B: int, In: int, Out: int) -> bool:
    """Would :func:`linear_wgrad_adam_dgrad` take K clients of this layer (its chip-filling rule, csrc/linear.hip)?"""
    return bool(load().p2pfl_linear_wgrad_adam_dgrad_accepts(K, B, In, Out))


def linear_head_accepts(B: int, In: int, C: int) -> bool:
    return bool(load().p2pfl_linear_head_accepts(B, In, C))


def linear_head_scratch(K: int, B: int, In: int, C: int, device) -> Optional[torch.Tensor]:
    """Scratch buffer with which :func:`linear_head` takes its two-kernel form over (client, column slice) grids (wide
    heads: In >= 1024); None: the one-kernel form covers this shape."""
    need = int(load().p2pfl_linear_head_scratch(K, B, In, C))
    return torch.empty(need, dtype=torch.float32, device=device) if need else None


def linear_head(x, w, b, labels, sidx, loss_sum, correct, count, confusion=None, n_valid=None, *, dx=None, dw=None,
                db=None, logits=None, relu_mask: bool = False, opt: Optional[OptStruct] = None,
                scratch: Optional[torch.Tensor] = None) -> None:
    """The classification head of a training step (csrc/head.hip): logits = x w^T + b, softmax cross-entropy + metric
    counters, ``dx = (dlogits w) * [x > 0 if relu_mask]``, and either the gradients ``dw`` / ``db`` or - with ``opt``
    (:func:`fold_opt` over these very ``w`` / ``b`` views) - the optimizer step in place. One kernel per client, or
    - with ``scratch`` from :func:`linear_head_scratch` - two kernels over column slices of the clients.
    x [K,B,In] (B <= 32), w [K,C,In] (C <= 16), b [K,C], labels [K,n] int32, dx [K,B,In] or None."""
    K, B, In = x.shape
    C = w.shape[1]
    assert w.shape == (K, C, In) and b.shape == (K, C)
    assert dx is None or dx.shape == (K, B, In)
    (xp, xs), (wp, ws), (bp, bs), (gp, gs) = _pc(x), _pc(w), _pc(b), _pc(dx)
    (dwp, dws), (dbp, dbs), (lp, ls) = _pc(dw), _pc(db), _pc(logits)
    (yp, ys), (sp, ss), (nvp, nvs) = _pc(labels, torch.int32), _pc(sidx, torch.int32), _pc(n_valid, torch.int32)
    _chk(load().p2pfl_linear_head(xp, xs, wp, ws, bp, bs, gp, gs, dwp, dws, dbp, dbs, lp, ls, yp, ys, sp, ss,
                                  _t(loss_sum).data_ptr(), _t(correct, torch.int32).data_ptr(),
                                  _t(count, torch.int32).data_ptr(),
                                  None if confusion is None else _t(confusion, torch.int32).data_ptr(),
                                  nvp, nvs, K, B, In, C, int(relu_mask), _optp(opt),
                                  None if scratch is None else scratch.data_ptr(),
                                  0 if scratch is None else scratch.numel(), _stream()), "linear_head")


def mse_input(out, data, sidx, dout, loss_sum, count, n_valid=None) -> None:
    """Reconstruction loss fwd/bwd of the auto-encoders: MSE between ``out`` [K,B,D] and the batch's INPUT rows, gathered
    from ``data`` [K,n,D] with ``sidx`` [K,B] (None: rows 0..B-1). ``dout`` None: evaluation. ``loss_sum`` += mean loss
    of the batch x its samples, ``count`` += samples."""
    K, B, D = out.shape
    if data.shape[2] != D or data.stride(1) != D:
        raise HipKernelError(f"mse_input: data rows of {data.shape[2]} (stride {data.stride(1)}) for outputs of {D}")
    (op, os_), (xp, xs), (sp, ss), (dp, ds) = _pc(out), _pc(data), _pc(sidx, torch.int32), _pc(dout)
    nvp, nvs = _pc(n_valid, torch.int32)
    _chk(load().p2pfl_mse_input(op, os_, xp, xs, sp, ss, dp, ds, _t(loss_sum).data_ptr(),
                                _t(count, torch.int32).data_ptr(), nvp, nvs, K, B, D, _stream()), "mse_input")


def zero_rows(t) -> None:
    """Zero a [K, ...] (possibly slab-strided) view."""
    K = t.shape[0]
    n = t[0].numel()
    p, s = _pc(t)
    _chk(load().p2pfl_zero_rows(p, s, n, K, _stream()), "zero_rows")


# ----------------------------------------------------------------------------- optimizers
def adam_step(p, g, m, v, *, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, t: int, bc=None) -> None:
    """One Adam step over [K, P] slabs; p / m / v share one row stride, the gradient slab may have its own.

    ``bc``: optional device tensor [2] = (1/(1-beta1^t), 1/sqrt(1-beta2^t)) read by the kernel at
    run time instead of the host-computed scalars (lets a captured graph be replayed for any t)."""
    K, n = p.shape
    cs = p.stride(0)
    assert m.stride(0) == cs and v.stride(0) == cs and g.shape == p.shape and g.stride(1) == 1
    _chk(load().p2pfl_adam_step(_t(p).data_ptr(), _t(g).data_ptr(), _t(m).data_ptr(), _t(v).data_ptr(), cs, g.stride(0), n, K,
                                lr, beta1, beta2, eps, weight_decay, t,
                                None if bc is None else _t(bc).data_ptr(), _stream()), "adam_step")


def adam_step_ranges(p, g, m, v, ranges, *, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, t: int,
                     bc=None) -> None:
    """Adam step over the column ranges ``[(a0, b0), (a1, b1)]`` (a0 < b0 <= a1 < b1) of [K, P] slabs in ONE launch."""
    (a0, b0), (a1, b1) = ranges
    assert 0 <= a0 < b0 <= a1 < b1 <= p.shape[1]
    cs = p.stride(0)
    assert g.stride(0) == cs and m.stride(0) == cs and v.stride(0) == cs and g.shape == p.shape
    _chk(load().p2pfl_adam_step2(_t(p[:, a0:]).data_ptr(), _t(g[:, a0:]).data_ptr(), _t(m[:, a0:]).data_ptr(),
                                 _t(v[:, a0:]).data_ptr(), cs, b0 - a0, a1 - b0, b1 - a1, p.shape[0], lr, beta1, beta2,
                                 eps, weight_decay, t, None if bc is None else _t(bc).data_ptr(), _stream()),
         "adam_step_ranges")


def sgdm_step(p, g, buf, *, lr, momentum=0.9, weight_decay=0.0) -> None:
    K, n = p.shape
    cs = p.stride(0)
    assert buf.stride(0) == cs and g.shape == p.shape and g.stride(1) == 1
    _chk(load().p2pfl_sgdm_step(_t(p).data_ptr(), _t(g).data_ptr(), _t(buf).data_ptr(), cs, g.stride(0), n, K, lr, momentum,
                                weight_decay, _stream()), "sgdm_step")


# ----------------------------------------------------------------------------- FedAvg
def fedavg(local, remote, out, src_ptr, src_row, src_w, dst_ptr, dst_row) -> None:
    """Grouped weighted row reduce (see csrc/fedavg.hip). `remote` may be None."""
    n_local, n = local.shape
    G = src_ptr.numel() - 1
    rp, rs = (remote.data_ptr(), remote.stride(0)) if remote is not None and remote.numel() else (local.data_ptr(), local.stride(0))
    _chk(load().p2pfl_fedavg(_t(local).data_ptr(), local.stride(0), n_local, rp, rs, _t(out).data_ptr(),
                             out.stride(0), _t(src_ptr, torch.int32).data_ptr(),
                             _t(src_row, torch.int32).data_ptr(), _t(src_w).data_ptr(),
                             _t(dst_ptr, torch.int32).data_ptr(), _t(dst_row, torch.int32).data_ptr(), G, n,
                             _stream()), "fedavg")


def fedavg_inplace(slab, remote, src_ptr, src_row, src_w, dst_ptr, dst_row) -> bool:
    """:func:`fedavg` with ``out`` = ``slab`` itself: every source row (slab rows, then ``remote`` rows) is read once,
    whatever the number of groups using it, and the destinations are overwritten in place (csrc/fedavg.hip).
    False (nothing ran): too many rows for the kernel's LDS pool - reduce into a second slab with :func:`fedavg`."""
    n_local, n = slab.shape
    n_remote = int(remote.shape[0]) if remote is not None and remote.numel() else 0
    lib = load()
    if int(lib.p2pfl_fedavg_inplace_chunk(n_local + n_remote)) == 0:
        return False
    rp, rs = (remote.data_ptr(), remote.stride(0)) if n_remote else (None, 0)
    G = src_ptr.numel() - 1
    _chk(lib.p2pfl_fedavg_inplace(_t(slab).data_ptr(), slab.stride(0), n_local, rp, rs, n_remote,
                                  _t(src_ptr, torch.int32).data_ptr(), _t(src_row, torch.int32).data_ptr(),
                                  _t(src_w).data_ptr(), _t(dst_ptr, torch.int32).data_ptr(),
                                  _t(dst_row, torch.int32).data_ptr(), G, n, _stream()), "fedavg_inplace")
    return True


def aes128_ecb(src, dst, round_keys, decrypt: bool) -> None:
    """AES-128-ECB over a flat uint8 buffer (length % 16 == 0) with expanded round keys [176] uint8."""
    n = src.numel()
    assert n % 16 == 0 and dst.numel() == n and round_keys.numel() == 176
    _chk(load().p2pfl_aes128_ecb(_t(src, torch.uint8).data_ptr(), _t(dst, torch.uint8).data_ptr(),
                                 _t(round_keys, torch.uint8).data_ptr(), n // 16, int(decrypt), _stream()),
         "aes128_ecb")
