"""autograd bindings of the HIP kernels (one C-ABI call per forward, one per backward).

Everything here only enqueues work on torch's current HIP stream; nothing synchronises, so a whole
train step built from these functions can be captured into a hipGraph (torch.cuda.CUDAGraph).
"""
import itertools
import torch

from . import _lib
from ._lib import DTYPE_CODE, PARAM_DTYPE, check, ptr, stream


class RngState:
    """Perf-mode RNG coordinates handed to the kernels (include/embrace_hip.h, "RNG contract").

    seed      Philox key
    step_val  host-side part of the step counter
    step_dev  optional uint64 device scalar added to step_val (advances under graph replay)
    row0      global index of local row 0 (data-parallel shard offset)
    """
    __slots__ = ("seed", "step_val", "step_dev", "row0")

    def __init__(self, seed=0, step_val=0, step_dev=None, row0=0):
        self.seed, self.step_val, self.step_dev, self.row0 = int(seed), int(step_val), step_dev, int(row0)


def _as(t, dtype):
    t = t if t.dtype == dtype else t.to(dtype)
    return t if t.is_contiguous() else t.contiguous()


import weakref

# bf16 (or other compute-dtype) copies of master weights.  Kept per Parameter OBJECT (keyed by id, guarded by a
# weak reference -- tensors cannot be WeakKeyDictionary keys because == is elementwise); refreshed when the
# parameter's version counter or storage changes (torch optimizers), written in place by the fused optimizers
# (optim.py passes the registered shadow to the update kernel, so no cast kernel runs in steady state).
_SHADOWS = {}


def _shadow_entry(w):
    ent = _SHADOWS.get(id(w))
    if ent is not None and ent[0]() is not w:
        del _SHADOWS[id(w)]
        ent = None
    return ent


def weight_as(w, T):
    """`w` viewed in compute dtype T (contiguous).  Same dtype: the parameter itself."""
    if w.dtype == T:
        wd = w.detach()
        return wd if wd.is_contiguous() else wd.contiguous()
    ent = _shadow_entry(w)
    key = (w._version, w.data_ptr(), tuple(w.shape), T)
    if ent is not None and ent[1] == key:
        return ent[2]
    out = ent[2] if (ent is not None and ent[2].shape == w.shape and ent[2].dtype == T and ent[2].device == w.device) else None
    sh = cast(w.detach(), T, out=out)
    if len(_SHADOWS) > 4096:                       # drop entries of parameters that no longer exist
        for k in [k for k, v in _SHADOWS.items() if v[0]() is None]:
            del _SHADOWS[k]
    _SHADOWS[id(w)] = (weakref.ref(w), key, sh)
    return sh


def shadow_lookup(w):
    """bf16 shadow registered for parameter `w` (None if there is none or it belongs to other storage)."""
    ent = _shadow_entry(w)
    if ent is None or ent[2].dtype != torch.bfloat16:
        return None
    if ent[1][1] != w.data_ptr() or ent[1][2] != tuple(w.shape):
        return None
    return ent[2]


# Packed images of conv weights (tap-major wpack, tap-flipped wflip; csrc/convblock.hip) cached per Parameter object like
# the shadows above.  For bf16 or fp32 compute on fp32 masters the pair is registered with the library, whose fused optimizer
# launches then keep it current in place (emb_conv_pack_register): no pack launch per step.  Any other update of the
# parameter bumps its version counter and triggers a re-pack here.
_PACKS = {}


_PACK_FINALIZERS = set()          # ids of parameters that already carry the clean-up finalizer
_PACK_REGISTERED = {}             # id(parameter) -> data_ptr currently registered with the library (emb_conv_pack_register)


def _pack_drop(key, w_ptr=None):
    """Forget parameter `key`: drops its images and unregisters WHATEVER pointer is registered for it now (the storage
    may have moved since the finalizer was armed: model.to(), .float(), ...), plus `w_ptr` if given."""
    _PACKS.pop(key, None)
    _PACK_FINALIZERS.discard(key)
    ptrs = {q for q in (_PACK_REGISTERED.pop(key, None), w_ptr) if q}
    for q in ptrs:
        try:
            _lib.lib().emb_conv_pack_unregister(q)
        except Exception:
            pass


def conv_packed(w, T, cin_pad, need_flip):
    """(wpack [Cout][k*cin_pad], wflip [cin_pad][k*Cout] or None) of conv weight `w` [Cout][Cin][k] in dtype T."""
    Cout, Cin, k = w.shape
    ent = _PACKS.get(id(w))
    if ent is not None and ent[0]() is not w:
        _pack_drop(id(w), ent[1][1])
        ent = None
    key = (w._version, w.data_ptr(), tuple(w.shape), T, cin_pad)
    registered = T in (torch.bfloat16, torch.float32) and w.dtype == torch.float32   # only then does the optimizer launch maintain the images;
    if registered and ent is not None and ent[1] == key and (ent[3] is not None or not need_flip):   # else: re-pack every call
        return ent[2], ent[3]
    if ent is not None:
        _lib.lib().emb_conv_pack_unregister(ent[1][1])
        _PACK_REGISTERED.pop(id(w), None)
    dev = w.device
    reuse = ent is not None and ent[2].dtype == T and ent[2].shape == (Cout, k * cin_pad) and ent[2].device == dev
    wpack = ent[2] if reuse else torch.empty(Cout, k * cin_pad, dtype=T, device=dev)
    wflip = None
    if need_flip:
        wflip = ent[3] if (reuse and ent[3] is not None) else torch.empty(cin_pad, k * Cout, dtype=T, device=dev)
    wd = w.detach()
    wd = wd if wd.is_contiguous() else wd.contiguous()
    check(_lib.lib().emb_conv_pack_weight(ptr(wd), ptr(wpack), ptr(wflip), Cout, Cin, cin_pad, k, DTYPE_CODE[T], stream()),
          "emb_conv_pack_weight")
    if registered and wd.data_ptr() == w.data_ptr():
        check(_lib.lib().emb_conv_pack_register(w.data_ptr(), ptr(wpack), ptr(wflip), Cout, Cin, cin_pad, k, DTYPE_CODE[T]),
              "emb_conv_pack_register")
        _PACK_REGISTERED[id(w)] = w.data_ptr()
        if id(w) not in _PACK_FINALIZERS:           # the table entry must not outlive the parameter's storage
            _PACK_FINALIZERS.add(id(w))
            weakref.finalize(w, _pack_drop, id(w))  # (looks the registered pointer up when the parameter dies)
    _PACKS[id(w)] = (weakref.ref(w), key, wpack, wflip)
    return wpack, wflip


# Optional gradient sinks: when a parameter has a registered sink (dist.FlatGrads: a view into one flat
# buffer that also is its .grad), the backward kernels write the parameter gradient straight into it and
# autograd is told "no gradient" -- no per-parameter accumulate / copy kernels, and the data-parallel
# all-reduce runs in place on the flat buffer.  Every parameter is used once per step, so overwrite == accumulate.
_GRAD_SINKS = {}


def register_grad_sink(p, buf):
    _GRAD_SINKS[id(p)] = (weakref.ref(p), buf)


def clear_grad_sinks():
    _GRAD_SINKS.clear()


def grad_sink(p, dtype):
    ent = _GRAD_SINKS.get(id(p))
    if ent is None or ent[0]() is not p:
        return None
    buf = ent[1]
    return buf if (buf.dtype == dtype and buf.shape == p.shape and buf.device == p.device) else None


def _out(sink, shape, dtype, device):
    return sink if sink is not None else torch.empty(shape, dtype=dtype, device=device)


def select_prep(p, avail, B, rng=None, device_dropout=False, status=None):
    """EmbraceNetMultimodal.py:63-76,178-184 + torch.multinomial's cdf -> cdf0[B] (fp32, device)."""
    _lib.require_cuda(p, avail)
    p = _as(p, torch.float32)
    if p.dim() == 1:
        p = p.view(1, -1)
    if p.shape[-1] != 2:
        raise NotImplementedError("the HIP path implements the two-modality EmbraceNet (epigenomic + sequence)")
    if avail is not None:
        avail = _as(avail, torch.float32)
    rng = rng or RngState()
    cdf0 = torch.empty(B, dtype=torch.float32, device=p.device)
    if status is None:
        status = torch.zeros(1, dtype=torch.int32, device=p.device)
    check(_lib.lib().emb_select_prep(ptr(p), p.shape[0], ptr(avail), int(bool(device_dropout)), rng.seed, rng.step_val,
                                     ptr(rng.step_dev), rng.row0, ptr(cdf0), ptr(status), B, stream()),
          "emb_select_prep")
    return cdf0, status


class SelectInline:
    """Handed to `embrace` in place of the cdf0 vector: selection probabilities [1|B, 2] fp32, availabilities [B, 2] fp32 or
    None, the device-dropout switch and the sticky int32 status word -- the arguments of `select_prep`, evaluated inside the
    forward launch instead (emb_embrace_fwd_select)."""
    __slots__ = ("p", "avail", "device_dropout", "status")

    def __init__(self, p, avail, device_dropout, status):
        _lib.require_cuda(p, avail, status)
        p = _as(p, torch.float32)
        p = p.view(1, -1) if p.dim() == 1 else p
        if p.shape[-1] != 2:
            raise NotImplementedError("the HIP path implements the two-modality EmbraceNet (epigenomic + sequence)")
        self.p, self.avail = p, (None if avail is None else _as(avail, torch.float32))
        self.device_dropout, self.status = bool(device_dropout), status


class _EmbraceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x0, x1, w0, b0, w1, b1, cdf0, u, rng, compute_dtype):
        _lib.require_cuda(x0, x1, w0, w1)
        T = compute_dtype
        P = PARAM_DTYPE[T]
        B, d0 = x0.shape
        d1 = x1.shape[1]
        c = w0.shape[0]
        x0c, x1c = _as(x0, T), _as(x1, T)
        w0c, w1c = weight_as(w0, T), weight_as(w1, T)
        b0c, b1c = _as(b0.detach(), P), _as(b1.detach(), P)
        E = torch.empty(B, c, dtype=T, device=x0.device)
        code = torch.empty(B, c, dtype=torch.uint8, device=x0.device)
        if u is not None:
            u = _as(u, torch.float64)
            assert u.shape == (B, c)
        if isinstance(cdf0, SelectInline):                      # thresholds computed inside the launch (no emb_select_prep)
            s_ = cdf0
            check(_lib.lib().emb_embrace_fwd_select(ptr(x0c), ptr(x1c), ptr(w0c), ptr(b0c), ptr(w1c), ptr(b1c), ptr(s_.p),
                                                    s_.p.shape[0], ptr(s_.avail), int(s_.device_dropout), ptr(s_.status), ptr(u),
                                                    rng.seed, rng.step_val, ptr(rng.step_dev), rng.row0, ptr(E), ptr(code),
                                                    B, d0, d1, c, DTYPE_CODE[T], stream()), "emb_embrace_fwd_select")
        else:
            check(_lib.lib().emb_embrace_fwd(ptr(x0c), ptr(x1c), ptr(w0c), ptr(b0c), ptr(w1c), ptr(b1c), ptr(cdf0), ptr(u),
                                             rng.seed, rng.step_val, ptr(rng.step_dev), rng.row0, ptr(E), ptr(code),
                                             B, d0, d1, c, DTYPE_CODE[T], stream()), "emb_embrace_fwd")
        ctx.save_for_backward(x0c, x1c, w0c, w1c, code)
        # a hand-over of pre-masked gradients (_PREMASKED) must come from THIS forward: the code tensor carries a serial number
        # the head copies into its hand-over and the backward checks (pointers alone repeat once the allocator recycles them)
        _PREMASKED.clear()
        ctx.code_serial = code._emb_serial = next(_CODE_SERIAL)
        ctx.T = T
        ctx.in_dtypes = (x0.dtype, x1.dtype, w0.dtype, b0.dtype, w1.dtype, b1.dtype)
        ctx.sinks = tuple(grad_sink(q, P) for q in (w0, b0, w1, b1))
        ctx.ws_owner = w1
        ctx.mark_non_differentiable(code)
        ctx.set_materialize_grads(False)          # no zero tensor for the (non-differentiable) code output
        return E, code

    @staticmethod
    def backward(ctx, dE, _dcode):
        if dE is None:
            return (None,) * 10
        x0c, x1c, w0c, w1c, code = ctx.saved_tensors
        T = ctx.T
        P = PARAM_DTYPE[T]
        B, d0 = x0c.shape
        d1 = x1c.shape[1]
        c = w0c.shape[0]
        dev = x0c.device
        dE = _as(dE, T)
        need0, need1 = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dX0 = torch.empty(B, d0, dtype=T, device=dev) if need0 else None
        dX1 = torch.empty(B, d1, dtype=T, device=dev) if need1 else None
        sk = ctx.sinks
        dW0, db0 = _out(sk[0], (c, d0), P, dev), _out(sk[1], (c,), P, dev)
        dW1, db1 = _out(sk[2], (c, d1), P, dev), _out(sk[3], (c,), P, dev)
        ws = _workspace(dev, 1 << 24, "embrace", ctx.ws_owner)
        L_ = _lib.lib()
        # the producer of dE may have left the pre-masked gradients dD_m = dE * keep_m of THIS forward's code bytes (the fused
        # classifier head does: _HeadCEFn): then the backward is four plain GEMMs (csrc/gemm_jobs.h)
        pm = _PREMASKED.pop(dE.data_ptr(), None)
        if pm is not None and not (pm[3] == ctx.code_serial and pm[2] == code.data_ptr() and pm[0].shape == (B, c) and pm[0].dtype == T):
            pm = None
        if pm is not None:
            pm = pm[:3]
        ok = bool(L_.emb_embrace_bwd_masked_supported(B, d0, d1, c, DTYPE_CODE[T])) and (B * c) % 8 == 0
        if pm is None and T != torch.float32:
            ok = False                 # bf16: a separate mask launch costs more than the fragment masks of emb_embrace_bwd
        if ok and pm is None:          # another producer of dE (hidden post layers, a plain autograd loss): mask here, once
            pm = (torch.empty(B, c, dtype=T, device=dev), torch.empty(B, c, dtype=T, device=dev), code.data_ptr())
            check(L_.emb_embrace_premask(ptr(dE), ptr(code), ptr(pm[0]), ptr(pm[1]), B, c, DTYPE_CODE[T], stream()),
                  "emb_embrace_premask")
        if ok:
            check(L_.emb_embrace_bwd_masked(ptr(pm[0]), ptr(pm[1]), ptr(x0c), ptr(x1c), ptr(w0c), ptr(w1c), ptr(dX0), ptr(dX1),
                                            ptr(dW0), ptr(db0), ptr(dW1), ptr(db1), ptr(ws), ws.numel(), B, d0, d1, c,
                                            DTYPE_CODE[T], stream()), "emb_embrace_bwd_masked")
        else:
            check(L_.emb_embrace_bwd(ptr(dE), ptr(code), ptr(x0c), ptr(x1c), ptr(w0c), ptr(w1c), ptr(dX0), ptr(dX1),
                                     ptr(dW0), ptr(db0), ptr(dW1), ptr(db1), ptr(ws), ws.numel(), B, d0, d1, c,
                                     DTYPE_CODE[T], stream()), "emb_embrace_bwd")
        if _AFTER_EMBRACE_BWD is not None:
            _AFTER_EMBRACE_BWD()
        t = ctx.in_dtypes
        cast = lambda g, d: None if g is None else (g if g.dtype == d else g.to(d))
        ret = lambda g, d, sink: None if sink is not None else cast(g, d)
        return (cast(dX0, t[0]), cast(dX1, t[1]), ret(dW0, t[2], sk[0]), ret(db0, t[3], sk[1]), ret(dW1, t[4], sk[2]),
                ret(db1, t[5], sk[3]), None, None, None, None)


_AFTER_EMBRACE_BWD = None
_PREMASKED = {}      # dE.data_ptr() -> (dD0, dD1, code.data_ptr(), code serial): left by the producer of dE for the fusion layer's backward
_CODE_SERIAL = itertools.count(1)


def set_after_embrace_backward(fn):
    """`fn()` is called right after the fusion layer's backward kernels have been enqueued -- the point of the backward
    pass at which every gradient of the post stack, the head and the docking layers is on its way (a data-parallel trainer
    starts reducing them there, training.StepRunner).  None clears the hook."""
    global _AFTER_EMBRACE_BWD
    _AFTER_EMBRACE_BWD = fn


def embrace(x0, x1, w0, b0, w1, b1, cdf0, u=None, rng=None, compute_dtype=None):
    """Fused docking + ReLU + modality selection (EmbraceNetMultimodal.py:52-60, 80-88).
    cdf0: the vector from `select_prep`, or a `SelectInline` (thresholds computed inside the launch).
    returns (E [B,c], code [B,c] uint8 with bit0 = selected modality)."""
    compute_dtype = compute_dtype or x0.dtype
    return _EmbraceFn.apply(x0, x1, w0, b0, w1, b1, cdf0, u, rng or RngState(), compute_dtype)


class _EmbraceBypassFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x0, x1, cdf0, u, rng, compute_dtype):
        _lib.require_cuda(x0, x1)
        T = compute_dtype
        if x0.shape != x1.shape or x0.dim() != 2:
            raise ValueError("bypass_docking: both inputs must be [batch_size, embracement_size]")
        B, c = x0.shape
        x0c, x1c = _as(x0, T), _as(x1, T)
        E = torch.empty(B, c, dtype=T, device=x0.device)
        code = torch.empty(B, c, dtype=torch.uint8, device=x0.device)
        if u is not None:
            u = _as(u, torch.float64)
            assert u.shape == (B, c)
        if isinstance(cdf0, SelectInline):
            s_ = cdf0
            args = (None, ptr(s_.p), s_.p.shape[0], ptr(s_.avail), int(s_.device_dropout), ptr(s_.status))
        else:
            args = (ptr(cdf0), None, 0, None, 0, None)
        check(_lib.lib().emb_embrace_bypass_fwd(ptr(x0c), ptr(x1c), *args, ptr(u), rng.seed, rng.step_val, ptr(rng.step_dev),
                                                rng.row0, ptr(E), ptr(code), B, c, DTYPE_CODE[T], stream()),
              "emb_embrace_bypass_fwd")
        ctx.save_for_backward(code)
        ctx.T, ctx.in_dtypes = T, (x0.dtype, x1.dtype)
        ctx.mark_non_differentiable(code)
        ctx.set_materialize_grads(False)
        return E, code

    @staticmethod
    def backward(ctx, dE, _dcode):
        if dE is None:
            return (None,) * 6
        code, = ctx.saved_tensors
        T = ctx.T
        B, c = code.shape
        dE = _as(dE, T)
        dX0 = torch.empty(B, c, dtype=T, device=code.device) if ctx.needs_input_grad[0] else None
        dX1 = torch.empty(B, c, dtype=T, device=code.device) if ctx.needs_input_grad[1] else None
        check(_lib.lib().emb_embrace_bypass_bwd(ptr(dE), ptr(code), ptr(dX0), ptr(dX1), B, c, DTYPE_CODE[T], stream()),
              "emb_embrace_bypass_bwd")
        if _AFTER_EMBRACE_BWD is not None:      # (the data-parallel trainers start the first bucket's all-reduce here)
            _AFTER_EMBRACE_BWD()
        cast = lambda g, d: None if g is None else (g if g.dtype == d else g.to(d))
        return cast(dX0, ctx.in_dtypes[0]), cast(dX1, ctx.in_dtypes[1]), None, None, None, None


def embrace_bypass(x0, x1, cdf0, u=None, rng=None, compute_dtype=None):
    """Modality selection over ready-made docking outputs (EmbraceNet(bypass_docking=True), EmbraceNetMultimodal.py:54-55,
    63-88): E[b, j] = x_{idx[b, j]}[b, j].  Arguments and return value as `embrace` without the docking parameters."""
    return _EmbraceBypassFn.apply(x0, x1, cdf0, u, rng or RngState(), compute_dtype or x0.dtype)


def select_prep_m(p, avail, B, M, status=None):
    """EmbraceNetMultimodal.py:63-76 + torch.multinomial's cdf for M modalities -> cdf [B, M] (fp32, device)."""
    _lib.require_cuda(p, avail)
    p = _as(p, torch.float32)
    p = p.view(1, -1) if p.dim() == 1 else p
    if p.shape[-1] != M or p.shape[0] not in (1, B):
        raise ValueError("selection_probabilities must be [B, M] or [M]")
    if avail is not None:
        avail = _as(avail, torch.float32)
        if tuple(avail.shape) != (B, M):
            raise ValueError("availabilities must be [B, M]")
    cdf = torch.empty(B, M, dtype=torch.float32, device=p.device)
    if status is None:
        status = torch.zeros(1, dtype=torch.int32, device=p.device)
    check(_lib.lib().emb_select_prep_m(ptr(p), p.shape[0], ptr(avail), ptr(cdf), ptr(status), B, M, stream()),
          "emb_select_prep_m")
    return cdf, status


class _EmbraceSelectFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cdf, u, rng, compute_dtype, *xs):
        _lib.require_cuda(cdf, *xs)
        T, M = compute_dtype, len(xs)
        B, c = xs[0].shape
        if any(tuple(x.shape) != (B, c) for x in xs) or tuple(cdf.shape) != (B, M):
            raise ValueError("embrace_select: M inputs of [batch_size, embracement_size] and a cdf of [batch_size, M]")
        xc = [_as(x, T) for x in xs]
        E = torch.empty(B, c, dtype=T, device=xs[0].device)
        code = torch.empty(B, c, dtype=torch.uint8, device=xs[0].device)
        if u is not None:
            u = _as(u, torch.float64)
            assert u.shape == (B, c)
        D = (_ct.c_void_p * M)(*[ptr(x) for x in xc])
        check(_lib.lib().emb_embrace_select_fwd(D, M, ptr(_as(cdf, torch.float32)), ptr(u), rng.seed, rng.step_val,
                                                ptr(rng.step_dev), rng.row0, ptr(E), ptr(code), B, c, DTYPE_CODE[T], stream()),
              "emb_embrace_select_fwd")
        ctx.save_for_backward(code)
        ctx.T, ctx.in_dtypes = T, tuple(x.dtype for x in xs)
        ctx.mark_non_differentiable(code)
        ctx.set_materialize_grads(False)
        return E, code

    @staticmethod
    def backward(ctx, dE, _dcode):
        M = len(ctx.in_dtypes)
        if dE is None:
            return (None,) * (4 + M)
        code, = ctx.saved_tensors
        T = ctx.T
        B, c = code.shape
        dE = _as(dE, T)
        dD = [torch.empty(B, c, dtype=T, device=code.device) if ctx.needs_input_grad[4 + m] else None for m in range(M)]
        P = (_ct.c_void_p * M)(*[ptr(d) for d in dD])
        check(_lib.lib().emb_embrace_select_bwd(ptr(dE), ptr(code), P, M, B, c, DTYPE_CODE[T], stream()),
              "emb_embrace_select_bwd")
        if _AFTER_EMBRACE_BWD is not None:
            _AFTER_EMBRACE_BWD()
        cast = lambda g, d: None if g is None else (g if g.dtype == d else g.to(d))
        return (None, None, None, None) + tuple(cast(g, d) for g, d in zip(dD, ctx.in_dtypes))


def embrace_select(xs, cdf, u=None, rng=None, compute_dtype=None):
    """Modality selection over M docking outputs (EmbraceNetMultimodal.py:80-88): E[b, j] = xs[idx[b, j]][b, j] with
    idx drawn from the rows of `cdf` ([B, M], from `select_prep_m`).  returns (E [B,c], idx [B,c] uint8)."""
    return _EmbraceSelectFn.apply(cdf, u, rng or RngState(), compute_dtype or xs[0].dtype, *xs)


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, relu, dropout_p, layer_id, rng, compute_dtype):
        _lib.require_cuda(x, w, b)
        T = compute_dtype
        P = PARAM_DTYPE[T]
        B, K = x.shape
        N = w.shape[0]
        xc, wc, bc = _as(x, T), weight_as(w, T), _as(b.detach(), P)
        y = torch.empty(B, N, dtype=T, device=x.device)
        need_mask = bool(relu) or dropout_p > 0
        mask = torch.empty(B, N, dtype=torch.uint8, device=x.device) if need_mask else None
        check(_lib.lib().emb_linear_fwd(ptr(xc), ptr(wc), ptr(bc), ptr(y), ptr(mask), int(bool(relu)), float(dropout_p),
                                        int(layer_id), rng.seed, rng.step_val, ptr(rng.step_dev), rng.row0, B, K, N,
                                        DTYPE_CODE[T], stream()), "emb_linear_fwd")
        ctx.save_for_backward(xc, wc, mask)
        ctx.cfg = (T, bool(relu), float(dropout_p), x.dtype, w.dtype, b.dtype)
        ctx.layer_id = int(layer_id)
        ctx.ws_owner = w
        ctx.sinks = (grad_sink(w, P), grad_sink(b, P))
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, wc, mask = ctx.saved_tensors
        T, relu, dropout_p, tx, tw, tb = ctx.cfg
        P = PARAM_DTYPE[T]
        B, K = xc.shape
        N = wc.shape[0]
        dy = _as(dy, T)
        dx = torch.empty(B, K, dtype=T, device=xc.device) if ctx.needs_input_grad[0] else None
        sk = ctx.sinks
        dw, db = _out(sk[0], (N, K), P, xc.device), _out(sk[1], (N,), P, xc.device)
        # scratch: weight-gradient slabs of the batch slices; large fp32 layers also park the pre-masked gradient there (linear.hip)
        nbytes = max(1 << 22, B * N * 4 + 4 * N * (K + 4) * 4 + 4096) if T == torch.float32 else 1 << 22
        ws = _workspace(xc.device, nbytes, f"linear{ctx.layer_id}", ctx.ws_owner)
        check(_lib.lib().emb_linear_bwd(ptr(dy), ptr(mask), ptr(xc), ptr(wc), ptr(dx), ptr(dw), ptr(db), int(relu),
                                        dropout_p, ptr(ws), ws.numel(), B, K, N, DTYPE_CODE[T], stream()), "emb_linear_bwd")
        cast = lambda g, d: None if g is None else (g if g.dtype == d else g.to(d))
        return (cast(dx, tx), None if sk[0] is not None else cast(dw, tw), None if sk[1] is not None else cast(db, tb),
                None, None, None, None, None)


def linear(x, w, b, relu=False, dropout_p=0.0, layer_id=0, rng=None, compute_dtype=None):
    """dropout(relu(x w^T + b)) with the epilogue fused (EmbraceNetMultimodal.py:143-151)."""
    return _LinearFn.apply(x, w, b, relu, dropout_p, layer_id, rng or RngState(), compute_dtype or x.dtype)


import ctypes as _ct


def _parr(tensors):
    arr = (_ct.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


_RIDER_KEEP = []     # tensors a parked rider launch reads or writes: kept alive until the flush
_JOBS_KEEP = []      # the same tensor from the forward on: the totals of the lag statistics are written into it by jobs that the head
                     # launch carries (csrc/first_fin.h), or that the block's next forward / its backward flush
_FIN_KEEP = []       # BatchNorm vectors + lag statistics of the first conv block: its backward's finish may be parked until the
                     # optimizer launch (csrc/first_fin.h) and reads them there; released when the next backward parks its own


def rider_flush():
    """Issue a parked rider launch on its own if no carrier took it (csrc/rider.h) and release its tensors."""
    check(_lib.lib().emb_rider_flush(stream()), "emb_rider_flush")
    _RIDER_KEEP.clear()


def _mlp_launch(x, rng, T, meta, params, park=False):
    """Allocate the outputs of a fused MLP stack and launch it; park=True leaves the launch to the next carrier kernel of
    the stream (the caller flushes).  Returns (x in T, weights in T, activations, masks, widths)."""
    P = PARAM_DTYPE[T]
    L_ = len(meta)
    B, Fin = x.shape
    dev = x.device
    xc = _as(x, T)
    Ws = [weight_as(params[2 * l], T) for l in range(L_)]
    bs = [_as(params[2 * l + 1].detach(), P) for l in range(L_)]
    Ns = [w.shape[0] for w in Ws]
    hs = [torch.empty(B, n, dtype=T, device=dev) for n in Ns]
    masks = [torch.empty(B, n, dtype=torch.uint8, device=dev) if (m[0] or m[1] > 0) else None for n, m in zip(Ns, meta)]
    iN = (_ct.c_int * L_)(*Ns)
    irelu = (_ct.c_int * L_)(*[int(bool(m[0])) for m in meta])
    fdrop = (_ct.c_float * L_)(*[float(m[1]) for m in meta])
    ilid = (_ct.c_int * L_)(*[int(m[2]) for m in meta])
    L = _lib.lib()
    if park:
        check(L.emb_rider_defer(stream(), 1), "emb_rider_defer")
    try:
        check(L.emb_mlp_fwd(ptr(xc), _parr(Ws), _parr(bs), _parr(hs), _parr(masks), iN, irelu, fdrop, ilid, L_, B, Fin,
                            rng.seed, rng.step_val, ptr(rng.step_dev), rng.row0, DTYPE_CODE[T], stream()), "emb_mlp_fwd")
    finally:
        if park:
            check(L.emb_rider_defer(stream(), 0), "emb_rider_defer")
    if park:
        _RIDER_KEEP.extend([xc, *Ws, *bs, *hs, *[m for m in masks if m is not None]])
    return xc, Ws, hs, masks, Ns


class _MlpFn(torch.autograd.Function):
    """Whole Linear(+ReLU+Dropout) stack in one forward / two backward launches (csrc/mlp.hip).
    args: x, rng, T, meta (tuple of (relu, drop_p, layer_id) per layer), then W_0, b_0, W_1, b_1, ..."""

    @staticmethod
    def forward(ctx, x, rng, T, meta, pre, *params):
        _lib.require_cuda(x)
        P = PARAM_DTYPE[T]
        L_ = len(meta)
        ctx.ride = pre is not None             # a prelaunched stack is an independent chain: its backward may ride as well
        if pre is None:
            pre = _mlp_launch(x, rng, T, meta, params)
        else:                                  # launched (or parked as a rider) by mlp_prelaunch: make sure it has been issued
            rider_flush()
        xc, Ws, hs, masks, Ns = pre
        ctx.save_for_backward(xc, *Ws, *hs, *[m if m is not None else hs[0] for m in masks])
        ctx.ws_tag = f"mlp{int(meta[0][2])}"
        ctx.ws_owner = params[0]
        ctx.cfg = (T, L_, Ns, [bool(m[0]) for m in meta], [float(m[1]) for m in meta], [m is not None for m in masks],
                   x.dtype, [params[i].dtype for i in range(2 * L_)])
        ctx.sinks = tuple(grad_sink(q, P) for q in params)
        return hs[-1]

    @staticmethod
    def backward(ctx, dy):
        T, L_, Ns, relus, drops, has_mask, xdt, pdts = ctx.cfg
        P = PARAM_DTYPE[T]
        sv = ctx.saved_tensors
        xc, Ws, hs, mk = sv[0], sv[1:1 + L_], sv[1 + L_:1 + 2 * L_], sv[1 + 2 * L_:1 + 3 * L_]
        masks = [m if ok else None for m, ok in zip(mk, has_mask)]
        B, Fin = xc.shape
        dev = xc.device
        dy = _as(dy, T)
        dx = torch.empty(B, Fin, dtype=T, device=dev) if ctx.needs_input_grad[0] else None
        Ks = [Fin] + list(Ns[:-1])
        sk = ctx.sinks
        dWs = [_out(sk[2 * l], (Ns[l], Ks[l]), P, dev) for l in range(L_)]
        dbs = [_out(sk[2 * l + 1], (Ns[l],), P, dev) for l in range(L_)]
        iN = (_ct.c_int * L_)(*Ns)
        need = _lib.lib().emb_mlp_workspace_bytes(Fin, iN, L_, B, DTYPE_CODE[T])
        ws = _workspace(dev, max(need, 1 << 22), ctx.ws_tag, ctx.ws_owner)
        irelu = (_ct.c_int * L_)(*[int(r) for r in relus])
        fdrop = (_ct.c_float * L_)(*drops)
        # parked for the BatchNorm-backward pass of the conv stack (csrc/rider.h) -- unless the caller wants the input gradient:
        # autograd hands dx on (a cast, AccumulateGrad's copy) as soon as this node returns, i.e. before a parked launch has run
        ride = ctx.ride and dx is None
        if ride:
            check(_lib.lib().emb_rider_defer(stream(), 1), "emb_rider_defer")
        try:
            check(_lib.lib().emb_mlp_bwd(ptr(xc), _parr(Ws), _parr(hs), _parr(masks), ptr(dy), ptr(dx), _parr(dWs), _parr(dbs), iN, irelu,
                                         fdrop, L_, B, Fin, ptr(ws), ws.numel(), DTYPE_CODE[T], stream()), "emb_mlp_bwd")
        finally:
            if ride:
                check(_lib.lib().emb_rider_defer(stream(), 0), "emb_rider_defer")
        if ride:
            _RIDER_KEEP.extend([xc, dy, *Ws, *hs, *[m for m in masks if m is not None]])
        cast = lambda g, d: None if g is None else (g if g.dtype == d else g.to(d))
        grads = []
        for l in range(L_):
            grads += [None if sk[2 * l] is not None else cast(dWs[l], pdts[2 * l]),
                      None if sk[2 * l + 1] is not None else cast(dbs[l], pdts[2 * l + 1])]
        return (cast(dx, xdt), None, None, None, None, *grads)


def mlp(x, layers, rng=None, compute_dtype=None):
    """layers: list of (weight, bias, relu, dropout_p, layer_id).  One fused launch when the stack fits the kernel
    (emb_mlp_supported), otherwise one tiled GEMM launch per layer."""
    T = compute_dtype or x.dtype
    rng = rng or RngState()
    Ns = [w.shape[0] for w, *_ in layers]
    ok = x.is_cuda and 1 <= len(layers) <= 4 and _lib.lib().emb_mlp_supported(
        x.shape[1], (_ct.c_int * len(Ns))(*Ns), len(Ns), DTYPE_CODE[T])
    if not ok:
        for w, b, relu, p, lid in layers:
            x = linear(x, w, b, relu=relu, dropout_p=p, layer_id=lid, rng=rng, compute_dtype=T)
        return x
    meta = tuple((bool(relu), float(p), int(lid)) for _, _, relu, p, lid in layers)
    params = [t for w, b, *_ in layers for t in (w, b)]
    return _MlpFn.apply(x, rng, T, meta, None, *params)


def mlp_prelaunch(x, layers, rng=None, compute_dtype=None):
    """First half of `mlp` for a stack whose result is not needed until other, independent kernels have been launched on the
    stream: the forward launch is PARKED and rides on the next carrier kernel (the statistics pass of a conv stack's first
    block, csrc/rider.h), so the two chains overlap on the GPU.  Returns a handle for `mlp_attach`, or None when the stack
    does not take the fused kernels (call `mlp` then).  Nothing may read the activations before `mlp_attach`."""
    T = compute_dtype or x.dtype
    rng = rng or RngState()
    Ns = [w.shape[0] for w, *_ in layers]
    ok = x.is_cuda and 1 <= len(layers) <= 4 and _lib.lib().emb_mlp_supported(
        x.shape[1], (_ct.c_int * len(Ns))(*Ns), len(Ns), DTYPE_CODE[T])
    if not ok:
        return None
    meta = tuple((bool(relu), float(p), int(lid)) for _, _, relu, p, lid in layers)
    params = [t for w, b, *_ in layers for t in (w, b)]
    with torch.no_grad():
        pre = _mlp_launch(x, rng, T, meta, params, park=True)
    return (x, rng, T, meta, pre, params)


def mlp_attach(handle):
    """Second half of `mlp_prelaunch`: builds the autograd node (created AFTER the nodes of the kernels launched in between,
    so its backward runs before theirs and can ride on them in turn) and returns the stack's output."""
    x, rng, T, meta, pre, params = handle
    return _MlpFn.apply(x, rng, T, meta, pre, *params)


class _WeightedCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, class_counts, global_counts, confusion, loss_out):
        _lib.require_cuda(logits, target)
        T = logits.dtype
        z = logits if logits.is_contiguous() else logits.contiguous()
        tgt = _as(target.reshape(-1), torch.int64)
        B = z.shape[0]
        if z.shape[1] != 2:
            raise NotImplementedError("weighted CE kernel implements the reference's 2-class task")
        loss = loss_out if loss_out is not None else torch.empty(1, dtype=torch.float32, device=z.device)
        dz = torch.empty_like(z)
        check(_lib.lib().emb_weighted_ce(ptr(z), ptr(tgt), ptr(class_counts), int(bool(global_counts)), ptr(loss), ptr(dz),
                                         ptr(confusion), None, None, B, DTYPE_CODE[T], stream()), "emb_weighted_ce")
        ctx.save_for_backward(dz)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        (dz,) = ctx.saved_tensors
        return dz * g.to(dz.dtype), None, None, None, None, None


def weighted_ce(logits, target, class_counts=None, global_counts=False, confusion=None, loss_out=None):
    """Per-batch class-weighted 2-class cross-entropy of the reference's train/eval step
    (utils/utils.py:121-140, training_models_multimodal.py:140-141,151-154), loss in fp32.
    class_counts: int64[2] device tensor; written (pos, n) unless global_counts (then read).
    confusion: optional int64[4] device slot, written with (TP, predicted-positive, positive, n) of this batch.
    loss_out: optional fp32[1] device slot that receives the loss (e.g. a row of metrics.StepTable)."""
    if class_counts is None:
        class_counts = torch.empty(2, dtype=torch.int64, device=logits.device)
    return _WeightedCEFn.apply(logits, target, class_counts, global_counts, confusion, loss_out)


def weighted_ce_with_grad(logits, target, class_counts=None, global_counts=False, confusion=None, loss_out=None, ticks=()):
    """Same kernel as `weighted_ce`, returning (loss [fp32 scalar, detached], dlogits) so that a trainer can call
    ``logits.backward(dlogits)`` directly: the loss is the root of the graph and its own gradient is 1.
    ticks: up to two int64 device counters the launch increments by one (the model's deferred RNG step, the
    optimizer's step) -- see emb_weighted_ce."""
    _lib.require_cuda(logits, target)
    z = logits.detach()
    z = z if z.is_contiguous() else z.contiguous()
    tgt = _as(target.reshape(-1), torch.int64)
    if z.shape[1] != 2:
        raise NotImplementedError("weighted CE kernel implements the reference's 2-class task")
    if class_counts is None:
        class_counts = torch.empty(2, dtype=torch.int64, device=z.device)
    loss = loss_out if loss_out is not None else torch.empty(1, dtype=torch.float32, device=z.device)
    dz = torch.empty_like(z)
    ticks = [t for t in ticks if t is not None]
    if len(ticks) > 2:
        raise ValueError("at most two tick counters")
    ta, tb = (list(ticks) + [None, None])[:2]
    check(_lib.lib().emb_weighted_ce(ptr(z), ptr(tgt), ptr(class_counts), int(bool(global_counts)), ptr(loss), ptr(dz),
                                     ptr(confusion), ptr(ta), ptr(tb), z.shape[0], DTYPE_CODE[z.dtype], stream()), "emb_weighted_ce")
    return loss.view(()), dz


class FusedLoss:
    """What the classifier head needs to take the loss (and its own backward) into its launch -- handed to the model for ONE
    forward by a trainer (``EmbraceNetMultimodal.arm_fused_loss``): labels of the local rows, the class-count tensor
    (filled by the kernel, or given when the counts are global), where to put the loss and the confusion counts, and the
    step counters to advance (see weighted_ce_with_grad)."""
    __slots__ = ("target", "class_counts", "global_counts", "loss_out", "confusion", "ticks")

    def __init__(self, target, class_counts, global_counts, loss_out, confusion, ticks=()):
        # global_counts: False = count this batch's labels (class_counts int64[2] is filled), True = class_counts holds the
        # GLOBAL (positives, rows), 2 = class_counts is the float32[4] data-parallel exchange block of csrc/head.hip
        self.target, self.class_counts, self.global_counts = target, class_counts, int(global_counts)
        self.loss_out, self.confusion, self.ticks = loss_out, confusion, tuple(t for t in ticks if t is not None)


def head_ce_supported(B, K, T):
    return T in (torch.float32, torch.bfloat16) and bool(_lib.lib().emb_head_ce_supported(int(B), int(K), DTYPE_CODE[T]))


class _HeadCEFn(torch.autograd.Function):
    """logits = E W^T + b, the class-weighted CE on them, and the head's own backward, in one launch (csrc/head.hip).
    The incoming gradient of `backward` is IGNORED: by contract the loss armed for this forward is the root of the graph
    (trainers call ``logits.backward(<anything of the right shape>)``)."""

    @staticmethod
    def forward(ctx, E, W, b, T, arm, train, code=None):
        _lib.require_cuda(E, W, b, arm.target)
        _PREMASKED.clear()
        L_ = _lib.lib()
        B, K = E.shape
        dev = E.device
        Ec = _as(E, T)
        if W.dtype != torch.float32 or b.dtype != torch.float32:
            raise TypeError("fused head: fp32 master weights expected")
        Wc, bc = W.detach().contiguous(), b.detach().contiguous()
        tgt = _as(arm.target.reshape(-1), torch.int64)
        logits = torch.empty(B, 2, dtype=T, device=dev)
        dE = torch.empty(B, K, dtype=T, device=dev) if train else None
        ws = _workspace(dev, L_.emb_head_ce_workspace_bytes(B, K), "head", W)
        if len(arm.ticks) > 2:
            raise ValueError("at most two tick counters")
        ta, tb = (list(arm.ticks) + [None, None])[:2]
        # `code`: E is the fusion layer's output and these are its code bytes -- the head also writes the pre-masked gradients
        # its backward GEMMs multiply (emb_embrace_bwd_masked); dE itself is still written: the fallback stays exact
        masked = (train and code is not None and tuple(code.shape) == (B, K) and code.is_contiguous()
                  and L_.emb_embrace_bwd_masked_supported(B, 8, 8, K, DTYPE_CODE[T]))
        dD0 = torch.empty(B, K, dtype=T, device=dev) if masked else None
        dD1 = torch.empty(B, K, dtype=T, device=dev) if masked else None
        check(L_.emb_head_ce_masked(ptr(Ec), ptr(Wc), ptr(bc), ptr(tgt), ptr(arm.class_counts), int(arm.global_counts), ptr(logits),
                                    ptr(dE), ptr(code) if masked else None, ptr(dD0), ptr(dD1), ptr(ws), ws.numel(), ptr(ta),
                                    ptr(tb), B, K, DTYPE_CODE[T], stream()), "emb_head_ce_masked")
        ctx.premasked = (dD0, dD1, code.data_ptr(), getattr(code, "_emb_serial", None)) if masked else None
        if not train:                                            # evaluation: nothing else will come, finish now
            check(L_.emb_head_ce_finish(ptr(ws), None, None, ptr(arm.loss_out), ptr(arm.confusion), B, K, stream()), "emb_head_ce_finish")
        ctx.keep = (dE, ws, arm.loss_out, arm.confusion, B, K, E.dtype)
        ctx.sinks = (grad_sink(W, torch.float32), grad_sink(b, torch.float32))
        return logits

    @staticmethod
    def backward(ctx, _ignored):
        dE, ws, loss_out, confusion, B, K, edt = ctx.keep
        dev = ws.device
        dW = _out(ctx.sinks[0], (2, K), torch.float32, dev)
        db = _out(ctx.sinks[1], (2,), torch.float32, dev)
        check(_lib.lib().emb_head_ce_finish(ptr(ws), ptr(dW), ptr(db), ptr(loss_out), ptr(confusion), B, K, stream()), "emb_head_ce_finish")
        if ctx.premasked is not None and dE.dtype == edt:
            _PREMASKED[dE.data_ptr()] = ctx.premasked
        return (dE if dE.dtype == edt else dE.to(edt), None if ctx.sinks[0] is not None else dW,
                None if ctx.sinks[1] is not None else db, None, None, None, None)


def head_ce(E, W, b, arm, compute_dtype=None, code=None):
    """Final Linear(width, 2) + armed loss (FusedLoss) -> logits; see _HeadCEFn.  `code`: E is the output of `embrace` and these
    are its code bytes (the head then prepares the fusion layer's backward, emb_head_ce_masked)."""
    # (needs_input_grad inside the node does not see torch.no_grad(): decide here whether a backward will follow)
    train = torch.is_grad_enabled() and (E.requires_grad or W.requires_grad or b.requires_grad)
    return _HeadCEFn.apply(E, W, b, compute_dtype or E.dtype, arm, train, code if (train and E.requires_grad) else None)


def count_labels(target, out=None):
    tgt = _as(target.reshape(-1), torch.int64)
    _lib.require_cuda(tgt)
    out = out if out is not None else torch.empty(2, dtype=torch.int64, device=tgt.device)
    check(_lib.lib().emb_count_labels(ptr(tgt), ptr(out), tgt.numel(), stream()), "emb_count_labels")
    return out


def cast(src, dtype, out=None):
    src = src if src.is_contiguous() else src.contiguous()
    out = out if out is not None else torch.empty(src.shape, dtype=dtype, device=src.device)
    check(_lib.lib().emb_cast(ptr(src), DTYPE_CODE[src.dtype], ptr(out), DTYPE_CODE[dtype], src.numel(), stream()), "emb_cast")
    return out


# ------------------------------------------------------------------------------------------------------
# sequence pre-network (CNN_pre.py:24-76): the whole Conv1d->BN->ReLU->MaxPool(->Dropout) stack as ONE
# autograd node; activations are channels-last between blocks, the last block emits the reference's
# [B, C*Lp] flatten order.
_WORKSPACE = {}
_RETIRED_WORKSPACES = []          # outgrown scratch buffers, kept alive for graphs captured on them (see _workspace)


def reduce_defer(enable):
    """Queue the weight-gradient slab reductions of the backward kernels instead of launching one per layer; `reduce_flush()`
    then runs them all in ONE launch.  Between the two the parameter gradients are incomplete (training.StepRunner and
    bench.py bracket `backward()` with them).  See include/embrace_hip.h."""
    check(_lib.lib().emb_reduce_defer(stream(), int(bool(enable))), "emb_reduce_defer")


def reduce_flush():
    check(_lib.lib().emb_reduce_flush(stream()), "emb_reduce_flush")


def park_copy(src, dst):
    """dst[:] = src (1 .. 64 fp32 values) by the NEXT fused optimizer launch of the stream instead of a launch of its own
    (emb_copy_park); `flush_copy()` runs it if no optimizer launch follows."""
    if src.dtype != torch.float32 or dst.dtype != torch.float32 or src.numel() != dst.numel() or not (src.is_contiguous() and dst.is_contiguous()):
        raise ValueError("park_copy: contiguous fp32 tensors of equal size")
    check(_lib.lib().emb_copy_park(stream(), ptr(src), ptr(dst), src.numel()), "emb_copy_park")


def flush_copy():
    check(_lib.lib().emb_copy_flush(stream()), "emb_copy_flush")


def parked_count(all_streams=False):
    """Launch descriptors the library holds for the current stream (or all streams): queued slab reductions, a parked rider, the
    first conv block's parked finish / totals jobs.  0 after a completed step."""
    return int(_lib.lib().emb_parked_count(stream(), int(bool(all_streams))))


def reset(all_streams=False):
    """Drop everything parked on the current stream (or on every stream) WITHOUT launching it, switch deferral off and release
    the tensors kept alive for parked launches: the recovery call after a step raised between a deferring call and its flush
    (training.StepRunner does it).  Returns the number of descriptors dropped."""
    L = _lib.lib()
    n = int(L.emb_reset() if all_streams else L.emb_reset_stream(stream()))
    _RIDER_KEEP.clear()
    return n


_WS_FINALIZERS = set()


def _workspace_drop(owner_id):
    _WS_FINALIZERS.discard(owner_id)
    for k in [k for k in _WORKSPACE if k[2] == owner_id]:
        del _WORKSPACE[k]


def _workspace(device, nbytes, tag="", owner=None):
    """Scratch buffer of call site `tag` of the layer that owns the parameter object `owner`: kernels of different streams may run
    concurrently, and a deferred slab reduction still reads one call site's buffer after the next backward kernels have run --
    every (layer, call site) keeps its own, so two models trained side by side in one process never share scratch (and keeps it
    across steps: nothing is allocated under capture)."""
    key = (torch.device(device), tag, 0 if owner is None else id(owner))
    if owner is not None and id(owner) not in _WS_FINALIZERS:     # the scratch dies with the parameter object it belongs to
        _WS_FINALIZERS.add(id(owner))
        weakref.finalize(owner, _workspace_drop, id(owner))
    buf = _WORKSPACE.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            # A captured step graph (training.StepRunner) may have baked this buffer's address in and will keep writing
            # partial sums there on every replay: a retired buffer is never handed back to the allocator.  Growth is
            # geometric, so the retired total stays below the size of the live buffer.
            _RETIRED_WORKSPACES.append(buf)
            nbytes = max(int(nbytes), (3 * buf.numel()) // 2)
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _WORKSPACE[key] = buf
    return buf


def _vec(dtype):
    return 16 // torch.empty(0, dtype=dtype).element_size()


def pool_out_len(L):
    return (L - 10) // 2 + 1


class _ConvStackFn(torch.autograd.Function):
    """args: x [B,C,L], training, rng, T, meta (tuple of per-layer dicts: k, drop_p, momentum, eps, layer_id), bn_sync, then
    per layer: conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var.
    bn_sync: None, or a callable that SUM-all-reduces a float64 device vector in place over the data-parallel ranks: the
    BatchNorm statistics (forward) and the two BN-backward means then come from the GLOBAL batch (include/embrace_hip.h,
    bn_phase), one exchange of 2*Cout+1 doubles per block and direction."""

    @staticmethod
    def forward(ctx, x, training, rng, T, meta, bn_sync, *tensors):
        _lib.require_cuda(x)
        L_ = _lib.lib()
        P = PARAM_DTYPE[T]
        dev, code = x.device, DTYPE_CODE[T]
        codes_in = x.dtype == torch.uint8 and x.dim() == 2          # base codes [B][L] (pack_onehot), SURVEY 8 row f4
        if codes_in:
            (B, L), C0 = x.shape, 4
        else:
            B, C0, L = x.shape
        x = x if x.is_contiguous() else x.contiguous()
        vec = _vec(T)
        cin_pad = -(-C0 // vec) * vec
        x_codes = 0
        if codes_in:
            Cout0, _, k0 = tensors[0].shape
            if L_.emb_convblock_needs_y(B, L, cin_pad, Cout0, k0, code) == 0:
                cur, x_codes = x, 1                                  # the fused first block expands the codes while staging
            else:                                                    # other configurations: expand here, channels-last
                cur = torch.zeros(B, L, cin_pad, dtype=T, device=dev)
                cur[:, :, :4] = (x.unsqueeze(-1) == torch.arange(4, device=dev, dtype=torch.uint8)).to(T)
        else:
            cur = torch.empty(B, L, cin_pad, dtype=T, device=dev)
            Cout0, _, k0 = tensors[0].shape
            # training step on the loader's [B, 4, L] windows in the compute dtype: the first block's statistics pass stages
            # straight from that layout and writes the channels-last image itself (x_codes = 2) -- no conversion launch
            ncl_direct = (training and bn_sync is None and T == torch.bfloat16 and x.dtype == T and C0 == 4 and cin_pad == 8
                          and L_.emb_convblock_needs_y(B, L, cin_pad, Cout0, k0, code) == 0)
            if ncl_direct:
                x_codes = 2
            else:
                check(L_.emb_ncl_to_nlc(ptr(x), DTYPE_CODE[x.dtype], ptr(cur), code, B, C0, L, cin_pad, stream()), "emb_ncl_to_nlc")
        saved, shapes = [], []
        n_layers = len(meta)
        for i, m in enumerate(meta):
            w, b, g, beta, rmean, rvar = tensors[6 * i:6 * i + 6]
            nbt = m.get("num_batches_tracked")
            for t in (w, b, g, beta, rmean, rvar):
                if t.dtype != P:
                    raise TypeError(f"conv stack parameters must be {P} for compute dtype {T}")
            Cout, Cin, k = w.shape
            wpack, wflip = conv_packed(w, T, cin_pad, need_flip=i > 0)
            Lp = pool_out_len(L)
            last = i == n_layers - 1
            # first block with a few-channel input: recomputed in every pass instead of stored (csrc/conv_first.hip)
            fused = i == 0 and L_.emb_convblock_needs_y(B, L, cin_pad, Cout, k, code) == 0
            y = None if fused else torch.empty(B, L, Cout, dtype=T, device=dev)
            stats = torch.empty(L_.emb_convblock_stats_elems(B, L, cin_pad, Cout, k, code), dtype=P, device=dev)   # (+ lag statistics, first block)
            out = torch.empty((B, Cout, Lp) if last else (B, Lp, Cout), dtype=T, device=dev)
            argmax = torch.empty(B, Lp, Cout, dtype=torch.uint8, device=dev)
            nbytes = L_.emb_convblock_workspace_bytes(B, L, cin_pad, Cout, k, code)
            ws = _workspace(dev, nbytes, f"conv{i}", w)
            sync = bn_sync if training else None
            sums = torch.empty(2 * Cout + 1, dtype=torch.float64, device=dev) if sync is not None else None
            for phase in ((1, 2) if sync is not None else (0,)):
                ncl_in = i == 0 and x_codes == 2                # loader layout in, channels-last image out through `y`
                check(L_.emb_convblock_fwd(ptr(x if ncl_in else cur), ptr(wpack), ptr(b.detach()), ptr(g.detach()),
                                           ptr(beta.detach()), ptr(rmean),
                                           ptr(rvar), int(training), float(m["momentum"]), float(m["eps"]), float(m["drop_p"]),
                                           rng.seed, rng.step_val, ptr(rng.step_dev), rng.row0, int(m["layer_id"]),
                                           ptr(cur) if ncl_in else ptr(y),
                                           ptr(stats), ptr(out), ptr(argmax), int(last), ptr(ws), ws.numel(), ptr(nbt),
                                           x_codes if i == 0 else 0, phase, ptr(sums), B, L, cin_pad, Cout, k, code, stream()),
                      "emb_convblock_fwd")
                if phase == 1:
                    sync(sums)                                       # {sum y, sum y^2, rows} of the shard -> of the global batch
            if fused:
                _JOBS_KEEP[:] = [stats]
            saved += [cur, y if y is not None else stats, stats, argmax, wflip if wflip is not None else stats, wpack, b.detach()]
            shapes.append((L, Cin, cin_pad, Cout, k, float(m["drop_p"]), fused))
            cur, L, cin_pad = out, Lp, Cout
        ctx.save_for_backward(*saved)
        ctx.cfg = (T, int(training), B, shapes, 0 if x_codes == 2 else x_codes, bn_sync if training else None)   # (backward reads the saved channels-last image)
        ctx.sinks = tuple(grad_sink(tensors[6 * i + j], P) for i in range(n_layers) for j in range(4))
        ctx.ws_owners = tuple(tensors[6 * i] for i in range(n_layers))
        if _RIDER_KEEP:
            rider_flush()                      # a parked MLP forward that no kernel of this stack carried
        return cur.reshape(B, -1)

    @staticmethod
    def backward(ctx, dout):
        T, training, B, shapes, x_codes, sync = ctx.cfg
        P = PARAM_DTYPE[T]
        L_ = _lib.lib()
        code = DTYPE_CODE[T]
        saved = ctx.saved_tensors
        dev = dout.device
        g = _as(dout, T)
        grads = [None] * (6 * len(shapes))
        for i in reversed(range(len(shapes))):
            xin, y, stats, argmax, wflip, wpack, bias = saved[7 * i:7 * i + 7]
            L, Cin, cin_pad, Cout, k, drop_p, fused = shapes[i]
            last = i == len(shapes) - 1
            dy = None if fused else torch.empty(B, L, Cout, dtype=T, device=dev)
            dx = torch.empty(B, L, cin_pad, dtype=T, device=dev) if i > 0 else None
            sk = ctx.sinks[4 * i:4 * i + 4]
            dW = _out(sk[0], (Cout, Cin, k), P, dev)
            db, dgam, dbeta = (_out(sk[j], (Cout,), P, dev) for j in (1, 2, 3))
            nbytes = L_.emb_convblock_workspace_bytes(B, L, cin_pad, Cout, k, code)
            ws = _workspace(dev, nbytes, f"conv{i}", ctx.ws_owners[i])
            sums = torch.empty(2 * Cout + 1, dtype=torch.float64, device=dev) if sync is not None else None
            for phase in ((1, 2) if sync is not None else (0,)):
                check(L_.emb_convblock_bwd(ptr(g), int(last), ptr(argmax), None if fused else ptr(y), ptr(stats), ptr(xin),
                                           ptr(wflip) if i > 0 else None, ptr(wpack), ptr(bias), drop_p, training, ptr(dx),
                                           ptr(dW), ptr(db), ptr(dgam), ptr(dbeta), ptr(dy), ptr(ws), ws.numel(),
                                           x_codes if i == 0 else 0, phase, ptr(sums), B, L, Cin, cin_pad, Cout, k, code, stream()),
                      "emb_convblock_bwd")
                if phase == 1:
                    sync(sums)                                       # {sum dz, sum dz*xhat, rows} -> of the global batch
            grads[6 * i:6 * i + 4] = [None if sk[j] is not None else g_ for j, g_ in enumerate((dW, db, dgam, dbeta))]
            if fused:
                _FIN_KEEP[:] = [stats]
            g = dx
        if _RIDER_KEEP:
            rider_flush()                      # a parked MLP backward that no kernel of this stack carried
        return (None, None, None, None, None, None, *grads)


class RowGather:
    """Batch staging: out[t][i] = tables[t][idx[i]] for up to four device tables sharing one int64 device
    index vector, in ONE launch (csrc/gather.hip).  Tables are [N, ...] contiguous tensors with the same N; they are
    validated once here, a call only allocates the outputs and launches."""

    def __init__(self, tables):
        import ctypes
        tables = tuple(tables)
        if not 1 <= len(tables) <= 4:
            raise ValueError("gather_rows: 1 to 4 tables per call")
        _lib.require_cuda(*tables)
        self.n_rows = tables[0].shape[0]
        for t in tables:
            if t.dim() < 1 or t.shape[0] != self.n_rows or not t.is_contiguous() or t.device != tables[0].device:
                raise ValueError("gather_rows: tables must be contiguous, on one device, with the same number of rows")
        if self.n_rows == 0:
            raise ValueError("gather_rows: empty tables")
        self.tables, k = tables, len(tables)
        self._src = (ctypes.c_void_p * k)(*[t.data_ptr() for t in tables])
        self._rb = (ctypes.c_int64 * k)(*[(t.numel() // self.n_rows) * t.element_size() for t in tables])
        self._dst = (ctypes.c_void_p * k)()
        self._meta = [(tuple(t.shape[1:]), t.dtype) for t in tables]
        self._fn, self._dev = _lib.lib().emb_gather_rows, tables[0].device

    def __call__(self, idx, out=None):
        """`out`: tensors of a previous call with the same number of indices, to be overwritten (staging buffers)."""
        if idx.dtype != torch.int64 or idx.dim() != 1 or not idx.is_contiguous() or idx.device != self._dev:
            raise TypeError("gather_rows: idx must be a contiguous 1-D int64 tensor on the tables' device")
        n = idx.shape[0]
        outs = out if out is not None else [torch.empty((n,) + shape, dtype=dt, device=self._dev) for shape, dt in self._meta]
        if out is not None and (len(outs) != len(self._meta) or any(o.shape[0] != n for o in outs)):
            raise ValueError("gather_rows: `out` does not fit this call")
        if n:
            for i, o in enumerate(outs):
                self._dst[i] = o.data_ptr()
            check(self._fn(self._src, self._dst, self._rb, len(outs), idx.data_ptr(), n, self.n_rows, stream()), "emb_gather_rows")
        return outs


def gather_rows(tables, idx):
    """One-shot form of RowGather."""
    _lib.require_cuda(idx)
    return RowGather(tables)(idx)


def pack_onehot(x):
    """[N, 4, L] one-hot windows (the loader's format, dataprepare.py:398-412) -> [N, L] uint8 base codes: the hot channel,
    4 for an all-zero column.  Done once per data set (host or device); batches then travel and are staged as one byte per
    position -- `conv_stack` / `CNN_pre` accept the codes directly."""
    if x.dim() != 3 or x.shape[1] != 4:
        raise ValueError("pack_onehot expects [N, 4, L] one-hot windows")
    hot = x != 0
    if bool((hot.sum(1) > 1).any()) or bool(((x != 0) & (x != 1)).any()):
        raise ValueError("pack_onehot: input is not one-hot (values other than 0/1 or several channels set)")
    codes = hot.to(torch.uint8).argmax(1).to(torch.uint8)
    return torch.where(hot.any(1), codes, torch.full_like(codes, 4)).contiguous()


def conv_stack(x, layers, training, rng=None, compute_dtype=None, bn_sync=None):
    """layers: list of dicts with conv (nn.Conv1d), bn (nn.BatchNorm1d), drop_p, layer_id.
    x: [B, C, L] float windows, or [B, L] uint8 base codes from `pack_onehot`.
    bn_sync: see _ConvStackFn (BatchNorm statistics of the global batch under data parallelism)."""
    T = compute_dtype or layers[0]["conv"].weight.dtype
    meta, tensors = [], []
    for ly in layers:
        conv, bn = ly["conv"], ly["bn"]
        meta.append(dict(k=conv.kernel_size[0], drop_p=ly["drop_p"] if training else 0.0,
                         momentum=0.1 if bn.momentum is None else bn.momentum, eps=bn.eps, layer_id=ly["layer_id"],
                         num_batches_tracked=bn.num_batches_tracked if (training and bn.track_running_stats) else None))
        tensors += [conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var]
    return _ConvStackFn.apply(x, bool(training), rng or RngState(), T, tuple(meta), bn_sync, *tensors)
