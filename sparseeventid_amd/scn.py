"""Drop-in mirror of the ``sparseconvnet`` (scn) nn.Module surface used by the reference network
(/root/reference/src/networks/resnet.py:26-50,105-110,124-125;
 /root/reference/src/networks/sparse_building_blocks.py:20,29-34,39,76-82,110-117,122),
backed by the hand-written gfx950 engine through the C ABI (include/sparseeventid_amd.h).

Same constructor keywords, same parameter names / shapes (``weight [K,nIn,nOut]``, ``bias [nOut]``,
BN ``weight, bias, running_mean, running_var``), same intermediate object attributes
(``.features``, ``.metadata``, ``.spatial_size``, ``get_spatial_locations()``).

PyTorch is used for device memory, autograd bookkeeping and the stream; every sparse op runs in
the engine.  There is no CPU fallback.
"""
import ctypes
import math
import weakref

import torch
from torch import nn

from . import _lib
from ._lib import check, i32x3, lib


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


# ---- dW on the library's side stream (sei_set_backward_streams): 0 back to back, 1 joined per call, 2 deferred join
_BWD_STREAMS = 0
_HELD = []
_DW_PTRS = set()


def set_backward_streams(mode):
    """0: dW then dX on the caller's stream (default).  1: dW on the library's side stream next to dX, joined before the
    call returns.  2: dW stays queued on the side stream; the caller joins ONCE per backward pass (``backward_join()``,
    ``Trainer.train_step`` does) -- the backlog of dW kernels then also runs beside the HBM-bound BatchNorm-backward
    kernels of the dX chain.  In mode 2 the operands of every dW (X, dY, dW) are kept allocated until the join: the
    whole backward pass's dY tensors and saved activations stay live to its end (DUNE batch: 38.1 GB peak allocated /
    43.8 GB reserved against 36.9 / 38.0 GB, flat over 30 steps).  (``Tensor.record_stream`` on the side stream would
    free them earlier, but the caching allocator then cannot reuse a block until the backlog has passed it and grows
    instead: 210 GB reserved after 13 steps, measured.)"""
    global _BWD_STREAMS
    if _HELD:
        torch.cuda.synchronize()
        _HELD.clear()
    _DW_PTRS.clear()
    check(lib().sei_set_backward_streams(int(mode)))
    _BWD_STREAMS = int(mode)


PRODUCT_MODES = {"fp32": 0, "bf16x3": 1}


def set_product_mode(mode):
    """How the forward / dX convolutions multiply (C ABI ``sei_set_product_mode``).  "fp32" (default): fp32 matrix
    instructions, exact fp32 products.  "bf16x3" (OPT-IN): every fp32 operand as an exact sum of three bf16 numbers,
    the six largest of the nine part products on the bf16 matrix pipe with fp32 accumulation -- per product within 2^-23
    of the exact one (the size of one fp32 rounding), 0.375 of the matrix-pipe cycles; applies where the sixteen-wave
    round kernels run (64 .. 192 planes on large levels).  dW and every other operation stay as they are."""
    check(lib().sei_set_product_mode(PRODUCT_MODES[mode] if isinstance(mode, str) else int(mode)))


def _pin_to_side(x, dy, dw):
    """mode 2: X and dY are still being read on the side stream when the autograd node returns: held until the join.
    dW is NOT held: autograd's AccumulateGrad takes over a gradient tensor as ``param.grad`` only while nobody else
    references it, and otherwise CLONES it on the spot -- on the caller's stream, before the side stream has written it.
    Its address is remembered instead, so that the join can verify the gradients autograd kept are the buffers the
    kernels wrote (``backward_join(params)``)."""
    if _BWD_STREAMS == 2:
        _HELD.append((x, dy))
        _DW_PTRS.add(dw.data_ptr())


def backward_join(params=None):
    """The current stream waits for every dW queued on its side stream; the operands held for them are released (their
    memory is reused by work queued on this stream AFTER the wait).  ``params``: the parameters of the model -- every
    dW buffer queued since the last join must BE the ``.grad`` of one of them (autograd took it over instead of copying
    it before it was written), else RuntimeError.  No-op outside mode 2."""
    if _BWD_STREAMS == 2:
        check(lib().sei_backward_join(_stream()))
        _HELD.clear()
        queued = set(_DW_PTRS)
        _DW_PTRS.clear()
        if params is not None and queued:
            kept = {p.grad.data_ptr() for p in params if p.grad is not None}
            if not queued <= kept:
                raise RuntimeError("set_backward_streams(2): %d weight gradients were copied by autograd before their kernels "
                                   "ran (accumulation into an existing .grad, or a hook holding the tensor): use mode 1, or "
                                   "clear .grad before backward()" % len(queued - kept))



def _triple(v):
    if isinstance(v, torch.Tensor):
        v = v.tolist()
    if isinstance(v, int):
        return [v, v, v]
    v = [int(x) for x in v]
    if len(v) != 3:
        raise ValueError("expected 3 spatial dimensions, got %r" % (v,))
    return v


def _f32c(t):
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _require_cuda(t, what):
    if not t.is_cuda:
        raise RuntimeError("sparseeventid_amd.%s: tensors must live on the MI355X (cuda) device; "
                           "this engine has no CPU path" % what)


class Metadata:
    """scn.Metadata: owns the device hash maps and rulebooks of one batch (all levels)."""

    def __init__(self):
        h = ctypes.c_void_p()
        check(lib().sei_metadata_create(ctypes.byref(h)))
        self._h = h
        # python-side cache of Rulebook wrappers.  WEAK values: a Rulebook keeps its Metadata alive (autograd
        # contexts hold Rulebooks), so a strong cache would be a reference cycle and the device memory of every
        # batch would wait for the cyclic garbage collector; the C side caches the rulebooks themselves.
        self._rulebooks = weakref.WeakValueDictionary()

    def __del__(self):
        h = getattr(self, "_h", None)
        L = getattr(_lib, "_lib", None) if _lib is not None else None   # module globals are None at interpreter exit
        if h is not None and L is not None:
            L.sei_metadata_destroy(h)
            self._h = None

    @property
    def handle(self):
        return self._h

    def prebuild_like(self, other, forward_only=False):
        """Build, for THIS batch, every rulebook (and tiled variant) the metadata of an earlier batch holds
        (``sei_metadata_prebuild_like``) on the current stream.  A training loop calls it on a side stream for the next
        batch while the current step computes; the forward pass then finds its rulebooks built.  ``forward_only``: only
        the variants a forward convolution of ``other`` walked (``sei_metadata_prebuild_forward_like``) -- a pass that
        will not differentiate needs neither the transposed rules nor dW's variants."""
        if other is None or other is self or other._h is None:
            return
        fn = lib().sei_metadata_prebuild_forward_like if forward_only else lib().sei_metadata_prebuild_like
        check(fn(self._h, other._h, _stream()))

    def rows(self, spatial):
        n = ctypes.c_int64()
        check(lib().sei_level_rows(self._h, i32x3(spatial), ctypes.byref(n)))
        return n.value

    def batch_size(self):
        b = ctypes.c_int32()
        check(lib().sei_batch_size(self._h, ctypes.byref(b)))
        return b.value

    def coords(self, spatial, device):
        n = self.rows(spatial)
        out = torch.empty((n, 4), dtype=torch.int32, device=device)
        check(lib().sei_level_coords(self._h, i32x3(spatial), _ptr(out), _stream()))
        return out

    def subm_rulebook(self, spatial, filter_size):
        key = ("s", tuple(spatial), tuple(filter_size))
        rb = self._rulebooks.get(key)
        if rb is None:
            h = ctypes.c_void_p()
            check(lib().sei_subm_rulebook(self._h, i32x3(spatial), i32x3(filter_size), _stream(), ctypes.byref(h)))
            rb = Rulebook(self, h, list(spatial))
            self._rulebooks[key] = rb
        return rb

    def conv_rulebook(self, spatial, size, stride):
        key = ("c", tuple(spatial), tuple(size), tuple(stride))
        rb = self._rulebooks.get(key)
        if rb is None:
            h = ctypes.c_void_p()
            out_sp = (ctypes.c_int32 * 3)()
            check(lib().sei_conv_rulebook(self._h, i32x3(spatial), i32x3(size), i32x3(stride), _stream(),
                                          ctypes.byref(h), out_sp))
            rb = Rulebook(self, h, [int(x) for x in out_sp])
            self._rulebooks[key] = rb
        return rb


class Rulebook:
    """Borrowed handle of one rulebook; keeps its Metadata alive (autograd ctx holds this)."""

    def __init__(self, metadata, handle, spatial_out):
        self.metadata = metadata
        self.h = handle
        self.spatial_out = spatial_out
        info = (ctypes.c_int64 * 8)()
        check(lib().sei_rulebook_info(handle, info))
        self.n_dst, self.n_src, self.K = int(info[0]), int(info[1]), int(info[2])

    def stats(self, tile_rows=None):
        """(n_rules, n_sub, n_tiles, tile_rows) of a built tile size; builds `tile_rows` (default 128) if none is."""
        info = (ctypes.c_int64 * 8)()
        check(lib().sei_rulebook_info(self.h, info))
        if info[7] == 0 or (tile_rows is not None and info[7] != tile_rows):
            check(lib().sei_rulebook_build(self.h, tile_rows or 128, _stream()))
            check(lib().sei_rulebook_info(self.h, info))
        return int(info[3]), int(info[4]), int(info[5]), int(info[7])

    @property
    def n_rules(self):
        return self.stats()[0]

    def export(self, device):
        start = (ctypes.c_int64 * (self.K + 1))()
        rin = torch.empty(max(self.n_rules, 1), dtype=torch.int32, device=device)
        rout = torch.empty(max(self.n_rules, 1), dtype=torch.int32, device=device)
        check(lib().sei_rulebook_export(self.h, start, _ptr(rin), _ptr(rout), _stream()))
        return list(start), rin[:self.n_rules], rout[:self.n_rules]


class SparseConvNetTensor:
    """scn.SparseConvNetTensor: features [N,C] fp32 + metadata + spatial_size."""

    def __init__(self, features=None, metadata=None, spatial_size=None):
        self.features = features
        self.metadata = metadata
        self.spatial_size = spatial_size

    def get_spatial_locations(self, spatial_size=None):
        sp = _triple(self.spatial_size if spatial_size is None else spatial_size)
        return self.metadata.coords(sp, self.features.device).long().cpu()

    def batch_size(self):
        return self.metadata.batch_size()

    def cuda(self):
        self.features = self.features.cuda()
        return self

    def type(self, t=None):
        if t is not None:
            self.features = self.features.type(t)
            return self
        return self.features.type()

    def __repr__(self):
        return "SparseConvNetTensor<<features=%r, spatial_size=%r>>" % (
            tuple(self.features.shape), _triple(self.spatial_size))


# ============================================================================= autograd functions
class _InputFeatures(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, metadata, n_rows):
        ctx.md = metadata
        ctx.n_in = features.shape[0]
        C = features.shape[1]
        out = torch.empty((n_rows, C), dtype=torch.float32, device=features.device)
        check(lib().sei_input_features_fwd(metadata.handle, _ptr(features), C, _ptr(out), _stream()))
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = _f32c(dout)
        C = dout.shape[1]
        din = torch.empty((ctx.n_in, C), dtype=torch.float32, device=dout.device)
        check(lib().sei_input_features_bwd(ctx.md.handle, _ptr(dout), C, _ptr(din), _stream()))
        return din, None, None


class _Conv(torch.autograd.Function):
    """Shared by SubmanifoldConvolution and Convolution: the rulebook decides which."""

    @staticmethod
    def forward(ctx, features, weight, bias, rb):
        nIn, nOut = weight.shape[1], weight.shape[2]
        if features.shape[1] != nIn:
            raise RuntimeError("convolution: input has %d planes, weight expects %d" % (features.shape[1], nIn))
        if features.shape[0] != rb.n_src:
            raise RuntimeError("convolution: %d feature rows but the metadata level has %d"
                               % (features.shape[0], rb.n_src))
        ctx.rb = rb
        ctx.has_bias = bias is not None
        ctx.save_for_backward(features, weight)
        out = torch.empty((rb.n_dst, nOut), dtype=torch.float32, device=features.device)
        check(lib().sei_conv_forward(rb.h, _ptr(features), _ptr(weight), _ptr(bias), _ptr(out), nIn, nOut, _stream()))
        return out

    @staticmethod
    def backward(ctx, dout):
        features, weight = ctx.saved_tensors
        rb = ctx.rb
        dout = _f32c(dout)
        nIn, nOut = weight.shape[1], weight.shape[2]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((rb.n_src, nIn), dtype=torch.float32, device=dout.device)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw = torch.empty_like(weight)
            if ctx.has_bias:
                db = torch.empty((nOut,), dtype=torch.float32, device=dout.device)
        if dx is not None or dw is not None:   # one call: the library overlaps dX and dW (dW on its own side stream)
            check(lib().sei_conv_backward(rb.h, _ptr(features), _ptr(dout), _ptr(weight), None, _ptr(dx), _ptr(dw), _ptr(db),
                                          nIn, nOut, _stream()))
            if dx is not None and dw is not None:
                _pin_to_side(features, dout, dw)
        return dx, dw, db, None


class _BatchNorm(torch.autograd.Function):
    """BN (+ optional residual add) + leaky activation in two passes over the rows."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, training, leak, residual):
        n, C = x.shape
        y = torch.empty_like(x)
        save_mean = torch.empty((C,), dtype=torch.float32, device=x.device)
        save_invstd = torch.empty((C,), dtype=torch.float32, device=x.device)
        check(lib().sei_bn_forward(_ptr(x), n, C, _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var),
                                   eps, momentum, 1 if training else 0, leak, _ptr(residual), _ptr(y),
                                   _ptr(save_mean), _ptr(save_invstd), _stream()))
        ctx.leak = leak
        ctx.training = training
        ctx.has_res = residual is not None
        ctx.has_affine = gamma is not None
        ctx.save_for_backward(x, y, gamma, save_mean, save_invstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, save_mean, save_invstd = ctx.saved_tensors
        dy = _f32c(dy)
        n, C = x.shape
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if ctx.has_res else None
        dgamma = torch.empty((C,), dtype=torch.float32, device=x.device) if ctx.has_affine else None
        dbeta = torch.empty((C,), dtype=torch.float32, device=x.device) if ctx.has_affine else None
        check(lib().sei_bn_backward(_ptr(x), _ptr(y), _ptr(dy), n, C, _ptr(gamma), _ptr(save_mean), _ptr(save_invstd),
                                    ctx.leak, 1 if ctx.training else 0, _ptr(dx), _ptr(dres), _ptr(dgamma),
                                    _ptr(dbeta), _stream()))
        return dx, dgamma, dbeta, None, None, None, None, None, None, dres


class _ResidualBlock(torch.autograd.Function):
    """conv1 -> BN -> act -> conv2 -> BN -> (+x) -> act as ONE autograd node
    (ResidualBlock, sparse_building_blocks.py:61-100; SparseResidualBlock, torch/sparseresnet.py:47-103).

    Same kernels as the separate modules; the point of the node is the backward: the gradient of the skip
    connection initialises the accumulators of conv1's dX kernel (dX = dres + conv^T(dY)) instead of a separate
    [N,C] add, and five autograd nodes become one."""

    @staticmethod
    def forward(ctx, x, rb, w1, b1, g1, be1, rm1, rv1, w2, b2, g2, be2, rm2, rv2, bn1cfg, bn2cfg, leak1, leak2):
        (eps1, mom1, training1), (eps2, mom2, training2) = bn1cfg, bn2cfg
        n, C = x.shape
        dev = x.device
        L = lib()
        st = _stream()
        # conv -> BN(+act) pairs through the fused entry point: the BN statistics come out of the conv epilogue
        y1, a1 = torch.empty_like(x), torch.empty_like(x)
        m1, s1 = torch.empty(C, device=dev), torch.empty(C, device=dev)
        check(L.sei_conv_bn_forward(rb.h, _ptr(x), _ptr(w1), _ptr(b1), _ptr(y1), C, C, _ptr(g1), _ptr(be1), _ptr(rm1),
                                    _ptr(rv1), eps1, mom1, 1 if training1 else 0, leak1, None, _ptr(a1), _ptr(m1),
                                    _ptr(s1), st))
        y2, out = torch.empty_like(x), torch.empty_like(x)
        m2, s2 = torch.empty(C, device=dev), torch.empty(C, device=dev)
        check(L.sei_conv_bn_forward(rb.h, _ptr(a1), _ptr(w2), _ptr(b2), _ptr(y2), C, C, _ptr(g2), _ptr(be2), _ptr(rm2),
                                    _ptr(rv2), eps2, mom2, 1 if training2 else 0, leak2, _ptr(x), _ptr(out), _ptr(m2),
                                    _ptr(s2), st))
        ctx.rb, ctx.training1, ctx.training2, ctx.leak1, ctx.leak2 = rb, training1, training2, leak1, leak2
        ctx.has_b1, ctx.has_b2 = b1 is not None, b2 is not None
        ctx.save_for_backward(x, y1, a1, y2, out, w1, w2, g1, g2, m1, s1, m2, s2, be1)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, y1, a1, y2, out, w1, w2, g1, g2, m1, s1, m2, s2, be1 = ctx.saved_tensors
        dout = _f32c(dout)
        n, C = x.shape
        dev = x.device
        L = lib()
        st = _stream()
        rb = ctx.rb
        tr1, tr2 = (1 if ctx.training1 else 0), (1 if ctx.training2 else 0)
        vec = lambda: torch.empty(C, dtype=torch.float32, device=dev)
        # BN2 (+ skip + act) backward: dy2, and the skip branch's gradient dres
        dy2, dres, dg2, dbe2 = torch.empty_like(x), torch.empty_like(x), vec(), vec()
        # the conv bias gradient = column sums of the BN backward's output, taken while it is written (no extra pass)
        db2 = vec() if ctx.has_b2 else None
        check(L.sei_bn_backward_colsum(_ptr(y2), _ptr(out), _ptr(dout), n, C, _ptr(g2), None, _ptr(m2), _ptr(s2), ctx.leak2,
                                       tr2, _ptr(dy2), _ptr(dres), _ptr(dg2), _ptr(dbe2), _ptr(db2), st))
        dw2, da1 = torch.empty_like(w2), torch.empty_like(x)
        # dW and dX of a layer in one call: the library overlaps them (dW on its own side stream)
        check(L.sei_conv_backward(rb.h, _ptr(a1), _ptr(dy2), _ptr(w2), None, _ptr(da1), _ptr(dw2), None, C, C, st))
        _pin_to_side(a1, dy2, dw2)
        del dy2
        dy1, dg1, dbe1 = torch.empty_like(x), vec(), vec()
        db1 = vec() if ctx.has_b1 else None
        # no residual behind BN1: the activation mask is recomputed from y1 (a1 is not read)
        check(L.sei_bn_backward_colsum(_ptr(y1), None, _ptr(da1), n, C, _ptr(g1), _ptr(be1), _ptr(m1), _ptr(s1), ctx.leak1,
                                       tr1, _ptr(dy1), None, _ptr(dg1), _ptr(dbe1), _ptr(db1), st))
        del da1
        dw1 = torch.empty_like(w1)
        # dX = dres + conv1^T(dy1): the skip gradient seeds the kernel's accumulators; dW beside it
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        check(L.sei_conv_backward(rb.h, _ptr(x), _ptr(dy1), _ptr(w1), _ptr(dres) if dx is not None else None, _ptr(dx),
                                  _ptr(dw1), None, C, C, st))
        if dx is not None:
            _pin_to_side(x, dy1, dw1)
        return (dx, None, dw1, db1, dg1, dbe1, None, None, dw2, db2, dg2, dbe2, None, None, None, None, None, None)


class _ConvBN(torch.autograd.Function):
    """conv -> BatchNorm (+residual add) + LeakyReLU as ONE autograd node (Block.forward / ConvolutionDownsample.forward,
    sparse_building_blocks.py:47-57,130-139).  Same kernels as the separate modules; in training mode the BatchNorm
    statistics are accumulated in the convolution kernel's epilogue (sei_conv_bn_forward) instead of a second read of
    the convolution output."""

    @staticmethod
    def forward(ctx, x, rb, w, b, gamma, beta, rmean, rvar, eps, momentum, training, leak, residual):
        nIn, nOut = w.shape[1], w.shape[2]
        if x.shape[1] != nIn:
            raise RuntimeError("convolution: input has %d planes, weight expects %d" % (x.shape[1], nIn))
        if x.shape[0] != rb.n_src:
            raise RuntimeError("convolution: %d feature rows but the metadata level has %d" % (x.shape[0], rb.n_src))
        dev = x.device
        y = torch.empty((rb.n_dst, nOut), dtype=torch.float32, device=dev)
        a = torch.empty_like(y)
        mean, invstd = torch.empty(nOut, device=dev), torch.empty(nOut, device=dev)
        check(lib().sei_conv_bn_forward(rb.h, _ptr(x), _ptr(w), _ptr(b), _ptr(y), nIn, nOut, _ptr(gamma), _ptr(beta),
                                        _ptr(rmean), _ptr(rvar), eps, momentum, 1 if training else 0, leak,
                                        _ptr(residual), _ptr(a), _ptr(mean), _ptr(invstd), _stream()))
        ctx.rb, ctx.leak, ctx.training = rb, leak, training
        ctx.has_bias, ctx.has_res, ctx.has_affine = b is not None, residual is not None, gamma is not None
        ctx.save_for_backward(x, y, a, w, gamma, mean, invstd, beta)
        return a

    @staticmethod
    def backward(ctx, da):
        x, y, a, w, gamma, mean, invstd, beta = ctx.saved_tensors
        da = _f32c(da)
        rb = ctx.rb
        nIn, nOut = w.shape[1], w.shape[2]
        n, dev = y.shape[0], y.device
        L, st = lib(), _stream()
        dy = torch.empty_like(y)
        dres = torch.empty_like(y) if ctx.has_res else None
        dg = torch.empty(nOut, dtype=torch.float32, device=dev) if ctx.has_affine else None
        dbe = torch.empty(nOut, dtype=torch.float32, device=dev) if ctx.has_affine else None
        want_db = ctx.has_bias and ctx.needs_input_grad[3]
        db = torch.empty(nOut, dtype=torch.float32, device=dev) if want_db else None
        # without a residual the activation mask is recomputed from y (a is not read)
        check(L.sei_bn_backward_colsum(_ptr(y), _ptr(a) if ctx.has_res else None, _ptr(da), n, nOut, _ptr(gamma), _ptr(beta),
                                       _ptr(mean), _ptr(invstd), ctx.leak, 1 if ctx.training else 0, _ptr(dy), _ptr(dres),
                                       _ptr(dg), _ptr(dbe), _ptr(db), st))
        dx = torch.empty((rb.n_src, nIn), dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(w) if ctx.needs_input_grad[2] else None
        if dx is not None or dw is not None:   # one call: the library overlaps dX and dW (dW on its own side stream)
            check(L.sei_conv_backward(rb.h, _ptr(x), _ptr(dy), _ptr(w), None, _ptr(dx), _ptr(dw), None, nIn, nOut, st))
            if dx is not None and dw is not None:
                _pin_to_side(x, dy, dw)
        return dx, None, dw, db, dg, dbe, None, None, None, None, None, None, dres


def _inference(bn, *tensors):
    """eval-mode BatchNorm and nothing to differentiate: the conv kernel applies BN / residual / activation in its store."""
    if bn.training or bn.running_mean is None:
        return False
    if not torch.is_grad_enabled():
        return True
    return not any(t is not None and t.requires_grad for t in tensors)


def _conv_bn_inference(x, rb, conv, bn, leak, residual):
    nIn, nOut = conv.weight.shape[1], conv.weight.shape[2]
    if x.shape[1] != nIn:
        raise RuntimeError("convolution: input has %d planes, weight expects %d" % (x.shape[1], nIn))
    if x.shape[0] != rb.n_src:
        raise RuntimeError("convolution: %d feature rows but the metadata level has %d" % (x.shape[0], rb.n_src))
    y = torch.empty((rb.n_dst, nOut), dtype=torch.float32, device=x.device)
    check(lib().sei_conv_bn_inference(rb.h, _ptr(x), _ptr(conv.weight), _ptr(conv.bias), nIn, nOut, _ptr(bn.weight),
                                      _ptr(bn.bias), _ptr(bn.running_mean), _ptr(bn.running_var), bn.eps, float(leak),
                                      _ptr(residual), _ptr(y), _stream()))
    return y


def conv_bn_act(x, conv, bn, leak=1.0, residual=None):
    """Fused ``conv -> bn (+ residual) -> LeakyReLU(leak)`` on SparseConvNetTensor ``x`` (conv: SubmanifoldConvolution
    or Convolution module, bn: BatchNormalization module); returns the SparseConvNetTensor of the result.  In eval mode
    with nothing to differentiate (the reference's inference path, ``trainer.validate``) the whole chain is ONE kernel:
    ``sei_conv_bn_inference``."""
    sp = _triple(x.spatial_size)
    if isinstance(conv, Convolution):
        rb = x.metadata.conv_rulebook(sp, conv.filter_size, conv.filter_stride)
        out = SparseConvNetTensor(metadata=x.metadata, spatial_size=torch.LongTensor(rb.spatial_out))
    else:
        rb = x.metadata.subm_rulebook(sp, conv.filter_size)
        out = SparseConvNetTensor(metadata=x.metadata, spatial_size=x.spatial_size)
    if _inference(bn, x.features, residual, conv.weight, conv.bias, bn.weight, bn.bias):
        out.features = _conv_bn_inference(_f32c(x.features), rb, conv, bn, leak,
                                          None if residual is None else _f32c(residual))
        return out
    out.features = _ConvBN.apply(_f32c(x.features), rb, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean,
                                 bn.running_var, bn.eps, bn.momentum, bn.training, float(leak),
                                 None if residual is None else _f32c(residual))
    return out


def residual_block_supported(nplanes):
    """The fused node needs the MFMA conv kernel (planes a multiple of 16, <= 256)."""
    return nplanes % 16 == 0 and 16 <= nplanes <= 256


def residual_block(x, conv1, bn1, conv2, bn2, leak1, leak2):
    """Fused ResidualBlock on SparseConvNetTensor ``x``: conv1/conv2 are SubmanifoldConvolution modules with equal
    filter size and nIn == nOut, bn1/bn2 BatchNormalization modules (affine)."""
    sp = _triple(x.spatial_size)
    rb = x.metadata.subm_rulebook(sp, conv1.filter_size)
    out = SparseConvNetTensor(metadata=x.metadata, spatial_size=x.spatial_size)
    if _inference(bn1, x.features, conv1.weight, conv1.bias, bn1.weight, bn1.bias, conv2.weight, conv2.bias, bn2.weight,
                  bn2.bias) and not bn2.training and bn2.running_mean is not None:
        xf = _f32c(x.features)
        a1 = _conv_bn_inference(xf, rb, conv1, bn1, leak1, None)
        out.features = _conv_bn_inference(a1, rb, conv2, bn2, leak2, xf)
        return out
    out.features = _ResidualBlock.apply(_f32c(x.features), rb, conv1.weight, conv1.bias, bn1.weight, bn1.bias,
                                        bn1.running_mean, bn1.running_var, conv2.weight, conv2.bias, bn2.weight, bn2.bias,
                                        bn2.running_mean, bn2.running_var, (bn1.eps, bn1.momentum, bn1.training),
                                        (bn2.eps, bn2.momentum, bn2.training), float(leak1), float(leak2))
    return out


class _Leaky(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, leak):
        y = torch.empty_like(x)
        check(lib().sei_leaky_forward(_ptr(x), _ptr(y), x.numel(), leak, _stream()))
        ctx.leak = leak
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = _f32c(dy)
        dx = torch.empty_like(y)
        check(lib().sei_leaky_backward(_ptr(y), _ptr(dy), _ptr(dx), y.numel(), ctx.leak, _stream()))
        return dx, None


class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        out = torch.empty_like(a)
        check(lib().sei_add_forward(_ptr(a), _ptr(b), _ptr(out), a.numel(), _stream()))
        return out

    @staticmethod
    def backward(ctx, d):
        return d, d


class _AddLeaky(torch.autograd.Function):
    """leaky(a + b): AddTable + LeakyReLU of the residual block in one pass."""

    @staticmethod
    def forward(ctx, a, b, leak):
        y = torch.empty_like(a)
        check(lib().sei_add_leaky_forward(_ptr(a), _ptr(b), _ptr(y), a.numel(), leak, _stream()))
        ctx.leak = leak
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = _f32c(dy)
        dz = torch.empty_like(y)
        check(lib().sei_leaky_backward(_ptr(y), _ptr(dy), _ptr(dz), y.numel(), ctx.leak, _stream()))
        return dz, dz, None


class _SparseToDense(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, metadata, spatial, n_planes):
        B = metadata.batch_size()
        ctx.md, ctx.spatial = metadata, spatial
        ctx.n = features.shape[0]
        dense = torch.empty((B, n_planes, spatial[0], spatial[1], spatial[2]), dtype=torch.float32,
                            device=features.device)
        check(lib().sei_sparse_to_dense_forward(metadata.handle, i32x3(spatial), _ptr(features), n_planes, _ptr(dense),
                                                _stream()))
        return dense

    @staticmethod
    def backward(ctx, ddense):
        ddense = _f32c(ddense)
        C = ddense.shape[1]
        dfeat = torch.empty((ctx.n, C), dtype=torch.float32, device=ddense.device)
        check(lib().sei_sparse_to_dense_backward(ctx.md.handle, i32x3(ctx.spatial), _ptr(ddense), C, _ptr(dfeat),
                                                 _stream()))
        return dfeat, None, None, None


def larcv_to_scn_device(larcv_batch):
    """Device-side ``larcvsparse_to_scnsparse_3d / _2d`` (src/io/data_transforms.py:21-49,198-253):
    larcv batch [B,P,M,D+1] (any device; -999 padded) -> (coords int32 [N,4] cuda, features fp32 [N,1] cuda, B),
    same values and order as the host transform, ready for InputLayer (no float64 host round trip)."""
    x = torch.as_tensor(larcv_batch)
    if not torch.cuda.is_available():
        raise RuntimeError("sparseeventid_amd.larcv_to_scn_device: no MI355X (cuda) device")
    x = _f32c(x.cuda())
    B, P, M, D1 = x.shape
    coords = torch.empty((B * P * M, 4), dtype=torch.int32, device=x.device)
    feats = torch.empty((B * P * M, 1), dtype=torch.float32, device=x.device)
    n = ctypes.c_int64()
    check(lib().sei_larcv_to_scn(_ptr(x), B, P, M, D1 - 1, _ptr(coords), _ptr(feats), _stream(), ctypes.byref(n)))
    return coords[:n.value], feats[:n.value], B


class _PoolTanh(torch.autograd.Function):
    """pooled[B,C] = sum over active rows of tanh(x) / volume  ==  AvgPool3d_full(tanh(SparseToDense(x)))."""

    @staticmethod
    def forward(ctx, features, metadata, spatial):
        B, C = metadata.batch_size(), features.shape[1]
        ctx.md, ctx.spatial = metadata, spatial
        ctx.save_for_backward(features)
        out = torch.empty((B, C), dtype=torch.float32, device=features.device)
        check(lib().sei_pool_tanh_forward(metadata.handle, i32x3(spatial), _ptr(features), C, _ptr(out), _stream()))
        return out

    @staticmethod
    def backward(ctx, dout):
        (features,) = ctx.saved_tensors
        dout = _f32c(dout)
        dfeat = torch.empty_like(features)
        check(lib().sei_pool_tanh_backward(ctx.md.handle, i32x3(ctx.spatial), _ptr(features), _ptr(dout),
                                           features.shape[1], _ptr(dfeat), _stream()))
        return dfeat, None, None


def pool_tanh(input):
    """Fused sparse head input (opt-in): global average of tanh over the full spatial extent, [B, C]."""
    return _PoolTanh.apply(_f32c(input.features), input.metadata, _triple(input.spatial_size))


# ============================================================================= functional helpers
def batchnorm_act(x, bn, leak=1.0, residual=None):
    """Fused BatchNormalization (+AddTable residual) + LeakyReLU on a features tensor."""
    return _BatchNorm.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum,
                            bn.training, float(leak), residual)


# ============================================================================= modules
class InputLayer(nn.Module):
    """scn.InputLayer(dimension, spatial_size, mode=3)  (resnet.py:26-29,40-43).

    forward((coords [N,4] (x0,x1,x2,batch) any real dtype / device, features [N,C], batch_size)).
    mode: 0/2 keep first duplicate, 1 keep last, 3 sum, 4 mean.
    """

    def __init__(self, dimension, spatial_size, mode=3):
        super().__init__()
        if dimension != 3:
            raise ValueError("sparseeventid_amd supports dimension=3 (2-D data uses [planes,H,W], resnet.py:26-29)")
        self.dimension = dimension
        self.spatial_size = torch.LongTensor(_triple(spatial_size))
        self.mode = mode
        self.device = None
        # Metadata of an EARLIER batch through the same network (set by a training loop, e.g. Trainer): the strided levels
        # it holds are discovered together with the input level (sei_input_build_like: ONE count read-back for the
        # InputLayer and all levels instead of one each).  None: the plain InputLayer.
        self.template = None

    def __getstate__(self):
        # the template is a live ctypes handle (device hash tables, rulebooks): never pickled or deep-copied with the model
        state = self.__dict__.copy()
        state["template"] = None
        return state

    def forward(self, input):
        if isinstance(input, SparseConvNetTensor):
            # already an input-level tensor (built ahead of time, e.g. Trainer.prefetch on a side stream)
            if _triple(input.spatial_size) != _triple(self.spatial_size):
                raise ValueError("InputLayer: prepared tensor has spatial size %r, layer expects %r"
                                 % (_triple(input.spatial_size), _triple(self.spatial_size)))
            return input
        coords, features = input[0], input[1]
        batch_size = int(input[2]) if len(input) > 2 and input[2] is not None else None
        dev = features.device if isinstance(features, torch.Tensor) and features.is_cuda else torch.device("cuda")
        if not torch.cuda.is_available():
            raise RuntimeError("sparseeventid_amd.InputLayer: no MI355X (cuda) device; this engine has no CPU path")
        coords = torch.as_tensor(coords)
        features = torch.as_tensor(features)
        if coords.dim() != 2 or coords.shape[1] != 4:
            raise ValueError("InputLayer: coords must be [N,4] = (x0,x1,x2,batch)")
        # upstream: coords.cpu().long(); here the cast happens on the device (plumbing)
        c32 = coords.to(device=dev).to(torch.int32).contiguous()
        feats = _f32c(features.to(device=dev))
        if feats.dim() == 1:
            feats = feats[:, None]
        if batch_size is None:
            batch_size = int(c32[:, 3].max().item()) + 1 if c32.shape[0] else 1
        md = Metadata()
        sp = _triple(self.spatial_size)
        n_rows = ctypes.c_int64()
        like = self.template
        if like is not None and like.handle is not None:
            check(lib().sei_input_build_like(md.handle, _ptr(c32), c32.shape[0], i32x3(sp), batch_size, self.mode,
                                             like.handle, _stream(), ctypes.byref(n_rows)))
        else:
            check(lib().sei_input_build(md.handle, _ptr(c32), c32.shape[0], i32x3(sp), batch_size, self.mode, _stream(),
                                        ctypes.byref(n_rows)))
        out = SparseConvNetTensor(metadata=md, spatial_size=self.spatial_size)
        out.features = _InputFeatures.apply(feats, md, n_rows.value)
        return out

    def __repr__(self):
        return "InputLayer(spatial_size=%r, mode=%d)" % (_triple(self.spatial_size), self.mode)


class SubmanifoldConvolution(nn.Module):
    """scn.SubmanifoldConvolution(dimension, nIn, nOut, filter_size, bias) (resnet.py:30-36,44-50,105-110)."""

    def __init__(self, dimension, nIn, nOut, filter_size, bias, groups=1):
        super().__init__()
        if dimension != 3 or groups != 1:
            raise ValueError("only dimension=3, groups=1 is supported")
        self.dimension, self.nIn, self.nOut = dimension, nIn, nOut
        self.filter_size = _triple(filter_size)
        self.filter_volume = self.filter_size[0] * self.filter_size[1] * self.filter_size[2]
        std = (2.0 / nIn / self.filter_volume) ** 0.5
        self.weight = nn.Parameter(torch.empty(self.filter_volume, nIn, nOut).normal_(0, std))
        if bias:
            self.bias = nn.Parameter(torch.zeros(nOut))
        else:
            self.register_parameter("bias", None)

    def forward(self, input):
        _require_cuda(input.features, "SubmanifoldConvolution")
        sp = _triple(input.spatial_size)
        rb = input.metadata.subm_rulebook(sp, self.filter_size)
        out = SparseConvNetTensor(metadata=input.metadata, spatial_size=input.spatial_size)
        out.features = _Conv.apply(_f32c(input.features), self.weight, self.bias, rb)
        return out

    def input_spatial_size(self, out_size):
        return out_size

    def __repr__(self):
        return "SubmanifoldConvolution %d->%d C%s" % (self.nIn, self.nOut, "x".join(map(str, self.filter_size)))


ValidConvolution = SubmanifoldConvolution


class Convolution(nn.Module):
    """scn.Convolution(dimension, nIn, nOut, filter_size, filter_stride, bias) (sparse_building_blocks.py:110-117)."""

    def __init__(self, dimension, nIn, nOut, filter_size, filter_stride, bias, groups=1):
        super().__init__()
        if dimension != 3 or groups != 1:
            raise ValueError("only dimension=3, groups=1 is supported")
        self.dimension, self.nIn, self.nOut = dimension, nIn, nOut
        self.filter_size = _triple(filter_size)
        self.filter_stride = _triple(filter_stride)
        self.filter_volume = self.filter_size[0] * self.filter_size[1] * self.filter_size[2]
        std = (2.0 / nIn / self.filter_volume) ** 0.5
        self.weight = nn.Parameter(torch.empty(self.filter_volume, nIn, nOut).normal_(0, std))
        if bias:
            self.bias = nn.Parameter(torch.zeros(nOut))
        else:
            self.register_parameter("bias", None)

    def forward(self, input):
        _require_cuda(input.features, "Convolution")
        sp = _triple(input.spatial_size)
        rb = input.metadata.conv_rulebook(sp, self.filter_size, self.filter_stride)
        out = SparseConvNetTensor(metadata=input.metadata, spatial_size=torch.LongTensor(rb.spatial_out))
        out.features = _Conv.apply(_f32c(input.features), self.weight, self.bias, rb)
        return out

    def input_spatial_size(self, out_size):
        o = _triple(out_size)
        return torch.LongTensor([(o[d] - 1) * self.filter_stride[d] + self.filter_size[d] for d in range(3)])

    def __repr__(self):
        return "Convolution %d->%d C%s/%s" % (self.nIn, self.nOut, "x".join(map(str, self.filter_size)),
                                              "x".join(map(str, self.filter_stride)))


class BatchNormalization(nn.Module):
    """scn.BatchNormalization(nPlanes, eps=1e-4, momentum=0.9, affine=True, leakiness=1)
    (sparse_building_blocks.py:39,122): statistics over all active rows of the batch."""

    def __init__(self, nPlanes, eps=1e-4, momentum=0.9, affine=True, leakiness=1):
        super().__init__()
        self.nPlanes, self.eps, self.momentum, self.affine, self.leakiness = nPlanes, eps, momentum, affine, leakiness
        self.register_buffer("running_mean", torch.zeros(nPlanes))
        self.register_buffer("running_var", torch.ones(nPlanes))
        if affine:
            self.weight = nn.Parameter(torch.ones(nPlanes))
            self.bias = nn.Parameter(torch.zeros(nPlanes))
        else:
            self.register_parameter("weight", None)
            self.register_parameter("bias", None)

    def forward_features(self, x, leak=None, residual=None):
        if x.shape[1] != self.nPlanes:
            raise RuntimeError("BatchNormalization: %d planes, expected %d" % (x.shape[1], self.nPlanes))
        return batchnorm_act(_f32c(x), self, self.leakiness if leak is None else leak, residual)

    def forward(self, input):
        _require_cuda(input.features, "BatchNormalization")
        out = SparseConvNetTensor(metadata=input.metadata, spatial_size=input.spatial_size)
        out.features = self.forward_features(input.features)
        return out

    def input_spatial_size(self, out_size):
        return out_size

    def __repr__(self):
        return "BatchNorm(%d,eps=%g,momentum=%g,affine=%s,leakiness=%g)" % (
            self.nPlanes, self.eps, self.momentum, self.affine, self.leakiness)


class BatchNormReLU(BatchNormalization):
    def __init__(self, nPlanes, eps=1e-4, momentum=0.9):
        super().__init__(nPlanes, eps, momentum, True, 0)


class BatchNormLeakyReLU(BatchNormalization):
    def __init__(self, nPlanes, eps=1e-4, momentum=0.9, leakiness=0.333):
        super().__init__(nPlanes, eps, momentum, True, leakiness)


class LeakyReLU(nn.Module):
    """scn.LeakyReLU(leak=1/3) (sparse_building_blocks.py:20,80,128)."""

    def __init__(self, leak=1 / 3):
        super().__init__()
        self.leak = leak

    def forward(self, input):
        _require_cuda(input.features, "LeakyReLU")
        out = SparseConvNetTensor(metadata=input.metadata, spatial_size=input.spatial_size)
        out.features = _Leaky.apply(_f32c(input.features), float(self.leak))
        return out

    def input_spatial_size(self, out_size):
        return out_size


class ReLU(LeakyReLU):
    def __init__(self):
        super().__init__(0.0)


class Identity(nn.Module):
    """scn.Identity (sparse_building_blocks.py:76,79)."""

    def forward(self, input):
        return input

    def input_spatial_size(self, out_size):
        return out_size


class AddTable(nn.Module):
    """scn.AddTable (sparse_building_blocks.py:82,96): sum of row-aligned feature tensors."""

    def forward(self, input):
        _require_cuda(input[0].features, "AddTable")
        out = SparseConvNetTensor(metadata=input[0].metadata, spatial_size=input[0].spatial_size)
        f = _f32c(input[0].features)
        for other in input[1:]:
            if other.features.shape != f.shape:
                raise RuntimeError("AddTable: feature shapes differ")
            f = _Add.apply(f, _f32c(other.features))
        out.features = f
        return out

    def input_spatial_size(self, out_size):
        return out_size


class SparseToDense(nn.Module):
    """scn.SparseToDense(dimension, nPlanes) (resnet.py:124-125): -> [B, nPlanes, D0, D1, D2]."""

    def __init__(self, dimension, nPlanes):
        super().__init__()
        self.dimension, self.nPlanes = dimension, nPlanes

    def forward(self, input):
        _require_cuda(input.features, "SparseToDense")
        if input.features.shape[1] != self.nPlanes:
            raise RuntimeError("SparseToDense: %d planes, expected %d" % (input.features.shape[1], self.nPlanes))
        return _SparseToDense.apply(_f32c(input.features), input.metadata, _triple(input.spatial_size), self.nPlanes)

    def input_spatial_size(self, out_size):
        return out_size


class Sequential(nn.Sequential):
    def input_spatial_size(self, out_size):
        for m in reversed(self._modules.values()):
            out_size = m.input_spatial_size(out_size)
        return out_size

    def add(self, module):
        self._modules[str(len(self._modules))] = module
        return self
