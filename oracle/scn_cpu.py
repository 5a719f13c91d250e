"""oracle/scn_cpu.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the SparseConvNet-CPU algorithm the reference reaches through
``import sparseconvnet as scn`` (/root/reference/src/networks/resnet.py:2,
sparse_building_blocks.py:3), plus the reference's own network topology
(resnet.py:12-161, sparse_building_blocks.py:18-139,231-264,
classification_head.py:7-55) and loss (src/utils/supervised_eventID.py:168-196).

The arithmetic lives in the third-party ``sparseconvnet`` package
(facebookresearch/SparseConvNet; NO pinned version -- requirements.txt is empty;
absent from /root/reference and from this image).  Its published CPU algorithm
is restated: per filter offset k, ``index_select -> mm -> index_add_`` over the
rule list built by oracle/scn_rules.c (integer part, plain C), float part in
torch-CPU fp32 so autograd supplies the matching backward.

PARITY UNPINNED: the reference holds no tests, golden vectors or fixtures for
this path.  The oracle is pinned by dense-convolution
equivalence instead (oracle/dense_ref.py; tests/test_oracle.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  It must never be on the product path.
"""
import ctypes
import math
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build_lib(force=False):
    """Compile oracle/scn_rules.c with gcc (building the checker is not using it)."""
    so = os.path.join(_HERE, "liboracle_rules.so")
    src = os.path.join(_HERE, "scn_rules.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", so, src])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build_lib())
        _LIB.oracle_input_layer.restype = ctypes.c_int64
        _LIB.oracle_subm_rules.restype = ctypes.c_int64
        _LIB.oracle_conv_rules.restype = ctypes.c_int64
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


# ----------------------------------------------------------------------------- C++/OpenMP backend (CPU baseline)
_OMP = None
_BACKEND = "torch"


def build_omp_lib(force=False):
    """Compile oracle/scn_omp.cpp (g++ -O3 -fopenmp; building the baseline is not using it)."""
    so = os.path.join(_HERE, "liboracle_omp.so")
    src = os.path.join(_HERE, "scn_omp.cpp")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-O3", "-std=c++17", "-fopenmp", "-fPIC", "-shared", "-o", so, src, "-ldl"])
    return so


def _omp():
    global _OMP
    if _OMP is None:
        import glob
        lib = ctypes.CDLL(build_omp_lib())
        lib.omp_subm_rules.restype = ctypes.c_int64
        lib.omp_load_blas.restype = ctypes.c_int
        blas = None
        try:      # the OpenBLAS inside the scipy wheel (symbols prefixed scipy_)
            import scipy
            cand = glob.glob(os.path.join(os.path.dirname(scipy.__file__), "..", "scipy.libs", "libscipy_openblas-*.so"))
            if cand and lib.omp_load_blas(cand[0].encode(), b"scipy_") == 0:
                blas = cand[0]
        except ImportError:
            pass
        lib.blas_path = blas   # None: plain-loop sgemm (slow; only for boxes without scipy)
        _OMP = lib
    return _OMP


def use_backend(name, threads=None):
    """'torch' (default: the checker of every parity test) or 'omp' (C++/OpenMP inner loops: bench.py's cpu_baseline)."""
    global _BACKEND
    if name not in ("torch", "omp"):
        raise ValueError(name)
    if name == "omp":
        lib = _omp()
        if threads:
            lib.omp_set_threads(int(threads))
    _BACKEND = name


def omp_info():
    lib = _omp()
    return {"threads": int(lib.omp_threads()), "blas": lib.blas_path}


def _triple(v):
    if isinstance(v, torch.Tensor):
        v = v.tolist()
    if isinstance(v, (int, np.integer)):
        return [int(v)] * 3
    return [int(x) for x in v]


class OracleTensor:
    """Mirror of scn.SparseConvNetTensor: .features [N,C], coords [N,4] (x0,x1,x2,b)."""

    def __init__(self, features, coords, spatial_size, batch_size, cache=None):
        self.features = features
        self.coords = coords            # numpy int32 [N,4]
        self.spatial_size = list(spatial_size)
        self.batch_size = batch_size
        self.cache = cache if cache is not None else {}

    def like(self, features):
        return OracleTensor(features, self.coords, self.spatial_size, self.batch_size, self.cache)


# ----------------------------------------------------------------------------- integer part
def input_rows(coords, spatial_size, batch_size):
    """scn.InputLayer coordinate pass (resnet.py:40-43,143)."""
    coords = np.ascontiguousarray(np.asarray(coords).astype(np.int64).astype(np.int32))
    n = coords.shape[0]
    sp = np.asarray(_triple(spatial_size), dtype=np.int32)
    row_of = np.empty(n, np.int32); first = np.empty(max(n, 1), np.int32)
    last = np.empty(max(n, 1), np.int32); cnt = np.empty(max(n, 1), np.int32)
    oc = np.empty((max(n, 1), 4), np.int32)
    nu = _lib().oracle_input_layer(_p(coords), ctypes.c_int64(n), _p(sp), ctypes.c_int32(batch_size),
                                   _p(row_of), _p(first), _p(last), _p(cnt), _p(oc))
    if nu < 0:
        raise RuntimeError("InputLayer: coordinate %d out of bounds" % (-nu - 1))
    return row_of, first[:nu].copy(), last[:nu].copy(), cnt[:nu].copy(), oc[:nu].copy()


def subm_rules(coords, spatial_size, filter_size):
    """Per-offset (in,out) rule lists for scn.SubmanifoldConvolution."""
    coords = np.ascontiguousarray(coords, dtype=np.int32)
    n = coords.shape[0]
    sp = np.asarray(_triple(spatial_size), np.int32); f = np.asarray(_triple(filter_size), np.int32)
    K = int(f.prod())
    start = np.zeros(K + 1, np.int64)
    cap = max(n * K, 1)
    # first try a modest capacity, regrow if needed
    cap_try = max(n * min(K, 32), 1)
    fn = _omp().omp_subm_rules if _BACKEND == "omp" else _lib().oracle_subm_rules
    while True:
        ri = np.empty(cap_try, np.int32); ro = np.empty(cap_try, np.int32)
        tot = fn(_p(coords), ctypes.c_int64(n), _p(sp), _p(f), _p(start), _p(ri), _p(ro), ctypes.c_int64(cap_try))
        if tot <= cap_try:
            break
        cap_try = min(cap, int(tot))
    return start, ri[:tot], ro[:tot]


def conv_rules(coords, spatial_size, size, stride):
    """Output sites + rule lists for scn.Convolution (sparse_building_blocks.py:110-117)."""
    coords = np.ascontiguousarray(coords, dtype=np.int32)
    n = coords.shape[0]
    sp = np.asarray(_triple(spatial_size), np.int32)
    sz = np.asarray(_triple(size), np.int32); st = np.asarray(_triple(stride), np.int32)
    K = int(sz.prod())
    mult = int(np.prod((sz + st - 1) // st))
    cap = max(n * mult, 1)
    oc = np.empty((cap, 4), np.int32); start = np.zeros(K + 1, np.int64)
    ri = np.empty(cap, np.int32); ro = np.empty(cap, np.int32)
    tot = ctypes.c_int64(0)
    nout = _lib().oracle_conv_rules(_p(coords), ctypes.c_int64(n), _p(sp), _p(sz), _p(st), _p(oc),
                                    ctypes.c_int64(cap), _p(start), _p(ri), _p(ro), ctypes.c_int64(cap),
                                    ctypes.byref(tot))
    out_sp = [int((sp[d] - sz[d]) // st[d] + 1) for d in range(3)]
    return oc[:nout].copy(), out_sp, start, ri[:tot.value].copy(), ro[:tot.value].copy()


# ----------------------------------------------------------------------------- float part
def input_layer(coords, features, batch_size, spatial_size, mode=3):
    """scn.InputLayer(dimension=3, spatial_size, mode) (resnet.py:26-29,40-43)."""
    if isinstance(coords, torch.Tensor):
        coords = coords.detach().cpu().numpy()
    row_of, first, last, cnt, oc = input_rows(coords, spatial_size, batch_size)
    # fp32 like the reference; a float64 feature tensor is kept (tests use an fp64 run of this oracle to
    # tell which fp32 side carries the rounding error when the two disagree near the tolerance)
    feats = features if features.requires_grad else features.detach().cpu()
    if feats.dtype != torch.float64 and not features.requires_grad:
        feats = feats.float()
    nu = oc.shape[0]
    if mode in (0, 2):
        out = feats.index_select(0, torch.from_numpy(first.astype(np.int64)))
    elif mode == 1:
        out = feats.index_select(0, torch.from_numpy(last.astype(np.int64)))
    else:
        out = torch.zeros(nu, feats.shape[1], dtype=feats.dtype).index_add(
            0, torch.from_numpy(row_of.astype(np.int64)), feats)
        if mode == 4:
            out = out / torch.from_numpy(cnt.astype(np.float32)).to(out.dtype)[:, None]
    return OracleTensor(out, oc, _triple(spatial_size), batch_size)


class _RuleConv(torch.autograd.Function):
    """SCN-CPU inner loops [upstream, SURVEY.md 2.2]: per offset k,
    forward  out[rules_k.out] += in[rules_k.in] @ W[k]       (index_select -> mm -> index_add_)
    backward dIn[in] += dOut[out] @ W[k]^T ;  dW[k] += in[in]^T @ dOut[out]."""

    @staticmethod
    def forward(ctx, x, W, start, ri_t, ro_t, n_out):
        out = torch.zeros(n_out, W.shape[2], dtype=x.dtype)
        for k in range(W.shape[0]):
            a, b = int(start[k]), int(start[k + 1])
            if b > a:
                out.index_add_(0, ro_t[a:b], x.index_select(0, ri_t[a:b]) @ W[k])
        ctx.save_for_backward(x, W, ri_t, ro_t)
        ctx.start = start
        return out

    @staticmethod
    def backward(ctx, dout):
        x, W, ri_t, ro_t = ctx.saved_tensors
        start = ctx.start
        dx = torch.zeros_like(x) if ctx.needs_input_grad[0] else None
        dW = torch.zeros_like(W)
        for k in range(W.shape[0]):
            a, b = int(start[k]), int(start[k + 1])
            if b > a:
                g = dout.index_select(0, ro_t[a:b])
                if dx is not None:
                    dx.index_add_(0, ri_t[a:b], g @ W[k].t())
                dW[k] = x.index_select(0, ri_t[a:b]).t() @ g
        return dx, dW, None, None, None, None


class _RuleConvOMP(torch.autograd.Function):
    """The same three loops in C++/OpenMP (oracle/scn_omp.cpp): blocks of rules, gather -> sgemm -> scatter-add."""

    @staticmethod
    def forward(ctx, x, W, start, ri, ro, n_out):
        x = x.contiguous(); W = W.contiguous()
        out = torch.zeros(n_out, W.shape[2], dtype=torch.float32)
        start = np.ascontiguousarray(start, dtype=np.int64)
        _omp().omp_conv_fwd(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(W.data_ptr()), _p(start), _p(ri), _p(ro),
                            ctypes.c_int(W.shape[0]), ctypes.c_int(W.shape[1]), ctypes.c_int(W.shape[2]),
                            ctypes.c_void_p(out.data_ptr()))
        ctx.save_for_backward(x, W)
        ctx.rules = (start, ri, ro)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, W = ctx.saved_tensors
        start, ri, ro = ctx.rules
        dout = dout.contiguous()
        dx = torch.zeros_like(x) if ctx.needs_input_grad[0] else None
        dW = torch.empty_like(W)
        _omp().omp_conv_bwd(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(W.data_ptr()), ctypes.c_void_p(dout.data_ptr()),
                            _p(start), _p(ri), _p(ro), ctypes.c_int(W.shape[0]), ctypes.c_int(W.shape[1]),
                            ctypes.c_int(W.shape[2]), ctypes.c_void_p(dx.data_ptr()) if dx is not None else None,
                            ctypes.c_void_p(dW.data_ptr()))
        return dx, dW, None, None, None, None


def _apply_rules(x, W, start, ri, ro, n_out):
    if _BACKEND == "omp" and x.dtype == torch.float32:
        return _RuleConvOMP.apply(x, W, start, np.ascontiguousarray(ri, dtype=np.int32),
                                  np.ascontiguousarray(ro, dtype=np.int32), n_out)
    ri_t = torch.from_numpy(ri.astype(np.int64)); ro_t = torch.from_numpy(ro.astype(np.int64))
    return _RuleConv.apply(x, W, start, ri_t, ro_t, n_out)


def subm_conv(t, weight, bias, filter_size):
    """scn.SubmanifoldConvolution forward; weight [K,nIn,nOut], bias [nOut] or None."""
    key = ("subm", tuple(_triple(filter_size)), tuple(t.spatial_size), t.coords.shape[0])
    if key not in t.cache:
        t.cache[key] = subm_rules(t.coords, t.spatial_size, filter_size)
    start, ri, ro = t.cache[key]
    out = _apply_rules(t.features, weight, start, ri, ro, t.coords.shape[0])
    if bias is not None:
        out = out + bias
    return t.like(out)


def strided_conv(t, weight, bias, size, stride):
    """scn.Convolution forward (new active set / spatial size)."""
    key = ("conv", tuple(_triple(size)), tuple(_triple(stride)), tuple(t.spatial_size), t.coords.shape[0])
    if key not in t.cache:
        t.cache[key] = conv_rules(t.coords, t.spatial_size, size, stride)
    oc, osp, start, ri, ro = t.cache[key]
    out = _apply_rules(t.features, weight, start, ri, ro, oc.shape[0])
    if bias is not None:
        out = out + bias
    return OracleTensor(out, oc, osp, t.batch_size, {})


class _BatchNormOMP(torch.autograd.Function):
    """Training-mode BatchNorm (+ LeakyReLU slope) as two OpenMP passes over the rows (oracle/scn_omp.cpp)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, leak):
        x = x.contiguous()
        n, C = x.shape
        y = torch.empty_like(x)
        mean, invstd, var = torch.empty(C), torch.empty(C), torch.empty(C)
        vp = lambda t: ctypes.c_void_p(t.data_ptr())
        _omp().omp_bn_fwd(vp(x), ctypes.c_int64(n), ctypes.c_int(C), vp(weight.contiguous()), vp(bias.contiguous()),
                          ctypes.c_float(eps), ctypes.c_float(leak), vp(y), vp(mean), vp(invstd), vp(var))
        ctx.save_for_backward(x, y, weight, mean, invstd)
        ctx.leak = leak
        ctx.mark_non_differentiable(mean, var)
        return y, mean, var

    @staticmethod
    def backward(ctx, dy, _dm, _dv):
        x, y, weight, mean, invstd = ctx.saved_tensors
        n, C = x.shape
        dy = dy.contiguous()
        dx, dg, db = torch.empty_like(x), torch.empty(C), torch.empty(C)
        vp = lambda t: ctypes.c_void_p(t.data_ptr())
        _omp().omp_bn_bwd(vp(x), vp(y), vp(dy), ctypes.c_int64(n), ctypes.c_int(C), vp(weight.contiguous()), vp(mean),
                          vp(invstd), ctypes.c_float(ctx.leak), vp(dx), vp(dg), vp(db))
        return dx, dg, db, None, None


class _LeakyOMP(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, leak):
        x = x.contiguous()
        y = torch.empty_like(x)
        _omp().omp_leaky(ctypes.c_void_p(x.data_ptr()), None, ctypes.c_int64(x.numel()), ctypes.c_float(leak),
                         ctypes.c_void_p(y.data_ptr()))
        ctx.save_for_backward(y)
        ctx.leak = leak
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(y)
        _omp().omp_leaky(ctypes.c_void_p(y.data_ptr()), ctypes.c_void_p(dy.data_ptr()), ctypes.c_int64(y.numel()),
                         ctypes.c_float(ctx.leak), ctypes.c_void_p(dx.data_ptr()))
        return dx, None


def batchnorm(t, weight, bias, running_mean=None, running_var=None, eps=1e-4, momentum=0.9,
              training=True, leakiness=1.0):
    """scn.BatchNormalization (sparse_building_blocks.py:39,122): stats over all active rows."""
    x = t.features
    if (_BACKEND == "omp" and training and weight is not None and x.dtype == torch.float32 and x.shape[0] > 0):
        y, mean, var = _BatchNormOMP.apply(x, weight, bias, float(eps), float(leakiness))
        if running_mean is not None:
            n = x.shape[0]
            with torch.no_grad():
                running_mean.mul_(momentum).add_((1 - momentum) * mean)
                running_var.mul_(momentum).add_((1 - momentum) * var * n / max(n - 1, 1))
        return t.like(y)
    if training:
        mean = x.mean(0)
        var = ((x - mean) ** 2).mean(0)
        if running_mean is not None:
            n = x.shape[0]
            with torch.no_grad():
                running_mean.mul_(momentum).add_((1 - momentum) * mean.detach())
                running_var.mul_(momentum).add_((1 - momentum) * var.detach() * n / max(n - 1, 1))
    else:
        mean, var = running_mean, running_var
    y = (x - mean) * torch.rsqrt(var + eps)
    if weight is not None:
        y = y * weight + bias
    if leakiness != 1.0:
        y = torch.where(y > 0, y, y * leakiness)
    return t.like(y)


def leaky_relu(t, leak=1.0 / 3.0):
    """scn.LeakyReLU(leak=1/3) (sparse_building_blocks.py:20,80,128)."""
    x = t.features
    if _BACKEND == "omp" and x.dtype == torch.float32:
        return t.like(_LeakyOMP.apply(x, float(leak)))
    return t.like(torch.where(x > 0, x, x * leak))


def add_table(ts):
    """scn.AddTable (sparse_building_blocks.py:82,96)."""
    f = ts[0].features
    for o in ts[1:]:
        f = f + o.features
    return ts[0].like(f)


def sparse_to_dense(t, n_planes):
    """scn.SparseToDense (resnet.py:123-128,156): [B,C,D0,D1,D2]."""
    c = torch.from_numpy(t.coords.astype(np.int64))
    sp = t.spatial_size
    dense = torch.zeros(t.batch_size, sp[0], sp[1], sp[2], n_planes, dtype=t.features.dtype)
    dense = dense.index_put((c[:, 3], c[:, 0], c[:, 1], c[:, 2]), t.features)
    return dense.permute(0, 4, 1, 2, 3).contiguous()


# ----------------------------------------------------------------------------- network
class NetConfig:
    """Defaults of src/config/network.py:23-38 (the only constructible branch)."""

    def __init__(self, dimension=3, depth=5, n_initial_filters=32, n_output_filters=128,
                 blocks_per_layer=4, filter_size=3, bias=True):
        self.dimension = dimension
        self.depth = depth
        self.n_initial_filters = n_initial_filters
        self.n_output_filters = n_output_filters
        self.blocks_per_layer = blocks_per_layer
        self.filter_size = filter_size
        self.bias = bias


def param_shapes(cfg):
    """state_dict keys/shapes of Encoder (resnet.py:12-131) with SCN parameter layout."""
    shapes = {}
    d3 = cfg.dimension == 3
    k5 = 125 if d3 else 25
    k3 = cfg.filter_size ** 3 if d3 else cfg.filter_size ** 2
    k2 = 8 if d3 else 4
    f = cfg.n_initial_filters
    shapes["initial_convolution.weight"] = (k5, 1, f)
    if cfg.bias:
        shapes["initial_convolution.bias"] = (f,)

    def series(prefix, c):
        for j in range(cfg.blocks_per_layer):
            for cv in ("convolution_1", "convolution_2"):
                p = "%s.block_%d.%s" % (prefix, j, cv)
                shapes[p + ".conv1.weight"] = (k3, c, c)
                if cfg.bias:
                    shapes[p + ".conv1.bias"] = (c,)
                for nm in ("weight", "bias", "running_mean", "running_var"):
                    shapes[p + ".norm." + nm] = (c,)

    for l in range(cfg.depth):
        series("network_layers.%d" % (2 * l), f)
        nf = f + cfg.n_initial_filters
        p = "network_layers.%d" % (2 * l + 1)
        shapes[p + ".conv.weight"] = (k2, f, nf)
        for nm in ("weight", "bias", "running_mean", "running_var"):
            shapes[p + ".norm." + nm] = (nf,)
        f = nf
    series("final_layer", f)
    shapes["bottleneck.weight"] = (1, f, cfg.n_output_filters)
    if cfg.bias:
        shapes["bottleneck.bias"] = (cfg.n_output_filters,)
    return shapes


def init_params(cfg, seed=0, heads=None):
    """SCN init [upstream]: conv weight N(0, sqrt(2/(nIn*K))), bias 0, BN gamma 1 beta 0."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, s in param_shapes(cfg).items():
        if k.endswith("running_mean"):
            sd[k] = torch.zeros(s)
        elif k.endswith("running_var"):
            sd[k] = torch.ones(s)
        elif ".norm.weight" in k:
            sd[k] = torch.ones(s)
        elif len(s) == 1:
            sd[k] = torch.zeros(s)
        else:
            sd[k] = torch.randn(s, generator=g) * math.sqrt(2.0 / (s[1] * s[0]))
    if heads:
        for name, n_out in heads.items():
            p = "classification_head.%s" % name
            sd[p + ".2.weight"] = (torch.rand(256, cfg.n_output_filters, generator=g) - 0.5) * 2 / math.sqrt(cfg.n_output_filters)
            sd[p + ".2.bias"] = (torch.rand(256, generator=g) - 0.5) * 2 / math.sqrt(cfg.n_output_filters)
            sd[p + ".5.weight"] = (torch.rand(n_out, 256, generator=g) - 0.5) * 2 / 16.0
            sd[p + ".5.bias"] = (torch.rand(n_out, generator=g) - 0.5) * 2 / 16.0
    return sd


def _block(t, sd, p, cfg, fs, activation, training):
    """Block (sparse_building_blocks.py:18-57): subm-conv -> BN -> activation."""
    t = subm_conv(t, sd[p + ".conv1.weight"], sd.get(p + ".conv1.bias"), fs)
    t = batchnorm(t, sd[p + ".norm.weight"], sd[p + ".norm.bias"], sd[p + ".norm.running_mean"],
                  sd[p + ".norm.running_var"], training=training)
    return leaky_relu(t) if activation else t


def _series(t, sd, prefix, cfg, fs, training):
    """BlockSeries of ResidualBlocks (sparse_building_blocks.py:61-100,231-264)."""
    for j in range(cfg.blocks_per_layer):
        p = "%s.block_%d" % (prefix, j)
        res = t
        out = _block(t, sd, p + ".convolution_1", cfg, fs, True, training)
        out = _block(out, sd, p + ".convolution_2", cfg, fs, False, training)
        t = leaky_relu(add_table([out, res]))
    return t


def encoder_forward(sd, cfg, x, image_size, training=True, mode=3, return_sparse=False):
    """Encoder.forward (resnet.py:141-161). x = (coords, features, batch_size).

    image_size: for 3-D the reference passes [1, D0, D1, D2] and uses image_size[1:]
    (resnet.py:40-43); for 2-D it passes [3, H, W] (resnet.py:26-29).
    """
    coords, features, batch_size = x
    d3 = cfg.dimension == 3
    spatial = list(image_size[1:]) if d3 else list(image_size)
    f5 = [5, 5, 5] if d3 else [1, 5, 5]
    f3 = [cfg.filter_size] * 3 if d3 else [1, cfg.filter_size, cfg.filter_size]
    f2 = [2, 2, 2] if d3 else [1, 2, 2]
    t = input_layer(coords, features, batch_size, spatial, mode)
    t = subm_conv(t, sd["initial_convolution.weight"], sd.get("initial_convolution.bias"), f5)
    for l in range(cfg.depth):
        t = _series(t, sd, "network_layers.%d" % (2 * l), cfg, f3, training)
        p = "network_layers.%d" % (2 * l + 1)
        # ConvolutionDownsample (sparse_building_blocks.py:103-139)
        t = strided_conv(t, sd[p + ".conv.weight"], None, f2, f2)
        t = batchnorm(t, sd[p + ".norm.weight"], sd[p + ".norm.bias"], sd[p + ".norm.running_mean"],
                      sd[p + ".norm.running_var"], training=training)
        t = leaky_relu(t)
    t = _series(t, sd, "final_layer", cfg, f3, training)
    t = subm_conv(t, sd["bottleneck.weight"], sd.get("bottleneck.bias"), [1, 1, 1])
    if return_sparse:
        return t
    dense = sparse_to_dense(t, cfg.n_output_filters)
    return torch.tanh(dense)


def head_forward(sd, dense, heads, training=False):
    """create_final_dense_chain (classification_head.py:19-28), Dropout in eval mode."""
    pooled = dense.mean(dim=(2, 3, 4))
    out = {}
    for name in heads:
        p = "classification_head.%s" % name
        h = pooled @ sd[p + ".2.weight"].t() + sd[p + ".2.bias"]
        if training:
            raise NotImplementedError("oracle runs the head with Dropout disabled")
        h = torch.nn.functional.leaky_relu(h, 0.01)
        out[name] = h @ sd[p + ".5.weight"].t() + sd[p + ".5.bias"]
    return out


def focal_loss(logits, labels):
    """calculate_loss, focal branch (supervised_eventID.py:168-196)."""
    loss = 0.0
    for key in logits:
        y = torch.nn.functional.one_hot(labels[key], logits[key].size(-1))
        sm = torch.softmax(logits[key], dim=-1).clamp(1e-7, 1.0 - 1e-7)
        l = -y * torch.log(sm) * (1 - sm) ** 2
        loss = loss + l.sum(-1).mean()
    return loss


# ----------------------------------------------------------------------------- SBND multiplane network
class SBNDConfig:
    """Hyper-parameters of the legacy SBND model; the reference defines none."""

    def __init__(self, n_initial_filters=32, res_blocks_per_layer=2, network_depth=6, depth_pre_merge=3,
                 spatial_size=(3, 2048, 1280)):
        self.n_initial_filters = n_initial_filters
        self.res_blocks_per_layer = res_blocks_per_layer
        self.network_depth = network_depth
        self.depth_pre_merge = depth_pre_merge
        self.spatial_size = list(spatial_size)


def sbnd_param_shapes(cfg, output_shape):
    """state_dict keys/shapes of src/networks/torch/sparseresnet.py:194-302 (batch_norm, leaky_relu on)."""
    shapes = {"initial_convolution.weight": (25, 1, cfg.n_initial_filters)}
    nf = cfg.n_initial_filters

    def series(prefix, c, nplanes):
        k = nplanes * 9
        for j in range(cfg.res_blocks_per_layer):
            p = "%s.block_%d" % (prefix, j)
            for cv, bn in (("conv1", "bn1"), ("conv2", "bn2")):
                shapes["%s.%s.weight" % (p, cv)] = (k, c, c)
                for nm in ("weight", "bias", "running_mean", "running_var"):
                    shapes["%s.%s.%s" % (p, bn, nm)] = (c,)

    def down(prefix, c, co):
        shapes[prefix + ".conv.weight"] = (4, c, co)
        for nm in ("weight", "bias", "running_mean", "running_var"):
            shapes["%s.bn.%s" % (prefix, nm)] = (co,)

    for l in range(cfg.depth_pre_merge):
        series("pre_convolutional_layers.%d" % (2 * l), nf, 1)
        down("pre_convolutional_layers.%d" % (2 * l + 1), nf, nf + cfg.n_initial_filters)
        nf += cfg.n_initial_filters
    for l in range(cfg.network_depth - cfg.depth_pre_merge):
        series("post_convolutional_layers.%d" % (2 * l), nf, 3)
        down("post_convolutional_layers.%d" % (2 * l + 1), nf, nf + cfg.n_initial_filters)
        nf += cfg.n_initial_filters
    for key, n_out in output_shape.items():
        series("final_layer.%s.0" % key, nf, 3)
        shapes["final_layer.%s.1.weight" % key] = (1, nf, n_out)
    return shapes


def sbnd_init_params(cfg, output_shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, s in sbnd_param_shapes(cfg, output_shape).items():
        if k.endswith("running_mean"):
            sd[k] = torch.zeros(s)
        elif k.endswith("running_var"):
            sd[k] = torch.ones(s)
        elif len(s) == 1:
            sd[k] = torch.ones(s) if k.endswith(".weight") else torch.zeros(s)
        else:
            sd[k] = torch.randn(s, generator=g) * math.sqrt(2.0 / (s[1] * s[0]))
    return sd


def _sbnd_series(t, sd, prefix, cfg, nplanes, training):
    """SparseBlockSeries of SparseResidualBlocks (sparseresnet.py:47-103,139-158)."""
    fs = [nplanes, 3, 3]
    for j in range(cfg.res_blocks_per_layer):
        p = "%s.block_%d" % (prefix, j)
        res = t
        out = subm_conv(t, sd[p + ".conv1.weight"], None, fs)
        out = batchnorm(out, sd[p + ".bn1.weight"], sd[p + ".bn1.bias"], sd[p + ".bn1.running_mean"],
                        sd[p + ".bn1.running_var"], training=training, leakiness=0.333)   # BatchNormLeakyReLU
        out = subm_conv(out, sd[p + ".conv2.weight"], None, fs)
        out = batchnorm(out, sd[p + ".bn2.weight"], sd[p + ".bn2.bias"], sd[p + ".bn2.running_mean"],
                        sd[p + ".bn2.running_var"], training=training)
        t = leaky_relu(add_table([out, res]))
    return t


def _sbnd_down(t, sd, p, training):
    """SparseConvolutionDownsample (sparseresnet.py:108-137): [1,2,2]/[1,2,2] conv -> BN -> LeakyReLU."""
    t = strided_conv(t, sd[p + ".conv.weight"], None, [1, 2, 2], [1, 2, 2])
    t = batchnorm(t, sd[p + ".bn.weight"], sd[p + ".bn.bias"], sd[p + ".bn.running_mean"], sd[p + ".bn.running_var"],
                  training=training)
    return leaky_relu(t)


def sbnd_forward(sd, cfg, x, output_shape, training=True, mode=3):
    """ResNet.forward (sparseresnet.py:304-334): returns {key: [B, n_out]}."""
    coords, features, batch_size = x
    t = input_layer(coords, features, batch_size, cfg.spatial_size, mode)
    t = subm_conv(t, sd["initial_convolution.weight"], None, [1, 5, 5])
    for l in range(cfg.depth_pre_merge):
        t = _sbnd_series(t, sd, "pre_convolutional_layers.%d" % (2 * l), cfg, 1, training)
        t = _sbnd_down(t, sd, "pre_convolutional_layers.%d" % (2 * l + 1), training)
    for l in range(cfg.network_depth - cfg.depth_pre_merge):
        t = _sbnd_series(t, sd, "post_convolutional_layers.%d" % (2 * l), cfg, 3, training)
        t = _sbnd_down(t, sd, "post_convolutional_layers.%d" % (2 * l + 1), training)
    out = {}
    for key, n_out in output_shape.items():
        u = _sbnd_series(t, sd, "final_layer.%s.0" % key, cfg, 3, training)
        u = subm_conv(u, sd["final_layer.%s.1.weight" % key], None, [1, 1, 1])
        out[key] = sparse_to_dense(u, n_out).mean(dim=(2, 3, 4))
    return out
