"""CPU: the C++/OpenMP restatement of the SCN-CPU inner loops (oracle/scn_omp.cpp, bench.py's timed CPU baseline)
against the torch restatement (oracle/scn_cpu.py) it stands beside: identical rule lists, conv forward / dX / dW and a
whole tiny encoder forward+backward to 1e-5."""
import numpy as np
import pytest
import torch

from oracle import scn_cpu
from tests.util import blob_cloud, random_cloud, rel_err


@pytest.fixture
def omp_backend():
    scn_cpu.use_backend("omp", threads=4)
    yield
    scn_cpu.use_backend("torch")


@pytest.mark.parametrize("filt", [[3, 3, 3], [5, 5, 5], [1, 3, 3]])
def test_omp_rule_lists_identical(filt):
    sp, B = [20, 18, 22], 3
    coords, feats = blob_cloud(3, B, sp, 500, 1)
    t = scn_cpu.input_layer(coords, torch.from_numpy(feats), B, sp)
    a = scn_cpu.subm_rules(t.coords, sp, filt)
    scn_cpu.use_backend("omp", threads=3)
    try:
        b = scn_cpu.subm_rules(t.coords, sp, filt)
    finally:
        scn_cpu.use_backend("torch")
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


@pytest.mark.parametrize("ci,co,filt", [(3, 5, [3, 3, 3]), (32, 32, [3, 3, 3]), (16, 48, [1, 3, 3]), (1, 8, [5, 5, 5])])
def test_omp_conv_matches_torch_restatement(ci, co, filt):
    sp, B = [16, 14, 18], 2
    coords, feats = blob_cloud(21, B, sp, 2600 if ci == 32 else 300, ci)   # > one 2048-rule block per offset
    K = filt[0] * filt[1] * filt[2]
    g = torch.Generator().manual_seed(2)
    W0 = torch.randn(K, ci, co, generator=g); b0 = torch.randn(co, generator=g)
    res = {}
    for backend in ("torch", "omp"):
        scn_cpu.use_backend(backend, threads=4) if backend == "omp" else scn_cpu.use_backend(backend)
        try:
            f = torch.from_numpy(feats).clone().requires_grad_(True)
            W = W0.clone().requires_grad_(True); b = b0.clone().requires_grad_(True)
            t = scn_cpu.input_layer(coords, f, B, sp)
            out = scn_cpu.subm_conv(t, W, b, filt)
            up = torch.randn(out.features.shape, generator=torch.Generator().manual_seed(5))
            out.features.backward(up)
            res[backend] = (out.features.detach(), f.grad, W.grad, b.grad)
        finally:
            scn_cpu.use_backend("torch")
    for a, bb in zip(res["torch"], res["omp"]):
        assert rel_err(bb, a) < 1e-5


def test_omp_strided_conv_and_encoder(omp_backend):
    cfg = scn_cpu.NetConfig(depth=2, n_initial_filters=8, n_output_filters=8, blocks_per_layer=1)
    sp = [16, 16, 16]
    coords, feats = blob_cloud(7, 2, sp, 300, 1)
    sd0 = scn_cpu.init_params(cfg, seed=1)

    def run():
        sd = {k: (v.clone().requires_grad_(True) if "running" not in k else v.clone()) for k, v in sd0.items()}
        out = scn_cpu.encoder_forward(sd, cfg, (coords, torch.from_numpy(feats), 2), [1] + sp)
        out.square().sum().backward()
        return out.detach(), sd

    o1, s1 = run()
    scn_cpu.use_backend("torch")
    o2, s2 = run()
    assert rel_err(o1, o2) < 1e-5
    gmax = max(float(s2[k].grad.abs().max()) for k in s2 if s2[k].grad is not None)
    for k in s2:
        if s2[k].grad is None:
            continue
        if float(s2[k].grad.abs().max()) < 1e-5 * gmax:
            assert float(s1[k].grad.abs().max()) < 1e-4 * gmax, k
        else:
            assert rel_err(s1[k].grad, s2[k].grad) < 1e-4, k


def test_omp_backend_reports_blas_and_threads():
    info = scn_cpu.omp_info()
    assert info["threads"] >= 1 and (info["blas"] is None or "openblas" in info["blas"])
