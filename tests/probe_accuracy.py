"""Developer probe (not collected by pytest): per-op rounding error of the engine and of the fp32 oracle against the
fp64 oracle, on one 50k-voxel event."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import scn_cpu
from tests.util import rel_err
import sparseeventid_amd.scn as scn
from sparseeventid_amd.data import larcvsparse_to_scnsparse_3d, synthetic_dune3d_batch

arr, _ = synthetic_dune3d_batch(0, 0, 2)
c, f, b = larcvsparse_to_scnsparse_3d(arr)
sp = [900, 900, 900]
torch.manual_seed(0)
for C in (32, 96, 192):
    t = scn.InputLayer(3, sp)((torch.from_numpy(c), torch.from_numpy(f).cuda(), b))
    n = t.features.shape[0]
    x = torch.randn(n, C) + 0.5
    w = torch.randn(27, C, C) * (1.0 / (27 * C) ** 0.5)
    xg = x.cuda().requires_grad_(True)
    t.features = xg
    conv = scn.SubmanifoldConvolution(3, C, C, 3, False).cuda()
    with torch.no_grad():
        conv.weight.copy_(w.cuda())
    yg = conv(t).features
    dy = torch.randn(n, C)
    yg.backward(dy.cuda())
    y = yg.detach().cpu()
    ot = scn_cpu.input_layer(c, torch.from_numpy(f), b, sp)
    # same row order on both sides (first appearance)
    ot32 = scn_cpu.OracleTensor(x, ot.coords, ot.spatial_size, ot.batch_size)
    ot64 = scn_cpu.OracleTensor(x.double(), ot.coords, ot.spatial_size, ot.batch_size)
    ot32.features = x.clone().requires_grad_(True); w32 = w.clone().requires_grad_(True)
    ot64.features = x.double().requires_grad_(True); w64 = w.double().requires_grad_(True)
    y32 = scn_cpu.subm_conv(ot32, w32, None, [3, 3, 3]).features
    y64 = scn_cpu.subm_conv(ot64, w64, None, [3, 3, 3]).features
    y32.backward(dy); y64.backward(dy.double())
    rms = lambda a, b: float((a.double().cpu() - b).pow(2).mean().sqrt())
    print("conv C=%d dX: engine vs fp64 %.2e (rms %.2e) | oracle32 %.2e (rms %.2e)   dW: engine %.2e (rms %.2e) | oracle32 %.2e (rms %.2e)" % (
        C, rel_err(xg.grad, ot64.features.grad), rms(xg.grad, ot64.features.grad), rel_err(ot32.features.grad, ot64.features.grad),
        rms(ot32.features.grad, ot64.features.grad), rel_err(conv.weight.grad, w64.grad), rms(conv.weight.grad, w64.grad),
        rel_err(w32.grad, w64.grad), rms(w32.grad, w64.grad)), flush=True)
    y32, y64 = y32.detach(), y64.detach()
    ot32.features = x; ot64.features = x.double()
    print("conv C=%d: engine vs fp64 %.2e (rms %.2e) | oracle32 vs fp64 %.2e (rms %.2e)" % (
        C, rel_err(y, y64), float((y.double() - y64).pow(2).mean().sqrt()), rel_err(y32, y64),
        float((y32.double() - y64).pow(2).mean().sqrt())), flush=True)
    # BN + leaky
    bn = scn.BatchNormLeakyReLU(C, leakiness=1.0 / 3.0).cuda().train()
    t.features = x.cuda()
    z = bn(t).features.detach().cpu()
    z32 = scn_cpu.leaky_relu(scn_cpu.batchnorm(ot32, torch.ones(C), torch.zeros(C), training=True)).features
    z64 = scn_cpu.leaky_relu(scn_cpu.batchnorm(ot64, torch.ones(C).double(), torch.zeros(C).double(), training=True)).features
    print("  BN C=%d: engine vs fp64 %.2e | oracle32 vs fp64 %.2e" % (C, rel_err(z, z64), rel_err(z32, z64)), flush=True)
