#!/usr/bin/env python3
"""Kernel experiments: forward time of one submanifold conv layer for workgroup-tile plans
(SEI_FWD_PLAN="maxs,cs,slj,nOut,blocks" is read by the library per call).
  python tools/sweep_tile.py --events 64 --level 2 --c 96 --plans default 2,1,2,96,2 4,1,2,96,2"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--events", type=int, default=64)
    ap.add_argument("--level", type=int, default=2)
    ap.add_argument("--c", type=int, default=96)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--plans", nargs="*", default=["default"])
    a = ap.parse_args()
    from bench import generate_batch
    arr, _ = generate_batch(0, 0, a.events, (900, 900, 900), 50000, "tracks", min(16, os.cpu_count() or 8))
    import torch
    import sparseeventid_amd.scn as scn
    from sparseeventid_amd.data import larcvsparse_to_scnsparse_3d
    c, f, b = larcvsparse_to_scnsparse_3d(arr)
    t = scn.InputLayer(3, [900, 900, 900])((torch.from_numpy(c), torch.from_numpy(f).cuda(), b))
    t.features = torch.randn(t.features.shape[0], 16, device="cuda")
    os.environ["SEI_CONV_KERNEL"] = "tile"
    for _ in range(a.level):
        t = scn.Convolution(3, 16, 16, 2, 2, False).cuda()(t)
    C = a.c
    x = torch.randn(t.features.shape[0], C, device="cuda")
    tt = scn.SparseConvNetTensor(x, t.metadata, t.spatial_size)
    conv = scn.SubmanifoldConvolution(3, C, C, 3, True).cuda()
    rb = t.metadata.subm_rulebook([int(v) for v in t.spatial_size], [3, 3, 3])
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ref = None
    import statistics
    times = {pl: [] for pl in a.plans}
    errs = {}
    R = rb.n_rules
    for rnd in range(a.rounds):        # plans interleaved over several rounds: clocks drift during a run
        for plan in a.plans:
            # "cols": the column-owner kernel (conv_cols.hip), "wave": the wave-private one, "forced-rounds": rounds wherever instantiated
            # "default" = the tile kernel with one offset per phase (forced kind 1); every "rounds*" plan forces the round form
            # "rounds+split" (and "rounds+split:KNOB=..."): the opt-in product mode (three-way bf16 splits on the bf16 matrix pipe)
            scn.set_product_mode("bf16x3" if "+split" in plan else "fp32")
            os.environ["SEI_CONV_KERNEL"] = "wave" if plan.startswith("wave") else {"cols": "cols"}.get(plan, "rounds" if "rounds" in plan else "tile")
            os.environ["SEI_TILE_ROUNDS"] = "1" if "rounds" in plan else "0"   # conv_round_mfma_kernel (two-offset rounds)
            os.environ.pop("SEI_WAVE_PLAN", None)
            if plan.startswith("wave:"):                  # "wave:SEI_WAVE_PLAN=32:2,128,1" (nOut:nt,tile rows,group) -- note the commas: one knob only
                k_, _, val = plan.split(":", 1)[1].partition("=")
                os.environ[k_] = val
            for kv in ("SEI_ROUND_JIT", "SEI_ROUND_RING", "SEI_ROUND_SLJ", "SEI_ROUND_NOFF", "SEI_ROUND_NW", "SEI_ROUND_MO", "SEI_ROUND_LONG", "SEI_ROUND_NB", "SEI_ROUND_FAM", "SEI_ROUND_LATE", "SEI_ROUND_PERSIST", "SEI_SPLIT_L2_BLOCKS"):
                os.environ.pop(kv, None)
            if plan in ("rounds+jit", "rounds-jit"):      # force / forbid the refilled-fragment form
                os.environ["SEI_ROUND_JIT"] = "1" if plan == "rounds+jit" else "0"
            if plan.startswith("rounds:"):                # "rounds:SEI_ROUND_RING=4,SEI_ROUND_JIT=1": knobs of an experiment build
                for kv in plan.split(":", 1)[1].split(","):
                    k_, _, val = kv.partition("=")
                    os.environ[k_] = val
            if plan in ("default", "cols", "forced-rounds") or plan.startswith("rounds") or plan.startswith("wave"):
                os.environ.pop("SEI_FWD_PLAN", None)
            else:
                os.environ["SEI_FWD_PLAN"] = plan
            try:
                out = conv(tt); torch.cuda.synchronize()
            except RuntimeError as e:
                errs[plan] = "ERROR %s" % e
                continue
            if ref is None:
                ref = out.features.detach().clone()
            errs[plan] = "max diff %.1e" % float((out.features.detach() - ref).abs().max() / ref.abs().max())
            ev0.record()
            for _ in range(a.iters):
                conv(tt)
            ev1.record(); torch.cuda.synchronize()
            times[plan].append(ev0.elapsed_time(ev1) / a.iters)
    for plan in a.plans:
        if not times[plan]:
            print("plan %-16s %s" % (plan, errs.get(plan)), flush=True)
            continue
        ts = times[plan][1:] if len(times[plan]) > 1 else times[plan]   # round 0 = warm-up
        med, best = statistics.median(ts), min(ts)
        print("L%d C=%d N=%d plan %-16s fwd median %.3f ms %.1f TFLOP/s | best %.3f ms %.1f TFLOP/s  (%s)" % (
            a.level, C, x.shape[0], plan, med, 2.0 * R * C * C / med / 1e9, best, 2.0 * R * C * C / best / 1e9, errs[plan]), flush=True)


if __name__ == "__main__":
    main()
