#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X (BASELINE.json):
events/sec (fwd+bwd) of the DUNE-3D sparse ResNet, 64-event batch per GPU, synthetic ~50k-voxel
events in a 900^3 grid.  One "step" = forward + focal loss + backward + gradient
all-reduce (N>1) + Adam step over one batch whose input tensors are already resident in HBM.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (contract in the build prompt) carrying `roofline` (dominant kernel =
the MFMA gather-GEMM conv forward/dX kernel, HIP-event timed inside the timed region) and, at N=1,
`cpu_baseline` (the CPU oracle -- a port of the SparseConvNet-CPU algorithm -- on a bounded sample).
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the host driver of this pool only supports dmabuf IPC: RCCL between ranks fails without it (read when the HSA runtime starts)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402


def _gen_event(args):
    from sparseeventid_amd.data import synthetic_event, synthetic_sbnd_event, uniform_event
    seed, grid, nvox, kind = args
    if kind == "sbnd":
        return synthetic_sbnd_event(seed, 3, (grid[1], grid[2]), nvox)
    return (synthetic_event if kind == "tracks" else uniform_event)(seed, grid, nvox)


def generate_batch(rank, batch_index, batch, grid, nvox, kind, workers):
    """larcv-layout [B,1,nvox,4] fp32 + labels; seed = 1000003*rank + 64*batch + event."""
    jobs = [(1000003 * rank + 64 * batch_index + e, grid, nvox, kind) for e in range(batch)]
    if workers > 1:
        with mp.get_context("fork").Pool(workers) as pool:
            evs = pool.map(_gen_event, jobs)
    else:
        evs = [_gen_event(j) for j in jobs]
    arr = np.stack(evs) if kind == "sbnd" else np.stack(evs)[:, None]
    rng = np.random.default_rng(7 + 1000003 * rank + batch_index)
    names = ("labelneutID", "labelprotID", "labelnpiID", "labelcpiID")
    labels = {k: rng.integers(0, n, batch).astype(np.int64) for k, n in zip(names, (3, 3, 2, 2))}
    return arr, labels


def cpu_baseline(arr, labels, image_size, n_events, fwd_only, threads=16, warmup=2, passes=5, fwd_training=False):
    """Time the CPU restatement of the SCN-CPU algorithm (the SCN binary is unavailable) on the first `n_events`
    events of the same batch: C++/OpenMP inner loops (oracle/scn_omp.cpp: host hash + rule lists probed by all cores,
    per offset gather -> sgemm (OpenBLAS) -> scatter-add, two-pass BatchNorm, LeakyReLU) under the reference topology
    (oracle/scn_cpu.py), forward+backward (or forward only); `warmup` untimed passes, median of `passes` timed ones.
    Checker / baseline code, timed here ONLY as the reported baseline."""
    import torch
    from oracle import scn_cpu
    from sparseeventid_amd.data import larcvsparse_to_scnsparse_3d
    # a one-GPU box owns a 16-core CPU share; more threads than that only oversubscribes
    threads = max(1, min(threads, os.cpu_count() or 1))
    torch.set_num_threads(threads)
    scn_cpu.use_backend("omp", threads=threads)
    info = scn_cpu.omp_info()
    heads = {"labelneutID": 3, "labelprotID": 3, "labelnpiID": 2, "labelcpiID": 2}
    cfg = scn_cpu.NetConfig()
    sd = scn_cpu.init_params(cfg, seed=0, heads=heads)
    sd = {k: (v.requires_grad_(True) if "running" not in k else v) for k, v in sd.items()}
    c, f, b = larcvsparse_to_scnsparse_3d(arr[:n_events])
    lab = {k: torch.from_numpy(v[:n_events]) for k, v in labels.items()}
    times = []
    for it in range(warmup + passes):
        for v in sd.values():
            v.grad = None
        t0 = time.time()
        if fwd_only:
            with torch.no_grad():
                # inference as the reference runs it: eval-mode BatchNorm (running statistics), unless --fwd-bn train
                dense = scn_cpu.encoder_forward(sd, cfg, (c, torch.from_numpy(f), b), image_size, training=fwd_training)
                scn_cpu.head_forward(sd, dense, heads)
        else:
            dense = scn_cpu.encoder_forward(sd, cfg, (c, torch.from_numpy(f), b), image_size)
            loss = scn_cpu.focal_loss(scn_cpu.head_forward(sd, dense, heads), lab)
            loss.backward()
        if it >= warmup:
            times.append(time.time() - t0)
    scn_cpu.use_backend("torch")
    dt = float(np.median(times))
    return {"value": n_events / dt, "unit": "events/s", "cores": info["threads"], "kind": "port",
            "sample": "%d of the batch's events, %s, %d warm-up + %d timed passes (median %.1f s, all %s); C++/OpenMP "
                      "restatement of the SparseConvNet-CPU algorithm: per offset gather -> sgemm (%s) -> scatter-add, "
                      "OpenMP rule lists / BatchNorm / LeakyReLU (SCN binary unavailable)"
                      % (n_events, ("fwd (batch-statistics BatchNorm)" if fwd_training else "fwd (eval-mode BatchNorm)")
                         if fwd_only else "fwd+bwd", warmup, passes, dt,
                         "/".join("%.1f" % t for t in times),
                         "OpenBLAS from the scipy wheel" if info["blas"] else "plain loops, no BLAS found")}


def self_launch(n, one_device, launch_check):
    """`bench.py --gpus N` without a launcher: start N rank processes through torch.distributed.run (one per GPU, RCCL)
    and relay what they print.  Returns the launcher's exit code; never prints a line of its own."""
    import socket
    import subprocess
    if not one_device and not launch_check:
        import torch
        if torch.cuda.device_count() < n:      # counting does not initialise the GPU
            print("bench.py: --gpus %d but only %d device(s) visible" % (n, torch.cuda.device_count()), file=sys.stderr)
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def launch_check(args, world, rank):
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group(args.backend, rank=rank, world_size=world)
    t = torch.tensor([float(rank)])
    if args.backend == "nccl":
        torch.cuda.set_device(0 if args.one_device else int(os.environ.get("LOCAL_RANK", "0")))
        t = t.cuda()
    dist.all_reduce(t)
    ok = float(t.item()) == world * (world - 1) / 2
    if rank == 0:
        print(json.dumps({"launch_check": ok, "n_gpus": world, "backend": args.backend}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    if not ok:
        raise SystemExit("launch check: the ranks' sum is wrong")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="events per GPU")
    ap.add_argument("--voxels", type=int, default=50000)
    ap.add_argument("--grid", type=int, nargs=3, default=[900, 900, 900])
    ap.add_argument("--kind", default="tracks", choices=["tracks", "uniform"])
    ap.add_argument("--mode", default="train", choices=["train", "fwd"])
    ap.add_argument("--model", default="dune3d", choices=["dune3d", "sbnd"],
                    help="dune3d = BASELINE configs[1-3] (headline); sbnd = configs[4] 2-D multiplane shared-weight ResNet")
    ap.add_argument("--cpu-events", type=int, default=16, help="events in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=16, help="torch threads of the CPU-baseline leg")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--one-device", action="store_true", help="rehearsal: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--force-pg", action="store_true", help="initialise the process group (default backend nccl = RCCL) even at "
                    "--gpus 1: the gradient all-reduce, parameter / buffer broadcasts and the cross-rank checks then really run "
                    "through the backend at world size 1")
    ap.add_argument("--no-prof", action="store_true", help="skip HIP-event timing of the conv kernels")
    ap.add_argument("--overlap-backward", type=int, nargs="?", const=2, default=2, choices=[0, 1, 2],
                    help="where dW runs (scn.set_backward_streams): 2 (default) = on the library's side stream with ONE join per "
                    "backward pass, so that the backlog of MFMA-bound dW kernels runs beside the dX chain AND beside its "
                    "HBM-bound BatchNorm-backward kernels (353.7 -> 342.2 ms per step); 1 = joined per layer (350.3); "
                    "0 = dW then dX on one stream.  With 1 / 2 the dX and dW launches share the GPU, so their individual "
                    "durations measure the sharing: the roofline block then prices the FORWARD launches of the kernel")
    ap.add_argument("--prefetch", dest="prefetch", action="store_true", default=True,
                    help="(default) step i+1's InputLayer hash + rulebooks are built on a side stream WHILE step i computes "
                    "(Trainer.prefetch(after=event)): once per step, inside the timed region; 353.7 -> 349.5 ms per step")
    ap.add_argument("--fwd-bn", choices=["eval", "train"], default="eval",
                    help="--mode fwd only: BatchNorm as the reference's inference path runs it (bin/exec.py:206-224 -> "
                    "trainer.validate -> model.eval(): running statistics; conv + BN + residual + activation are then ONE "
                    "kernel, sei_conv_bn_inference), or 'train': batch statistics under no_grad (rounds 1-2 measured this)")
    ap.add_argument("--no-prefetch", dest="prefetch", action="store_false",
                    help="rulebooks at the top of the forward pass instead (Trainer.batch_rulebooks)")
    ap.add_argument("--batches", type=int, default=3,
                    help="distinct pre-staged batches (seeds 64*b + e, SURVEY 8d) cycled through the timed loop; `value` is "
                    "measured over them, and the same loop over batch 0 alone is reported beside it (single_batch_repeat)")
    ap.add_argument("--product-mode", default="fp32", choices=["fp32", "bf16x3"],
                    help="how the forward / dX convolutions multiply (scn.set_product_mode): fp32 = fp32 matrix instructions (the "
                    "default, what `value` is measured in); bf16x3 = OPT-IN: every fp32 operand as an exact sum of three bf16 "
                    "numbers, six part products on the bf16 matrix pipe, fp32 accumulation (per product within 2^-23 of exact)")
    ap.add_argument("--no-optin-leg", dest="optin_leg", action="store_false", default=True,
                    help="skip the extra timed loop that reports the opt-in bf16x3 product mode beside `value`")
    ap.add_argument("--launch-check", action="store_true",
                    help="rendezvous only: every rank joins the process group and all-reduces its rank, rank 0 prints "
                    '{"launch_check": true, "n_gpus": world}; no GPU work (tests of the self-launcher, runs on CPU with --backend gloo)')
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # Started WITHOUT a launcher: become one.  Nothing here has touched the GPU (counting devices does not initialise
        # it), the ranks are fresh processes, the parent only relays rank 0's JSON line and the exit code.
        sys.exit(self_launch(args.gpus, args.one_device, args.launch_check))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # never measure a world other than the one asked for (a line with n_gpus != --gpus would void a scaling record)
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node %d, or without a launcher" % (
            args.gpus, world, args.gpus))
    if args.launch_check:
        return launch_check(args, world, rank)
    if not args.one_device:
        import torch as _t
        if _t.cuda.device_count() < world:     # counting does not initialise the GPU
            raise SystemExit("--gpus %d but only %d device(s) visible (one rank per GPU; --one-device is the rehearsal mode)" % (
                world, _t.cuda.device_count()))

    # synthetic input is generated BEFORE the GPU is touched (fork pool)
    ncpu = os.cpu_count() or 8
    # half of this rank's share of the host's cores: 8 ranks on one host must not oversubscribe it while they all generate
    workers = int(os.environ.get("SEI_GEN_WORKERS", "0")) or max(1, min(16, ncpu // (2 * max(world, 1))))
    # (under `rocprofv3 --pmc` the GPU is initialised before python starts: set SEI_GEN_WORKERS=1, no fork)
    grid = tuple(args.grid)
    if args.model == "sbnd":
        grid, args.kind = (3, 2048, 1280), "sbnd"
        if args.voxels == 50000:
            args.voxels = 15000
    nb = max(1, args.batches)
    batches = [generate_batch(rank, b, args.batch, grid, args.voxels, args.kind, workers) for b in range(nb)]
    arr, labels = batches[0]

    import torch
    import torch.distributed as dist
    from sparseeventid_amd import _lib
    from sparseeventid_amd.distributed import init_from_env
    from sparseeventid_amd.train import Trainer

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    if args.one_device:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        os.environ["LOCAL_RANK"] = str(local)
    grouped = world > 1 or args.force_pg
    if args.force_pg and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
    init_from_env(args.backend if grouped else None, force=args.force_pg)
    image_size = [1] + list(grid)
    tr = Trainer(image_size=image_size, device="cuda", seed=0, model="sbnd" if args.model == "sbnd" else "resnet")
    tr.arena.broadcast_parameters()
    tr.buffers.broadcast()
    resident = [tr.stage(tr.to_scn_input(a), l) for a, l in batches]   # every batch's inputs resident in HBM before timing
    images, labs = [x[0] for x in resident], [x[1] for x in resident]
    image = images[0]

    # default: the InputLayer hash and the rulebooks of step i+1 are built on a side stream while step i computes
    # (Trainer.prefetch) -- once per step, inside the timed region: K timed steps consume K builds and issue K builds.
    # --no-prefetch: at the top of the forward pass.
    prefetch = args.prefetch
    if args.one_device and world > 1 and args.overlap_backward:
        # rehearsal only: TWO processes on one device, each with three busy queues (step, dW backlog, rulebook build),
        # time-slice each other to a crawl (14.8 s per step against 0.70 s with either overlap alone; tools/rehearsal_ab.sh)
        prefetch = False
    inputs_ready = torch.cuda.Event()
    inputs_ready.record()
    staged = [tr.prefetch(image, after=inputs_ready) if prefetch else image]
    cur = {"i": 0, "cycle": nb > 1}   # step i trains on batch i % nb (cycle) or on batch 0 (single-batch repeat)

    if args.mode == "fwd" and args.fwd_bn == "eval":
        tr.encoder.eval(); tr.head.eval()

    def step():
        b, bn = (cur["i"] % nb, (cur["i"] + 1) % nb) if cur["cycle"] else (0, 0)
        if args.mode == "train":
            tr.train_step(staged[0], labs[b])
        else:
            with torch.no_grad():
                tr.forward(staged[0])
        staged[0] = tr.prefetch(images[bn], after=inputs_ready) if prefetch else images[bn]
        cur["i"] += 1

    def fence():
        torch.cuda.synchronize()
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    lib = _lib.lib()
    from sparseeventid_amd import scn
    scn.set_product_mode(args.product_mode)
    if args.overlap_backward and args.mode == "train":
        scn.set_backward_streams(args.overlap_backward)
    shared = bool(args.overlap_backward) and args.mode == "train"   # dX launches share the GPU with dW launches
    import ctypes

    def pool_held_now():
        h, u = ctypes.c_int64(), ctypes.c_int64()
        lib.sei_pool_stats(ctypes.byref(h), ctypes.byref(u))
        return h.value

    for _ in range(args.warmup):
        step()
    fence()
    pool_warm = pool_held_now()
    if not args.no_prof:
        lib.sei_prof_reset()
        lib.sei_prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    lib.sei_prof_enable(0)
    pool_end = pool_held_now()
    def conv_tflop_per_step():
        q = (ctypes.c_double * 4)()
        tot = 0.0
        for kind in (0, 1, 2, 6, 7):
            lib.sei_prof_query(kind, q)
            tot += q[2]
        return tot / args.steps / 1e12

    single = single_tflop = saved = None
    cycled_tflop = conv_tflop_per_step() if not args.no_prof else None
    if nb > 1:
        # the same loop on ONE batch (what rounds 1-4 reported as `value`): identical sizes every step.  The profile of the
        # cycled loop (what `roofline` reports) is saved and restored around it; this loop's own flop count says how much
        # of the difference between the two rates is the batches' different amounts of work.
        if not args.no_prof:
            saved = [(ctypes.c_double * 4)() for _ in range(8)]
            for kind in range(8):
                lib.sei_prof_query(kind, saved[kind])
        cur["cycle"] = False
        staged[0] = tr.prefetch(images[0], after=inputs_ready) if prefetch else images[0]
        step()
        fence()
        if not args.no_prof:
            lib.sei_prof_reset()
            lib.sei_prof_enable(1)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        single = (time.perf_counter() - t1) / args.steps * 1e3
        lib.sei_prof_enable(0)
        if not args.no_prof:
            single_tflop = conv_tflop_per_step()
    # OPT-IN leg: the same cycled loop with the forward / dX products as three-way bf16 splits (never `value`), and how far the
    # encoder's dense output of batch 0 (57 layers deep) moves between the two modes with the same weights
    optin = None
    if args.optin_leg and args.product_mode == "fp32" and args.mode == "train" and args.model == "dune3d":
        def dense_of(mode):   # the encoder's dense output (the heads hold Dropout: their logits differ between any two passes)
            scn.set_product_mode(mode)
            with torch.no_grad():
                return tr.encoder(images[0]).detach().float().clone()
        da, db = dense_of("fp32"), dense_of("bf16x3")
        drift = float((db - da).abs().max() / da.abs().max().clamp_min(1e-30))
        del da, db
        cur["cycle"] = nb > 1
        staged[0] = tr.prefetch(images[cur["i"] % nb if cur["cycle"] else 0], after=inputs_ready) if prefetch else images[cur["i"] % nb if cur["cycle"] else 0]
        step(); step()
        fence()
        t2 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        oms = (time.perf_counter() - t2) / args.steps * 1e3
        scn.set_product_mode("fp32")
        optin = {"ms_per_step": oms, "encoder_output_max_rel_diff_vs_fp32_products": drift}
    ms = dt / args.steps * 1e3
    sync = None
    if grouped:
        t = torch.tensor([ms, single or 0.0, optin["ms_per_step"] if optin else 0.0], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms = float(t[0].item())
        single = float(t[1].item()) if single is not None else None
        if optin:
            optin["ms_per_step"] = float(t[2].item())
        # data-parallel invariants, checked AFTER the timed region: every rank ended with bit-identical parameters
        # (elementwise max == min over ranks) and every rank trained on a different batch
        if args.mode == "train":
            lo, hi = tr.arena.flat.clone(), tr.arena.flat.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            if not torch.equal(lo, hi):
                raise SystemExit("rank %d: parameters differ across ranks after %d steps" % (rank, args.steps + args.warmup))
        sig = torch.tensor([float(np.asarray(arr, dtype=np.float64).sum())], device="cuda", dtype=torch.float64)
        sigs = [torch.zeros_like(sig) for _ in range(world)]
        dist.all_gather(sigs, sig)
        if len({float(x.item()) for x in sigs}) != world:
            raise SystemExit("ranks were given identical batches")
        sync = {"parameters_bit_identical_across_ranks": args.mode == "train", "distinct_batches": True,
                "backend": dist.get_backend(), "world_size": world, "one_device": bool(args.one_device)}

    staging = None
    if world == 1 and args.mode == "train" and args.model == "dune3d":
        # the same step INCLUDING input staging: pinned host larcv block -> H2D -> device-side transform
        # (Trainer.stage_larcv) -> step.  Reported beside `value`, never as `value` (inputs resident is the metric).
        pinned = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32)).pin_memory()
        k2 = max(1, min(args.steps, 5))
        copy = torch.cuda.Stream()

        def stage_next():
            # H2D + device-side transform on a copy stream; with --prefetch the hash / rulebook build follows it on the
            # rulebook side stream, both beside the step that is still computing
            with torch.cuda.stream(copy):
                img, lb = tr.stage_larcv(pinned, labels)
                ev = torch.cuda.Event()
                ev.record(copy)
            return (tr.prefetch(img, after=ev) if prefetch else img), lb, ev

        def staged_step(item):
            img, lb, ev = item
            main = torch.cuda.current_stream()
            main.wait_event(ev)
            for t in list(lb.values()) + ([] if prefetch else [x for x in img[:2] if isinstance(x, torch.Tensor)]):
                t.record_stream(main)      # allocated on the copy stream, read by the step
            tr.train_step(img, lb)

        item = stage_next()
        staged_step(item)
        item = stage_next()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(k2):
            staged_step(item)
            item = stage_next()
        torch.cuda.synchronize()
        ms2 = (time.perf_counter() - t1) / k2 * 1e3
        staging = {"ms_per_step": ms2, "events_per_s": args.batch / (ms2 * 1e-3), "steps": k2,
                   "path": "pinned host larcv fp32 [B,1,M,4] -> H2D -> sei_larcv_to_scn on the device (copy stream, one batch ahead) -> train step"}

    pool_held = pool_held_now()
    if rank == 0:
        roof = None
        if not args.no_prof:
            # dominant kernel: the gather->MFMA->LDS-accumulate conv kernel (forward launches + dX launches run the
            # same kernel); algorithmic flops = 2*R*Cin*Cout per launch, duration from HIP events.
            q = (ctypes.c_double * 4)()
            tot_ms = tot_fl = tot_n = 0.0
            step_fl = 0.0
            kinds = {}
            for kind, name in ((0, "conv_fwd"), (1, "conv_bwd_input"), (2, "conv_bwd_weight"), (3, "rulebook_build"),
                               (4, "batchnorm"), (6, "conv_fwd_32planes"), (7, "conv_bwd_input_32planes")):
                if saved is not None:      # the profile of the CYCLED loop (the single-batch loop ran after it)
                    q = saved[kind]
                else:
                    lib.sei_prof_query(kind, q)
                kinds[name] = {"launches": q[0], "ms": q[1], "gflop": q[2] / 1e9, "gbyte": q[3] / 1e9}
                if kind in ((0, 6) if shared else (0, 1, 6, 7)):
                    tot_n += q[0]; tot_ms += q[1]; tot_fl += q[2]
                if kind in (0, 1, 2, 6, 7):
                    step_fl += q[2]
            # the 32 -> 32 plane layers (level 0) are counted in the dominant-kernel totals AND priced on their own against
            # the HBM roofline (8 flop per algorithmic byte): algorithmic bytes / HIP-event time
            c32 = {"launches": kinds["conv_fwd_32planes"]["launches"] + kinds["conv_bwd_input_32planes"]["launches"],
                   "ms": kinds["conv_fwd_32planes"]["ms"] + kinds["conv_bwd_input_32planes"]["ms"],
                   "gbyte": kinds["conv_fwd_32planes"]["gbyte"] + kinds["conv_bwd_input_32planes"]["gbyte"]}
            for k2 in ("conv_fwd", "conv_bwd_input"):     # the headline per-kind entries keep meaning "all forward / dX launches"
                for f2 in ("launches", "ms", "gflop", "gbyte"):
                    kinds[k2][f2] += kinds[k2 + "_32planes"][f2]
            if tot_ms > 0:
                ach = tot_fl / (tot_ms * 1e-3) / 1e12
                # HBM traffic cannot be read live: it comes from the committed PMC passes of this same command
                # (tools/traffic.sh); the file's git blob hash is emitted beside it so that a stale file shows
                traffic = traffic_src = None
                import glob
                found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r??_conv_fwd_hbm_traffic.json")))
                tname = os.path.basename(found[-1]) if found else "none"   # the latest round's passes
                tpath = os.path.join(ROOT, "profiles", tname)
                if (os.path.exists(tpath) and args.model == "dune3d" and args.mode == "train" and args.batch == 64
                        and args.voxels == 50000 and args.kind == "tracks" and grid == (900, 900, 900)):
                    import hashlib
                    raw = open(tpath, "rb").read()
                    tj = json.loads(raw)
                    traffic = tj["traffic_bytes_per_launch"]
                    traffic_src = {"file": "profiles/" + tname, "git_blob": hashlib.sha1(b"blob %d\0" % len(raw) + raw).hexdigest(),
                                   "measured_at_commit": tj.get("commit")}
                roof = {"bound": "mfma",
                        "kernel": "conv_round_mfma_kernel + conv_mround_mfma_kernel + conv_tile_mfma_kernel + conv_wave_mfma_kernel (" +
                                  ("all FORWARD launches; the dX launches of the same kernels run beside the dW backlog of the "
                                   "side stream by design -- their event durations measure the sharing, see whole_step"
                                   if shared else "all forward + dX launches") + ")",
                        "achieved": ach, "peak": 157.3, "unit": "TFLOP/s", "frac": ach / 157.3, "traffic": traffic,
                        "traffic_source": traffic_src,
                        "traffic_note": "HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes "
                                        "(tools/traffic.sh; counter collection serialises the kernels; average over the forward "
                                        "and dX launches, which mirror each other); algorithmic bytes per launch %.3g"
                                        % ((kinds["conv_fwd"]["gbyte"] + kinds["conv_bwd_input"]["gbyte"]) * 1e9
                                           / max(kinds["conv_fwd"]["launches"] + kinds["conv_bwd_input"]["launches"], 1)),
                        "launches_per_step": tot_n / args.steps, "avg_launch_ms": tot_ms / max(tot_n, 1),
                        # every convolution flop of the step (forward + dX + dW, 2*R*Cin*Cout each) over the STEP's wall time:
                        # a lower bound of the MFMA fraction that no overlap of kernels can flatter
                        "whole_step": {"algorithmic_tflop": step_fl / args.steps / 1e12,
                                       "achieved": step_fl / args.steps / 1e12 / (ms * 1e-3), "unit": "TFLOP/s",
                                       "frac": step_fl / args.steps / 1e12 / (ms * 1e-3) / 157.3},
                        "per_kind_note": ("kinds on different streams overlap (dW and rulebook_build run on side streams): "
                                          "their event durations do not add up to the step" if (shared or prefetch) else None),
                        "per_kind_per_step": {k: {kk: vv / args.steps for kk, vv in v.items()} for k, v in kinds.items()}}
                if c32["ms"] > 0:
                    roof["hbm_view_32plane_layers"] = {
                        "bound": "hbm", "kernel": "conv_wave_mfma_kernel<2,2> (32 -> 32 planes, forward + dX launches)",
                        "achieved": c32["gbyte"] / (c32["ms"] * 1e-3) / 1e3, "peak": 8.0, "unit": "TB/s",
                        "frac": c32["gbyte"] / (c32["ms"] * 1e-3) / 1e3 / 8.0,
                        "launches_per_step": c32["launches"] / args.steps, "avg_launch_ms": c32["ms"] / max(c32["launches"], 1)}
        cpu = None
        if world == 1 and args.cpu_events > 0 and args.model == "dune3d":
            cpu = cpu_baseline(arr, labels, image_size, min(args.cpu_events, args.batch), args.mode == "fwd",
                               args.cpu_threads, fwd_training=args.fwd_bn == "train")
        out = {
            "metric": ("events/sec (fwd+bwd) DUNE-3D sparse ResNet, 64-event batch, 1/2/4/8 MI355X"
                       if args.mode == "train" else "events/sec (fwd only) DUNE-3D sparse ResNet")
                      if args.model == "dune3d" else "events/sec (%s) SBND 2-D multiplane sparse ResNet"
                      % ("fwd+bwd" if args.mode == "train" else "fwd only"),
            "value": args.batch * world / (ms * 1e-3), "unit": "events/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.product_mode == "fp32" else "f32 (products as 3 x bf16 splits, fp32 accumulation)",
            "data": "synthetic",
            "config": {"workload": "%s %s, %d events/GPU x %d voxels in %dx%dx%d, %s generator"
                                   % ("dune3d sparse ResNet (depth 5, 32 filters, 4 res-blocks/level)"
                                      if args.model == "dune3d" else
                                      "sbnd 2-D multiplane shared-weight sparse ResNet (depth 6, pre-merge 3, 32 filters, "
                                      "2 res-blocks/level; voxels = pixels per plane)",
                                      "fwd+bwd+Adam" if args.mode == "train" else
                                      ("fwd (inference: eval-mode BatchNorm)" if args.fwd_bn == "eval" else "fwd (batch-statistics BatchNorm)"),
                                      args.batch, args.voxels,
                                      grid[0], grid[1], grid[2], args.kind),
                       "global_batch": args.batch * world, "parallelism": "dp%d" % world,
                       "backward": {0: "dX and dW back to back", 1: "dX and dW of a layer on two streams, joined per layer",
                                    2: "dW on a side stream, joined once per backward pass"}[args.overlap_backward if args.mode == "train" else 0],
                       "rulebooks": ("step i+1's InputLayer hash + rulebooks built on a side stream while step i computes (once per step, inside the timed region)"
                                     if prefetch else
                                     ("every tiled variant of the batch built at the top of the forward pass from the previous "
                                      "step's list, one count read-back for all (Trainer.batch_rulebooks); inside the timed region"
                                      if tr.batch_rulebooks else "built lazily in the forward pass"))},
            "roofline": roof, "cpu_baseline": cpu,
            "hbm_peak_gb": {"torch_allocator": torch.cuda.max_memory_allocated() / 1e9,
                            "torch_reserved": torch.cuda.max_memory_reserved() / 1e9,
                            # the library's own device pool (hash tables, rulebooks, scratch): bytes held at the end
                            "library_pool_held": pool_held / 1e9},
        }
        out["batches"] = {
            "distinct_batches_cycled": nb, "seeds": "1000003*rank + 64*b + e, b = 0..%d" % (nb - 1),
            # the same K steps on batch 0 alone (identical sizes every step: the quantity rounds 1-4 reported)
            "single_batch_repeat": None if single is None else {
                "ms_per_step": single, "events_per_s": args.batch * world / (single * 1e-3),
                # what explains the difference: the batches hold different amounts of work (rules x planes)
                "conv_tflop_per_step": single_tflop, "conv_tflop_per_step_cycled": cycled_tflop,
                "ms_per_conv_tflop": None if not single_tflop else single / single_tflop,
                "ms_per_conv_tflop_cycled": None if not cycled_tflop else ms / cycled_tflop},
            # bytes the library's device pool holds after the warm-up steps and after the timed loop: a pool that still
            # grows while the batches repeat would show here (warm-up must have seen every batch: warmup >= batches)
            "library_pool_held_after_warmup": pool_warm, "library_pool_held_after_timed_loop": pool_end,
            "library_pool_stopped_growing": (pool_end <= pool_warm) if args.warmup >= nb else None}
        if args.warmup >= nb and pool_end > pool_warm:
            print("bench.py: WARNING the library's device pool grew by %.1f MB during the timed loop" % ((pool_end - pool_warm) / 1e6),
                  file=sys.stderr)
        if optin is not None:
            optin["events_per_s"] = args.batch * world / (optin["ms_per_step"] * 1e-3)
            optin["what"] = ("OPT-IN, never `value`: scn.set_product_mode('bf16x3') -- every fp32 operand of the forward / dX convolutions as "
                             "an exact sum of three bf16 numbers, the six largest part products on the bf16 matrix pipe, fp32 accumulation "
                             "(per product within 2^-23 of the exact one); dW, BatchNorm and everything else unchanged; same cycled loop")
            out["optin_bf16x3_products"] = optin
        out["config"]["products"] = ("fp32 matrix instructions (v_mfma_f32_16x16x4_f32)" if args.product_mode == "fp32" else
                                     "OPT-IN three-way bf16 splits on v_mfma_f32_16x16x32_bf16, fp32 accumulation")
        if sync is not None:
            out["data_parallel_checks"] = sync
        if staging is not None:
            out["staging_inclusive"] = staging
        print(json.dumps(out), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
