"""Training step of the hot path's caller, without Lightning:
``supervised_eventID.forward / training_step / calculate_loss / configure_optimizers``
(/root/reference/src/utils/supervised_eventID.py:53-91,168-207) and
``init_optimizer`` Adam(lr=1.0 x schedule, eps=1e-6, betas=(0.8,0.9), weight_decay)
(/root/reference/src/utils/training_utils.py:12-13), LR schedule "standard"
(/root/reference/src/utils/learning_rate_scheduler.py:92-126).
"""
import math

import numpy as np
import torch

from . import scn
from .config import IMAGE_SIZE, OUTPUT_SHAPE, Config
from .data import larcvsparse_to_scnsparse_2d, larcvsparse_to_scnsparse_3d
from .distributed import FlatArena, FlatBuffers
from .loss import accuracy, focal_loss
from .networks import build_networks


class WarmupFlatDecay:
    """learning_rate_scheduler.py:92-126: linear warm-up (1 epoch) -> flat -> exponential decay."""

    def __init__(self, peak_learning_rate=3e-3, decay_floor=1e-3, epoch_length=100, decay_epochs=5, total_epochs=25):
        self.peak, self.floor, self.epoch_length = peak_learning_rate, decay_floor, epoch_length
        if total_epochs is None:
            total_epochs = 25
        self.flat_len = (total_epochs - decay_epochs - 1) * epoch_length
        self.decay_len = decay_epochs * epoch_length

    def __getitem__(self, idx):
        if idx < self.epoch_length:
            return 0.00001 + idx * (self.peak - 0.00001) / self.epoch_length
        idx -= self.epoch_length
        if idx < self.flat_len:
            return self.peak
        idx -= self.flat_len
        if idx < self.decay_len:
            return (self.peak - self.floor) * math.exp(-0.01 * idx) + self.floor
        return 0.0

    __call__ = __getitem__


class Prefetched:
    """An input batch whose InputLayer and rulebooks were built ahead of time on a side stream (Trainer.prefetch).
    Holds the input tensors the side stream reads, so that a caller who drops the staged batch cannot free them early."""

    def __init__(self, tensor, event, inputs=()):
        self.tensor, self.event, self.inputs = tensor, event, inputs


class Trainer:
    """Owns encoder + heads (reference topology), the flat parameter/gradient arenas and Adam."""

    def __init__(self, cfg=None, image_size=None, output_shape=None, weight_decay=1e-6, lr_schedule=None,
                 device="cuda", seed=0, model="resnet", sbnd_cfg=None, fused_head=False):
        self.cfg = cfg or Config()
        self.model = model
        self.fused_head = fused_head and model != "sbnd"   # opt-in: skips the dense tensor
        torch.manual_seed(seed)
        if model == "sbnd":
            # legacy multiplane network (src/networks/torch/sparseresnet.py): logits come straight out of it
            from .config import Data
            from .networks_sbnd import ResNet, SBNDNetworkConfig
            self.cfg = Config(data=Data(dimension=2))
            sbnd_cfg = sbnd_cfg or SBNDNetworkConfig()
            self.image_size = list(sbnd_cfg.spatial_size)
            shape = output_shape or OUTPUT_SHAPE
            self.encoder = ResNet({k: [v] for k, v in shape.items()}, sbnd_cfg)
            self.head = torch.nn.Identity()
        else:
            self.image_size = image_size or IMAGE_SIZE["dune3d" if self.cfg.data.dimension == 3 else "dune2d"]
            self.encoder, self.head = build_networks(self.cfg, self.image_size, output_shape or OUTPUT_SHAPE)
        self.encoder.to(device)
        self.head.to(device)
        self.arena = FlatArena([self.encoder, self.head])
        self.buffers = FlatBuffers([self.encoder, self.head])
        # weight_decay default: mode.optimizer.weight_decay = 1e-6 (src/config/optimizer.py:52, the value
        # supervised_eventID.py:201 hands to init_optimizer)
        self.weight_decay = weight_decay
        self._build_optimizer(0)
        self.lr_schedule = lr_schedule or WarmupFlatDecay()
        self.global_step = 0
        self.device = device
        self.batch_rulebooks = True   # build a batch's rulebooks up front from the previous batch's list (one read-back)
        self._side = None        # stream of Trainer.prefetch
        self._ahead = None       # main-stream event recorded at the previous prefetch() call (bounds the host's run-ahead)
        self._template = None    # metadata of the last batch that went through forward(): what prefetch() rebuilds

    def _build_optimizer(self, start):
        """Adam over arena[start:] as ONE flat tensor == Adam on every parameter in it (elementwise), one fused
        kernel.  start > 0 leaves the leading (encoder) part of the arena out of the optimizer entirely: frozen
        parameters are skipped as in the reference, where their grad is None (create_trainer.py:103-106) --
        merely zeroing their gradient would still let weight decay shrink them (update ~ lr*wd*p/eps)."""
        self.opt_start = start
        self.flat_param = torch.nn.Parameter(self.arena.flat[start:], requires_grad=True)
        self.flat_param.grad = self.arena.grad[start:]
        self.opt = torch.optim.Adam([self.flat_param], lr=1.0, eps=1e-6, betas=(0.8, 0.9),
                                    weight_decay=self.weight_decay, fused=torch.cuda.is_available())

    def to_scn_input(self, larcv_array):
        if self.cfg.data.dimension == 3:
            return larcvsparse_to_scnsparse_3d(larcv_array)
        return larcvsparse_to_scnsparse_2d(larcv_array)

    def stage(self, scn_input, labels):
        """Host tuple -> device tensors (the Lightning 'batch -> device' step)."""
        coords, feats, bs = scn_input
        c = torch.as_tensor(np.asarray(coords)).to(self.device).to(torch.int32)
        f = torch.as_tensor(np.asarray(feats), dtype=torch.float32).to(self.device)
        lab = {k: torch.as_tensor(v).to(self.device) for k, v in labels.items()}
        return (c, f, bs), lab

    def stage_larcv(self, larcv_batch, labels):
        """Device-side input transform (src/io/data_transforms.py:21-49,198-253 on the GPU): the larcv
        fp32 block [B,P,M,D+1] crosses PCIe as it is and is compacted / cast / keyed by ``sei_larcv_to_scn`` --
        no float64 host round trip, 16 B per voxel over the link instead of 40."""
        from . import scn
        x = larcv_batch if isinstance(larcv_batch, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(larcv_batch, dtype=np.float32))
        c, f, bs = scn.larcv_to_scn_device(x.to(self.device, non_blocking=True))
        lab = {k: torch.as_tensor(v).to(self.device, non_blocking=True) for k, v in labels.items()}
        return (c, f, bs), lab

    def prefetch(self, image, after=None):
        """Builds the NEXT batch's InputLayer hash and every rulebook the last forward pass used on a side stream, ahead
        of the forward pass that needs them (the reference builds them lazily inside it, resnet.py:141-161 -> SCN
        Metadata), and hands the result to the next ``train_step``.  Same kernels, same results (every kernel is
        deterministic); only WHERE and WHEN the integer work and its count read-backs run changes.

        ``after``: a ``torch.cuda.Event`` behind which the input tensors of ``image`` are complete (recorded by the
        caller after staging them).  The side stream then waits for that event ONLY, and the build overlaps with the
        step still computing on the main stream: hash probes and rule compaction are latency-bound integer kernels
        that fit beside the MFMA-bound convolutions, and their count read-backs block the host on the side stream while
        the main stream keeps draining its queue.  Device-pool blocks need no ordering from the caller: the pool
        releases a block behind one event per stream the library has seen (capi.hip, quarantine).  ``after=None`` is the
        conservative form -- the side stream waits for everything queued on the main stream so far, i.e. no overlap with
        the running step (only the placement of the work changes)."""
        if isinstance(image, Prefetched) or not torch.cuda.is_available():
            return image
        if self._side is None:
            self._side = torch.cuda.Stream()   # default priority (a high-priority build stream measured no different)
        main = torch.cuda.current_stream()
        # The side stream waits for nothing the main stream does, and the host blocks only on the side stream's count
        # read-backs: left alone, a loop that never synchronises queues step after step ahead of the GPU, each with its
        # batch's hash tables and rulebooks alive in the library's pool (bench.py: +9.5 GB over ten steps).  One batch
        # ahead is all the overlap needs: before the build for step j + 2 starts, the host waits for the END of step j
        # (the step in front of the one just queued); the GPU still holds a whole queued step, so it never idles.
        mark = torch.cuda.Event()
        mark.record(main)
        if self._ahead is not None:
            self._ahead.synchronize()
        self._ahead = mark
        if after is None:
            self._side.wait_stream(main)
        else:
            self._side.wait_event(after)
        inputs = tuple(x for x in image[:2] if isinstance(x, torch.Tensor) and x.is_cuda)
        with torch.cuda.stream(self._side):
            layer = self.encoder.input_tensor if self.model == "sbnd" else self.encoder.input_layer
            layer.template = self._template if self.batch_rulebooks else None   # input level + strided levels: one read-back
            try:
                t = layer(image)
            finally:
                layer.template = None   # the layer must not keep a batch's metadata alive, nor carry it into a deepcopy / pickle
            for x in inputs:
                x.record_stream(self._side)   # read by the side stream's kernels: not reusable before they are done
            try:
                t.metadata.prebuild_like(self._template)
            except RuntimeError as e:
                # a level of the template does not exist for this batch (tiny event): the forward builds lazily.
                # Anything else (out of memory, hash table full) is a real failure.
                if "no level with this spatial size" not in str(e) and "smaller than filter" not in str(e):
                    raise
            ev = torch.cuda.Event()
            ev.record(self._side)
        return Prefetched(t, ev, inputs)

    def forward(self, image):
        layer = self.encoder.input_tensor if self.model == "sbnd" else self.encoder.input_layer
        if isinstance(image, Prefetched):
            main = torch.cuda.current_stream()
            main.wait_event(image.event)
            image.tensor.features.record_stream(main)   # allocated on the side stream, consumed here
            image = image.tensor
        else:
            # with the previous batch's metadata as the template the InputLayer discovers the strided levels together with
            # the input level (scn.InputLayer.template -> sei_input_build_like: ONE count read-back for all of them)
            layer.template = self._template if self.batch_rulebooks else None
            try:
                image = layer(image)     # Encoder.forward / ResNet.forward pass a prepared tensor through
            finally:
                layer.template = None
            if self.batch_rulebooks and self._template is not None and image.features.is_cuda:
                # every rulebook (forward AND backward variants) the previous batch needed, built now with ONE count
                # read-back for all of them instead of one per variant spread over the forward and backward passes;
                # a pass that will not differentiate (eval_step, validation under no_grad) builds the forward variants only
                try:
                    image.metadata.prebuild_like(self._template, forward_only=not torch.is_grad_enabled())
                except RuntimeError as e:
                    if "no level with this spatial size" not in str(e) and "smaller than filter" not in str(e):
                        raise
        # the template lists what a TRAINING step walks: a no-grad pass (forward variants only) does not replace it
        if torch.is_grad_enabled() or self._template is None:
            self._template = image.metadata
        if self.fused_head:
            return self.head.forward_pooled(self.encoder.forward_pooled(image))
        return self.head(self.encoder(image))

    def train_step(self, image, labels):
        """forward + focal loss + backward + gradient all-reduce + Adam step. Returns loss tensor."""
        self.arena.zero_grad()
        self.arena.detach_grads()  # autograd assigns the gradients; one multi-tensor copy files them into the arena
        self.buffers.broadcast()   # DDP broadcast_buffers=True: rank 0's BN running statistics before every forward
        logits = self.forward(image)
        loss = focal_loss(logits, labels)
        loss.backward()
        # mode 2 of scn.set_backward_streams: the dW kernels still queued on the side stream (and a check that the
        # gradients autograd kept are the buffers those kernels wrote)
        scn.backward_join(self.arena.params)
        self.arena.gather_grads()
        self.arena.all_reduce_mean(self.opt_start)
        for g in self.opt.param_groups:
            g["lr"] = self.lr_schedule[self.global_step]
        self.opt.step()
        self.global_step += 1
        return loss.detach(), logits

    @torch.no_grad()
    def eval_step(self, image, labels=None):
        self.encoder.eval(); self.head.eval()
        logits = self.forward(image)
        self.encoder.train(); self.head.train()
        if labels is None:
            return logits
        return logits, accuracy(logits, labels)


    # ------------------------------------------------------------------ checkpoint / resume
    # state_dict-compatible checkpoints: {"state_dict": {"encoder.*", "head.*"}, "global_step", "optimizer"}.  The
    # "state_dict" / "global_step" entries are what the reference reads back (create_trainer.py:93-115: it loads the
    # ``encoder.*`` keys of ckpt["state_dict"]); Lightning's own "optimizer_states" / "lr_schedulers" / "epoch" / "loops"
    # entries are written only by save_checkpoint(lightning_layout=True) ("optimizer_states" = the flat Adam moments sliced
    # per parameter, "lr_schedulers", "epoch"; never "callbacks" / "loops"); "optimizer" holds the Adam state of this
    # Trainer's ONE flat parameter.  load_checkpoint() of a real Lightning file restores weights, step and -- from its
    # per-parameter "optimizer_states" -- the moments.
    def state_dict(self):
        sd = {"encoder." + k: v for k, v in self.encoder.state_dict().items()}
        sd.update({"head." + k: v for k, v in self.head.state_dict().items()})
        return sd

    def optimizer_state_per_parameter(self):
        """The Adam state in ``torch.optim.Adam(model.parameters()).state_dict()`` layout -- parameter i of
        [encoder.parameters() + head.parameters()] -> {"step", "exp_avg", "exp_avg_sq"} -- i.e. what a Lightning
        checkpoint holds in ``optimizer_states[0]`` for the reference's optimizer (supervised_eventID.py:198-207):
        slices of this Trainer's ONE flat moment pair.  Frozen parameters (before opt_start) carry no state, as
        parameters without a gradient do in torch."""
        own = self.opt.state_dict()
        flat = own["state"].get(0)
        state = {}
        if flat is not None:
            for i, (p, (off, n)) in enumerate(zip(self.arena.params, self.arena.spans)):
                if off < self.opt_start:
                    continue
                a = off - self.opt_start
                state[i] = {"step": flat["step"].detach().clone().cpu(),
                            "exp_avg": flat["exp_avg"][a:a + n].detach().view(p.shape).clone().cpu(),
                            "exp_avg_sq": flat["exp_avg_sq"][a:a + n].detach().view(p.shape).clone().cpu()}
        group = {k: v for k, v in own["param_groups"][0].items() if k != "params"}
        group["fused"] = None
        group["params"] = list(range(len(self.arena.params)))
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_per_parameter(self, sd):
        """Inverse of ``optimizer_state_per_parameter``: per-parameter Adam moments (a Lightning ``optimizer_states[0]``
        of the reference's run, or this Trainer's own export) -> the flat moment pair.  All parameters must be at the
        same step (they are: one optimizer, stepped together)."""
        per = sd.get("state", {})
        if not per:
            import warnings
            warnings.warn("the checkpoint's optimizer_states hold no moments: weights and global_step were restored, the "
                          "optimizer starts from zero moments")
            return
        per = {int(k): v for k, v in per.items()}
        steps = {float(torch.as_tensor(v["step"]).item()) for v in per.values()}
        if len(steps) != 1:
            raise ValueError("per-parameter optimizer state at different steps %s: not one Adam stepped together" % sorted(steps))
        dev = self.flat_param.device
        m = torch.zeros_like(self.flat_param.data)
        v2 = torch.zeros_like(self.flat_param.data)
        for i, (p, (off, n)) in enumerate(zip(self.arena.params, self.arena.spans)):
            if i not in per:
                continue
            if off < self.opt_start:
                raise ValueError("checkpoint holds optimizer state for parameter %d, which is frozen in this trainer" % i)
            for name, dst in (("exp_avg", m), ("exp_avg_sq", v2)):
                t = per[i][name]
                if tuple(t.shape) != tuple(p.shape):
                    raise ValueError("optimizer state %s of parameter %d has shape %s, the parameter %s"
                                     % (name, i, tuple(t.shape), tuple(p.shape)))
                a = off - self.opt_start
                dst[a:a + n].copy_(t.reshape(-1).to(dev))
        own = self.opt.state_dict()
        own["state"] = {0: {"step": torch.tensor(steps.pop(), dtype=torch.float32), "exp_avg": m, "exp_avg_sq": v2}}
        self.opt.load_state_dict(own)
        self.flat_param.grad = self.arena.grad[self.opt_start:]

    def save_checkpoint(self, path, lightning_layout=False):
        """``lightning_layout=True`` adds the keys a Lightning ``fit(ckpt_path=...)`` resume reads for the reference's
        single Adam + LambdaLR pair (create_trainer.py:83-115): ``optimizer_states`` (per-parameter moments, see
        ``optimizer_state_per_parameter``), ``lr_schedulers`` (step counters of the LambdaLR the reference wraps its
        schedule in) and ``epoch``.  The per-parameter optimizer state is checked against a plain ``torch.optim.Adam`` over
        the same parameters (tests/test_host.py); Lightning itself is not installed here, so its loader was never run on
        these files: ``callbacks`` / ``loops`` are not written."""
        import os
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        ck = {"state_dict": {k: v.detach().cpu() for k, v in self.state_dict().items()},
              "global_step": self.global_step, "optimizer": self.opt.state_dict(),
              "opt_start": self.opt_start, "arena_numel": int(self.arena.numel)}
        if lightning_layout:
            step = int(self.global_step)
            lr = float(self.lr_schedule[step])
            ck["epoch"] = 0
            ck["optimizer_states"] = [self.optimizer_state_per_parameter()]
            ck["lr_schedulers"] = [{"base_lrs": [1.0], "last_epoch": step, "_step_count": step + 1, "_last_lr": [lr],
                                    "lr_lambdas": [None]}]
        torch.save(ck, path)

    def _load_into(self, module, prefix, state):
        sub = {k[len(prefix):]: v for k, v in state.items() if k.startswith(prefix)}
        own = module.state_dict()
        missing = [k for k in own if k not in sub]
        if missing:
            raise KeyError("checkpoint lacks %s%s" % (prefix, missing[0]))
        with torch.no_grad():                        # copy IN PLACE: parameters are views of the flat arena
            for k, t in own.items():
                if tuple(sub[k].shape) != tuple(t.shape):    # copy_ would silently broadcast
                    raise ValueError("checkpoint tensor %s%s has shape %s, the module expects %s"
                                     % (prefix, k, tuple(sub[k].shape), tuple(t.shape)))
                t.copy_(sub[k].to(t.device))

    def load_checkpoint(self, path, resume=True):
        """Auto-resume (create_trainer.py:110-115): weights, step counter and optimizer state."""
        ck = torch.load(path, map_location="cpu")
        self._load_into(self.encoder, "encoder.", ck["state_dict"])
        self._load_into(self.head, "head.", ck["state_dict"])
        if resume:
            self.global_step = int(ck.get("global_step", 0))
            if "optimizer" in ck:
                # the optimizer state belongs to ONE flat parameter = arena[opt_start:]: a checkpoint written with another
                # arena layout (alignment padding, a frozen encoder) must not be loaded into it silently
                want = int(self.flat_param.numel())
                for st in ck["optimizer"].get("state", {}).values():
                    for k in ("exp_avg", "exp_avg_sq"):
                        if k in st and int(st[k].numel()) != want:
                            raise ValueError("checkpoint optimizer state has %d elements, this trainer optimises %d "
                                             "(arena layout / frozen-encoder setting differ: saved opt_start=%s, here %d); "
                                             "load with resume=False to take the weights only"
                                             % (int(st[k].numel()), want, ck.get("opt_start", "?"), self.opt_start))
                self.opt.load_state_dict(ck["optimizer"])
                self.flat_param.grad = self.arena.grad[self.opt_start:]
            elif ck.get("optimizer_states"):
                # a Lightning checkpoint of the reference's run: per-parameter Adam moments -> the flat pair
                self.load_optimizer_state_per_parameter(ck["optimizer_states"][0])

    def restore_encoder_only(self, path, freeze=True):
        """mode.weights_location + restore_encoder_only (create_trainer.py:93-106): load the ``encoder.*`` keys
        and freeze the encoder."""
        ck = torch.load(path, map_location="cpu")
        self._load_into(self.encoder, "encoder.", ck["state_dict"])
        if freeze:
            for p in self.encoder.parameters():
                p.requires_grad_(False)
            # the encoder's parameters come first in the arena: optimise (and all-reduce) only what follows them
            self._build_optimizer(self.arena.offset_after(self.encoder))


def format_log_message(metrics, batch_size, global_step, dt=None, mode=""):
    """training_utils.format_log_message (src/utils/training_utils.py:31-60): 'key: value' list + Img/s."""
    s = ", ".join("{0}: {1:.3}".format(k, float(v)) for k, v in metrics.items())
    if dt is not None and dt > 0:
        s += " ({:.2}s / {:.2} Img/s)".format(dt, batch_size / dt)
    return "{} Step {} metrics: {}".format(mode, global_step, s)
