"""Trainer core: the reference's ``Model`` base class and step loop (common/trainer.py:25-407) on the HIP path.

Same subclass contract (hooks ``extract_latents``, ``extract_embeddings``, ``validate``, ``optimize(ratio, latents,
embeddings, repa_tokens, generator)``, ``save_model``; attributes ``model``, ``scheduler``, ``aspect_ratios``,
``params``, ``accelerator``) and the same step semantics (per-batch CFG dropout :319-323, fresh generator per step
:325, exploration steps :326-336, accumulate/no_sync :317, clip 1.0 :347, AdamW :348, EMA :350-351, LR warm-up
:255-262,353-354, zero_grad :356, loss = cross-rank mean of the SUM of micro-losses :343,359-360, validate/save
cadence :371-401).

What is different (MI355X-first, documented in DESIGN.md):
* ``HipAccelerator`` replaces HuggingFace Accelerate: one process per GPU from the launcher's env, RCCL process group
  with xGMI P2P left ENABLED (the reference disables it, :27-28 -- right for its dual consumer GPUs, wrong here);
* DDP is ``yat_amd.ddp.HipDDP`` (flat bucketed all-reduce overlapped with backward), clip+AdamW+EMA+zero_grad are
  one fused launch pair, so ``clip_grad_norm_`` is a no-op hook here;
* logging (``tb_writer.SummaryWriter``, TensorBoard event files) keeps the loss on the device and writes the
  scalars YAT_LOG_FLUSH steps at a time, so there is no per-step ``.item()`` stall (:364).
PEFT adapters (:212-241) are built -- LoRA, DoRA, LoKr, LoHa through ``wrap_adapters`` (yat_amd/lora.py, dora.py, lokr.py,
loha.py); FourierFT is refused.  Dual-GPU mode, Dreambooth, REPA and DeepSpeed are out of scope.
"""
from __future__ import annotations

import collections
import contextlib
import os
import random
from datetime import timedelta

import torch
import torch.distributed as dist

from ..ddp import HipDDP
from .host import cap_host_threads
from ..optim import FlatAdamW
from .aspect_ratios import ASPECT_RATIO_1024_BIN, ASPECT_RATIO_512_BIN


@contextlib.contextmanager
def rescale_adapter_scale(adapters, multiplier):
    """``peft.helpers.rescale_adapter_scale(model, multiplier)`` [RECALL: a ``@contextmanager``]: inside the ``with`` block every
    adapter's scaling is multiplied by ``multiplier``; the original scaling comes back on exit.  ``adapters`` is this build's
    adapter set (yat_amd/lora.py, lokr.py, loha.py, dora.py: they read ``.scale`` at every launch; LoRA also keeps it in a
    device vector for the GEMM epilogue).  Like peft it refuses a model without adapters and a non-numeric multiplier."""
    if not isinstance(multiplier, (float, int)):
        raise TypeError(f"Argument multiplier should be of type float, got {type(multiplier)}")
    if adapters is None or not hasattr(adapters, "scale"):
        raise ValueError("scaling is only supported for models with adapters")
    original = adapters.scale

    def put(v):
        adapters.scale = v
        gate = getattr(adapters, "_gate", None)
        if gate is not None:
            gate.fill_(v)
        # a recorded launch plan holds the scalar arguments of its launches (ops.Recorder stores [fn, args]): none may
        # outlive a change of the scale.  (Today no plan is recorded while adapters are attached -- FlatParamModule.planned
        # -- so this is a guard for the day one is, round-4 advisor.)
        for owner in (adapters, getattr(adapters, "model", None)):
            plans = getattr(owner, "_plans", None)
            if plans:
                plans.clear()
    put(original * multiplier)
    try:
        yield
    finally:
        put(original)


class HipAccelerator:
    """The slice of accelerate.Accelerator the reference trainer touches, natively."""

    def __init__(self, gradient_accumulation_steps=1, device=None, backend=None, timeout_s=3600):
        self.gradient_accumulation_steps = int(gradient_accumulation_steps or 1)
        cap_host_threads()                               # (common/host.py: the OpenMP pool vs the CPUs this job really has)
        self.process_index = int(os.environ.get("RANK", "0"))
        self.num_processes = int(os.environ.get("WORLD_SIZE", "1"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if device is None:
            if torch.cuda.is_available():
                from ..ddp import forced_backend
                if (forced_backend() or "nccl") != "nccl":                    # one-GPU rehearsal: ranks share the devices present
                    local %= max(torch.cuda.device_count(), 1)
                device = torch.device("cuda", local)
            else:
                device = torch.device("cpu")
        self.device = torch.device(device)
        if self.device.type == "cuda":
            torch.cuda.set_device(self.device)
        if self.num_processes > 1 and not dist.is_initialized():
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            # backend of the launcher-level group (yat_amd/ddp.py group_backend): nccl (= RCCL) on a GPU; gloo when the
            # gradients were asked to travel through the library's own communicator (YAT_COMM=native) -- the group is then
            # only the rendezvous / barrier / consensus channel, no second RCCL communicator beside the library's
            from ..ddp import apply_channel_policy, group_backend
            backend = backend or group_backend(on_gpu=self.device.type == "cuda")
            apply_channel_policy(self.num_processes)     # (before any RCCL communicator exists)
            dist.init_process_group(backend, timeout=timedelta(seconds=timeout_s))
        self.is_main_process = self.process_index == 0
        self.sync_gradients = True
        self._micro = 0
        self.ddp = None
        self.state = type("State", (), {"deepspeed_plugin": None})()

    # -- accelerate surface
    def prepare(self, *objs):
        for o in objs:
            if hasattr(o, "flat_grad") and hasattr(o, "bucket_bounds"):
                self.ddp = HipDDP(o)
                self.ddp.broadcast_parameters()
        return objs if len(objs) != 1 else objs[0]

    def unwrap_model(self, model):
        return model

    @contextlib.contextmanager
    def accumulate(self, model):
        self._micro += 1
        self.sync_gradients = (self._micro % self.gradient_accumulation_steps) == 0
        model.accumulate_grads = (self._micro % self.gradient_accumulation_steps) != 1 and \
            self.gradient_accumulation_steps > 1
        if self.ddp is not None:
            self.ddp.sync = self.sync_gradients
        yield

    def backward(self, loss):
        # a recipe's device path has already run the backward (scaled by 1 / gradient_accumulation_steps) inside optimize()
        if not getattr(loss, "yat_backward_done", False):
            if self.ddp is not None and self.sync_gradients:
                self.ddp.on_loss(loss)                   # (the device path reports it itself: model.loss_ready)
            (loss / self.gradient_accumulation_steps if self.gradient_accumulation_steps > 1 else loss).backward()
        if self.ddp is not None and self.sync_gradients:
            self.ddp.wait()

    def clip_grad_norm_(self, parameters, max_norm):
        return None        # fused into FlatAdamW.step (yat_gradnorm_clip + yat_adamw_step)

    def gather(self, t):
        if self.num_processes == 1:
            return t.reshape(1)
        out = [torch.empty_like(t) for _ in range(self.num_processes)]
        dist.all_gather(out, t)
        return torch.stack(out)

    def track_loss(self, running_sum):
        """Before a micro-step: let its loss (+ ``running_sum`` of the window's earlier ones) ride with the gradient
        all-reduce instead of a collective of its own (yat_amd/ddp.py); ``mean_loss`` then needs no communication."""
        if self.ddp is not None and self.num_processes > 1:
            self.ddp.track_loss(running_sum)

    def mean_loss(self, t):
        """``accelerator.gather(avg_loss).mean()`` (common/trainer.py:359)."""
        if self.ddp is not None and self.ddp.carried_loss is not None:
            v, self.ddp.carried_loss = self.ddp.carried_loss, None
            return v.to(t.dtype) if t.is_floating_point() else v
        return self.gather(t).mean()

    def reduce(self, t, reduction="mean"):
        if self.num_processes > 1:
            dist.all_reduce(t)
            if reduction == "mean":
                t = t / self.num_processes
        return t

    def wait_for_everyone(self):
        if self.num_processes > 1:
            dist.barrier()


def adapter_pair(params):
    """Which arithmetic a LoRA / LoKr adapter set of the trainer uses.  True (the default): the adapter's factored term rides
    in the base GEMM as a second operand pair -- base + adapter accumulated in fp32 and rounded to bf16 ONCE
    (yat_amd/lokr.py ``forward_pair``).  False: peft's own op order under bf16 -- base output, adapter output and their sum
    each rounded (``pre_add`` form; what the reference computes, common/trainer.py:212-241 + peft).  The default is at least
    as close to the fp32 truth in every parity test but it is NOT the reference's rounding flow: a run's losses part from
    the peft-order run at step 2 (profiles/r05_zz_adapter_loss_trajectories.txt).  Switches: config key
    ``lora_fused_pair: false`` or ``YAT_ADAPTER_PAIR=0`` (the environment wins)."""
    env = os.environ.get("YAT_ADAPTER_PAIR")
    if env is not None and env != "":
        return env.strip().lower() not in ("0", "false", "no", "off")
    return bool(getattr(params, "lora_fused_pair", True))


def adapter_arithmetic(adapters):
    """One line for logs and bench lines: which rounding order the adapter set in force computes in."""
    if adapters is None:
        return "none"
    if getattr(adapters, "pair", False):
        return "fused_pair (adapter term inside the base GEMM, fp32 sum rounded once; not peft's rounding order)"
    return "pre_add (peft order: base output, adapter output and their sum each rounded to bf16)"


@contextlib.contextmanager
def _step_stream(dev):
    """Run the step loop on a stream of the compute-stream set (yat_amd/flat.py ``compute_stream``) instead of the default
    stream.  The HIP runtime multiplexes streams onto a handful of hardware queues per priority level, and two streams that
    land on one hardware queue run their kernels strictly one after the other: with the process group's and the copy
    engine's streams around, the default stream shared a queue with the weight-gradient and optimizer streams and the step
    lost 16 ms ("hardware queues").  In a data-parallel job the compute streams are therefore the only
    users of the high-priority level -- four streams, four queues; without a process group nothing changes
    (``flat.isolate_streams``)."""
    from ..flat import compute_stream, isolate_streams
    if torch.device(dev).type != "cuda" or not isolate_streams():
        yield
        return
    prev = torch.cuda.current_stream(dev)
    st = compute_stream(dev)
    st.wait_stream(prev)
    torch.cuda.set_stream(st)
    try:
        yield
    finally:
        prev.wait_stream(st)
        torch.cuda.set_stream(prev)


class Model:
    def __init__(self, params, accelerator: HipAccelerator | None = None):
        # NCCL_P2P_DISABLE / NCCL_IB_DISABLE of the reference (:27-28) are deliberately NOT set: xGMI needs P2P.
        self.accelerator = accelerator or HipAccelerator(params.gradient_accumulation_steps)
        self.params = params
        self.process_index = self.accelerator.process_index
        self.num_processes = self.accelerator.num_processes
        self.timesteps = [int(t) for t in params.timesteps]
        if self.timesteps:                                   # :51-64 timestep whitelist
            def get_timesteps_from_list(batch_size):
                idx = [random.choice(self.timesteps) for _ in range(batch_size)]
                return self.scheduler.timesteps[idx].to(self.accelerator.device)
            self.get_timesteps = get_timesteps_from_list
        # :270-281 -- with a timestep whitelist the samplers get a per-step callback that is MEANT to switch the adapter on for
        # the whitelisted timesteps only.  In the reference the body is ``rescale_adapter_scale(self.model, 1.0 | 0.0)`` as a
        # bare call: peft's helper is a context manager, a bare call builds the manager and drops it, nothing is rescaled --
        # and of the entry points only train_sdxl.py:107 hands the callback to its pipeline at all.  This build keeps the
        # reference's EFFECTIVE behaviour by default (same call sites, same no-op) and does what the code meant with
        # YAT_ADAPTER_RESCALE=1 (the multiplier then stays in force until the next call, as a setter would).
        self._rescale_live = os.environ.get("YAT_ADAPTER_RESCALE", "0") != "0"
        self._rescale_cm = None
        self.validation_step_callback = None
        if self.timesteps:
            def step_callback(pipe, step, timestep, callback_kwargs):
                t = timestep.item() if hasattr(timestep, "item") else timestep
                self._set_adapter_scale(1.0 if t in self.timesteps else 0.0)
                return callback_kwargs
            self.validation_step_callback = step_callback
        # per-rank shard range (:66-84)
        n = params.num_shards
        if getattr(params, "dreambooth_dataset_folder", None) is None and n is not None and n >= self.num_processes:
            per = n // self.num_processes
            self.shard_index_begin = self.process_index * per
            self.shard_index_end = n if self.process_index == self.num_processes - 1 else self.shard_index_begin + per
        else:
            self.shard_index_begin, self.shard_index_end = 0, (n or 0)
        self.global_step = 0
        # :137-138 -- the main process owns a SummaryWriter (runs/<date>_<host>/events.out.tfevents.*); YAT_TENSORBOARD=0
        # turns it off.  Scalars are queued on the device and written LOG_FLUSH steps at a time (no per-step .item()).
        self.logger = None
        if self.accelerator.is_main_process and os.environ.get("YAT_TENSORBOARD", "1") != "0":
            from .tb_writer import SummaryWriter
            self.logger = SummaryWriter()
        self.log_flush = int(os.environ.get("YAT_LOG_FLUSH", "20"))
        self._log_queue = []
        self.sampler = None
        self.optimizer = None
        self.lr_scheduler = None
        self.ema_model = None
        self.empty_embeddings = None

    # ---- hooks of the subclass contract (:93-107,122-124,283-296)
    def extract_latents(self, images):
        raise NotImplementedError

    def extract_embeddings(self, captions):
        raise NotImplementedError

    def format_embeddings(self, embeds):
        pass

    def validate(self):
        raise NotImplementedError

    def optimize(self, ratio, latents, embeddings, repa_tokens, generator):
        raise NotImplementedError

    def enable_efficient_attention(self):
        pass            # attention kernels are always the HIP ones

    def finalize(self):
        pass

    def get_timesteps(self, batch_size):
        """:96-101 -- logit-normal indices from the GLOBAL torch RNG -> scheduler.timesteps."""
        u = torch.sigmoid(torch.normal(mean=0.0, std=1.0, size=(batch_size,)))
        idx = (u * self.scheduler.config.num_train_timesteps).long()
        return self.scheduler.timesteps[idx].to(self.accelerator.device)

    def find_closest_ratio(self, ratio):
        best, dist_ = 0.6, 100
        for r in self.aspect_ratios.keys():
            d = abs(float(r) - ratio)
            if dist_ > d:
                best, dist_ = r, d
        return str(best)

    def load_empty_embeddings(self):
        """The embedding of the empty prompt that whole-batch CFG dropout substitutes (:306-308,319-323).  The reference gets
        it from ``extract_embeddings([''])`` -- a text-encoder pass, outside this build's scope -- so it is read from the
        cache like every other feature: ``empty_embeds.pt`` next to the shards (or in the cwd), holding what that call
        returns (SANA / PixArt: a list with one ``[L, C]`` tensor of the mask-true rows, train_sana.py:84-94).  A subclass
        with a text encoder may still override ``extract_embeddings``; it is tried first."""
        try:
            with torch.no_grad():
                return self.extract_embeddings([""])
        except NotImplementedError:
            pass
        cands = [os.path.join(os.path.dirname(q), "empty_embeds.pt") for q in (self.params.local_shard_paths or [])]
        path = next((c for c in cands + ["empty_embeds.pt"] if os.path.isfile(c)), None)
        if path is None:
            raise FileNotFoundError(
                "train_unconditional_prob > 0 needs the empty-prompt embedding: put `empty_embeds.pt` (the value of "
                "extract_embeddings(['']): a list with one [L, C] tensor) next to the shards or in the working directory "
                f"(looked in: {cands + ['empty_embeds.pt']})")
        emb = torch.load(path, map_location="cpu")
        return self.check_empty_embeddings(emb, path)

    def check_empty_embeddings(self, emb, path):
        """SANA / PixArt layout: a list with one ``[L, C]`` tensor (a bare tensor is wrapped).  Recipes with another
        per-sample embedding layout override this (SD3.5: a ``(prompt [T, C], pooled [P])`` pair, train_sd35.py)."""
        if torch.is_tensor(emb):
            emb = [emb if emb.ndim == 2 else emb[0]]
        if not (isinstance(emb, (list, tuple)) and len(emb) >= 1 and torch.is_tensor(emb[0]) and emb[0].ndim == 2):
            raise ValueError(f"{path}: expected extract_embeddings(['']) = a list with one [L, C] tensor")
        return emb

    def save_model(self):
        if getattr(self, "adapters", None) is not None:               # a PeftModel saves only its adapter
            self.adapters.save_pretrained(f"models/{self.global_step}")
        else:
            self.accelerator.unwrap_model(self.model).save_pretrained(f"models/{self.global_step}")

    def make_optimizer(self, trained):
        """:246-248 + :347-356 -- clip(1.0) + AdamW (+EMA) over the flat parameter buffer of a HIP model / adapter set."""
        p = self.params
        return FlatAdamW(trained, lr=p.learning_rate, weight_decay=p.weight_decay, max_grad_norm=1.0,
                         use_ema=bool(getattr(p, "use_ema", False)), ema_decay=0.999, overlap_update=True)

    def make_sampler(self):
        """Cached-feature sampler over this rank's shard range (the intended path of :165-181)."""
        from .bucket_sampler import BucketSampler
        shards = [f"shard-{i:06d}.tar" for i in range(self.shard_index_begin, self.shard_index_end)]
        # a legacy per-sample cache (cache/{idx}.npy tuples written by the reference's common/cache.py) is used when
        # present and no shard list is configured
        legacy = "cache" if (not self.params.local_shard_paths and os.path.isdir("cache")
                             and any(n.endswith(".npy") for n in os.listdir("cache"))) else None
        return BucketSampler(shards, self.accelerator, self.params.batch_size, model=self,
                             seed=self.params.dataset_seed, local_paths=self.params.local_shard_paths,
                             legacy_cache_dir=legacy)

    # ---- :126-281
    def initialize(self):
        p = self.params
        if p.aspect_ratios == 512:
            self.aspect_ratios = ASPECT_RATIO_512_BIN
        elif p.aspect_ratios == 1024:
            self.aspect_ratios = ASPECT_RATIO_1024_BIN
        self.enable_efficient_attention()
        if self.accelerator.is_main_process:
            os.makedirs("models", exist_ok=True)
        if self.sampler is None:
            self.sampler = self.make_sampler()
        self.adapters = None
        if getattr(p, "lora_rank", None) is not None:                 # :212-241 (get_peft_model)
            algo = getattr(p, "lora_algo", "lora")
            targets, rank, alpha = p.lora_target_modules, p.lora_rank, p.lora_alpha
            drop, rslora, saved = getattr(p, "lora_dropout", 0.0) or 0.0, bool(getattr(p, "lora_use_rslora", False)), None
            dora_saved = False
            if getattr(p, "lora_pretrained", None):
                # :236 PeftModel.from_pretrained(model, path, is_trainable=True): the saved adapter's own config decides
                import json
                from safetensors.torch import load_file
                with open(os.path.join(p.lora_pretrained, "adapter_config.json")) as f:
                    conf = json.load(f)
                algo = {"LORA": "lora", "LOKR": "lokr", "LOHA": "loha"}.get(conf.get("peft_type"))
                if algo is None:
                    raise NotImplementedError(f"lora_pretrained: peft_type {conf.get('peft_type')!r} is not built")
                targets, rank = conf["target_modules"], int(conf["r"])
                alpha = conf["lora_alpha"] if algo == "lora" else conf["alpha"]
                drop = conf.get("lora_dropout" if algo == "lora" else "module_dropout", 0.0) or 0.0
                rslora = bool(conf.get("use_rslora", False))
                dora_saved = bool(conf.get("use_dora", False))
                saved = load_file(os.path.join(p.lora_pretrained, "adapter_model.safetensors"))
            if algo not in ("lokr", "lora", "loha"):
                raise NotImplementedError(f"lora_algo: {algo} is not built (lokr -- BASELINE config 5 --, lora (+ DoRA) and loha are)")
            pair = adapter_pair(p)
            if algo == "lora" and (getattr(p, "lora_use_dora", False) or dora_saved):      # :214-219 with use_dora=True
                from ..dora import DoRAAdapters
                self.adapters = DoRAAdapters(self.model, targets, rank, alpha, dropout=drop, use_rslora=rslora)
            elif algo == "lora":                                      # :214-219
                from ..lora import LoRAAdapters
                self.adapters = LoRAAdapters(self.model, targets, rank, alpha, dropout=drop, use_rslora=rslora,
                                             seed=int(getattr(p, "dataset_seed", 0) or 0), pair=pair)
            elif algo == "loha":                                      # :220-224
                from ..loha import LoHaAdapters
                self.adapters = LoHaAdapters(self.model, targets, rank, alpha, module_dropout=drop)
            else:                                                     # :226-230
                from ..lokr import LoKrAdapters
                self.adapters = LoKrAdapters(self.model, targets, rank, alpha, module_dropout=drop, pair=pair)
            if saved is not None:
                self.adapters.load_state_dict(saved)
            n_ad = self.adapters.num_parameters()
            print(f"adapter arithmetic: {adapter_arithmetic(self.adapters)}")
            print(f"trainable params: {n_ad:,} || all params: {self.model.numel_flat + n_ad:,} || "
                  f"trainable%: {100.0 * n_ad / (self.model.numel_flat + n_ad):.4f}")       # print_trainable_parameters (:239)
        # with adapters only they are trained (the base has no gradients, so AdamW leaves it alone in the reference too)
        trained = self.adapters if self.adapters is not None else self.model
        self.optimizer = self.make_optimizer(trained)
        if getattr(p, "shard_optimizer", False) and self.adapters is None:
            os.environ.setdefault("YAT_SHARD_OPTIMIZER", "1")       # (HipDDP reads it; an adapter set keeps the replicated step)
        self.accelerator.prepare(trained)
        self.lr_scheduler = None
        if getattr(p, "warmup_steps", None) is not None:
            self.lr_scheduler = WarmupLR(self.optimizer, p.warmup_steps)
        self.ema_model = self.optimizer.ema_shadow

    # ---- :298-403
    def run(self, max_steps=None, on_step=None):
        """``on_step(global_step)`` (optional) is called after every optimizer step -- bench.py's clock."""
        p = self.params
        self.initialize()
        dev = self.accelerator.device
        with _step_stream(dev):
            return self._run(p, dev, max_steps, on_step)

    def _run(self, p, dev, max_steps, on_step):
        avg_loss = torch.zeros((), device=dev)
        self.accelerator.wait_for_everyone()
        if self.empty_embeddings is None and p.train_unconditional_prob > 0:       # :306-308
            self.empty_embeddings = self.load_empty_embeddings()
        steps = p.steps if max_steps is None else min(p.steps, max_steps)
        # last logged losses as device scalars (tests and callers peek at it); bounded: an unbounded list of 0-dim device
        # tensors pins one 512-byte allocator block per step for the whole run
        self.loss_history = collections.deque(maxlen=int(os.environ.get("YAT_LOSS_HISTORY", "1024")))
        while self.global_step < steps:
            for batch in self.sampler:
                ratio, latents, embeddings, repa = batch.ratio, batch.vae_features, batch.embeddings, batch.repa_features
                with self.accelerator.accumulate(self.model):
                    if random.random() < p.train_unconditional_prob:          # whole-batch CFG dropout (:319-323)
                        embeddings = [self.empty_embeddings[0] for _ in embeddings]
                    generator = torch.Generator()                             # fresh, unseeded (:325)
                    if p.exploration_steps is not None:                       # :326-336
                        with torch.no_grad():
                            states, losses = [], []
                            for _ in range(p.exploration_steps):
                                states.append(generator.get_state())
                                losses.append(self.optimize(ratio, latents, embeddings, repa, generator))
                        generator.set_state(states[int(torch.argmin(torch.stack(losses)))])
                    self.accelerator.track_loss(avg_loss)
                    loss = self.optimize(ratio, latents, embeddings, repa, generator)
                    avg_loss = avg_loss + loss.detach()
                    self.accelerator.backward(loss)
                    if self.accelerator.sync_gradients:
                        self.accelerator.clip_grad_norm_(None, max_norm=1.0)  # fused below
                        self.optimizer.step()                                 # clip + AdamW + EMA (fused)
                        if self.lr_scheduler is not None:
                            self.lr_scheduler.step()
                if self.accelerator.sync_gradients:
                    mean_loss = self.accelerator.mean_loss(avg_loss)
                    avg_loss = torch.zeros((), device=dev)
                    self.loss_history.append(mean_loss)
                    if self.logger is not None and self.accelerator.is_main_process:
                        lr = self.lr_scheduler.get_last_lr()[0] if self.lr_scheduler is not None else None
                        self._log_queue.append((self.global_step, mean_loss, lr))
                        if len(self._log_queue) >= self.log_flush:
                            self.flush_log()
                    if self.global_step % p.num_steps_per_validation == 0:
                        self.flush_log()
                        self._validate_and_save()
                    self.global_step += 1
                    if on_step is not None:
                        on_step(self.global_step)
                    if self.global_step >= steps:
                        break
        self.flush_log()
        self.finalize()

    def _set_adapter_scale(self, multiplier):
        """The reference's ``rescale_adapter_scale(self.model, multiplier)`` call sites (:274, :276, :388, :390, :397)."""
        cm = rescale_adapter_scale(getattr(self, "adapters", None), multiplier) if getattr(self, "adapters", None) is not None \
            else None
        if not self._rescale_live or cm is None:
            return                                          # the reference: the manager is built and dropped, nothing changes
        if self._rescale_cm is not None:
            self._rescale_cm.__exit__(None, None, None)     # back to the trained scaling before the next multiplier applies
        self._rescale_cm = cm if multiplier != 1.0 else None
        if self._rescale_cm is not None:
            self._rescale_cm.__enter__()

    def flush_log(self):
        """:362-369 -- ``add_scalar('train/loss' | 'train/lr', v, step)``; one device->host copy for the whole queue."""
        queue, self._log_queue = self._log_queue, []
        if not queue or self.logger is None:
            return
        try:
            losses = torch.stack([q[1].float() for q in queue]).tolist()
            for (step, _, lr), v in zip(queue, losses):
                self.logger.add_scalar("train/loss", v, step)
                if lr is not None:
                    self.logger.add_scalar("train/lr", lr, step)
            self.logger.flush()
        except Exception as e:  # :368-369
            print(f"[Warning] logging failed: {e}")

    def _validate_and_save(self):
        """:371-401: EMA mean across ranks, then rank 0 swaps EMA weights in, validates, saves, swaps back."""
        with torch.no_grad():
            opt = self.optimizer
            trained = opt.model                            # the transformer, or its adapter set
            trained.join_pending_update()                  # the overlapped AdamW/EMA update must have landed
            if opt.ema_shadow is not None and opt.gather_ema():
                pass        # sharded optimizer step: the ranks' slices of the (identical) shadow all-gathered instead of averaged
            elif opt.ema_shadow is not None and self.accelerator.num_processes > 1:
                # one flat all-reduce instead of ~600 per-tensor calls, through the transport the gradient buckets use (the
                # library's communicator when it is the transport: the launcher group is then gloo and would stage 3 GB
                # through the host)
                ddp = self.accelerator.ddp
                if ddp is not None:
                    ddp.allreduce_bulk(opt.ema_shadow, mean=True)
                else:
                    dist.all_reduce(opt.ema_shadow)
                    opt.ema_shadow /= self.accelerator.num_processes
            if self.accelerator.is_main_process:
                stored = None
                if opt.ema_shadow is not None:
                    stored = trained.flat_param.clone()
                    trained.flat_param.copy_(opt.ema_shadow)
                if self.timesteps:                          # :385-390 the first inference step's setting
                    self._set_adapter_scale(1.0 if 0 in self.timesteps else 0.0)
                try:
                    self.validate()
                except NotImplementedError:
                    pass
                if self.timesteps:                          # :396-397
                    self._set_adapter_scale(1.0)
                self.save_model()
                if stored is not None:
                    trained.flat_param.copy_(stored)


class WarmupLR:
    """LambdaLR(lr * min(1, step / warmup)) of common/trainer.py:255-262, including LambdaLR's initial step(0)."""

    def __init__(self, optimizer, warmup_steps):
        self.opt, self.warmup, self.last_epoch = optimizer, warmup_steps, 0
        self._apply()

    def _factor(self, s):
        return float(s) / float(max(1, self.warmup)) if s < self.warmup else 1.0

    def _apply(self):
        for g in self.opt.param_groups:
            g["lr"] = g["initial_lr"] * self._factor(self.last_epoch)

    def step(self):
        self.last_epoch += 1
        self._apply()

    def get_last_lr(self):
        return [g["lr"] for g in self.opt.param_groups]
