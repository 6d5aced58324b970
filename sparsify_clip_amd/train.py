"""Experiment runner: the reference's main / train_model / evaluate_model surface (sparsify_clip.py:534-676,
:682-986, :1071-1121) re-created for one process per GPU on the HIP library.

Step order, optimiser hyper-parameters, schedule inputs (1-based current_batch, lr = 0 on the first step when
there is no warm-up phase), phase switch on `epoch < only_lunif_epochs`, checkpoint names and logged keys follow the
reference; wandb is replaced by a local JSONL sink with the same keys, and the per-step `loss.item()` host sync
(:944) by a deferred read-back every `log_every` steps.
Deviations forced by the BASELINE batch sizes: the eval loader has its own batch size
(`eval_batch_size`, default min(batch_size, num_test_samples)), and directories are created before saving.
"""
from __future__ import annotations

import json
import os
import random
import time

import numpy as np
import torch

from . import dist as D
from . import ops
from . import uniformity as U
from ._lib import ScError
from .data import SyntheticLoader, caption_length, get_tokenizer
from .loss_dispatch import step_loss, step_loss_rows, validate_loss_type
from .model import create_model_and_transforms
from .optim import AdamW
from .schedules import get_cosine_schedule_with_warmup


class JsonlLogger:
    """wandb stand-in: one JSON object per log call, same keys as the reference (:659-667, :943-951)."""

    def __init__(self, path=None):
        self.path, self.rows = path, []
        if path:
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)

    def log(self, row: dict):
        self.rows.append(row)
        if self.path:
            with open(self.path, "a") as f:
                f.write(json.dumps(row) + "\n")


def set_seed(seed: int):
    """Reference :1071-1078 (cudnn flags have no counterpart here: every kernel of the library is deterministic)."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


class Trainer:
    """State of one training run; `step(images, captions_or_tokens)` is the hot path (reference :753-969)."""

    def __init__(self, config, device, steps_per_epoch, logger=None, model=None):
        validate_loss_type(config["loss_type"])
        self.config, self.device = config, torch.device(device)
        self.logger = logger or JsonlLogger(None)
        self.model = model or create_model_and_transforms(config["model"], pretrained=None, device=self.device, precision=config["precision"],
                                                          seed=config["seed"])[0]
        self.tokenizer = get_tokenizer(config["model"], self.model.cfg["ctx"], self.model.cfg["vocab"])
        self.model.train()
        self.temperature = config["anchor_temperature"]
        self.learnable_t = bool(config["anchor_temperature_learnable"])
        extra = []
        if self.learnable_t:   # :716-717 - a 0-dim fp32 parameter that the reference leaves on the CPU
            self.temperature = torch.nn.Parameter(torch.tensor(self.temperature, dtype=torch.float32), requires_grad=True)
            extra = [self.temperature]
        self.sync = D.GradSync(self.model)
        D.broadcast_parameters(self.model)
        self.optimizer = AdamW(self.model, lr=config["learning_rate"], extra_params=extra)          # :730 (torch defaults)
        self.t_total = steps_per_epoch * config["epochs"]                                          # :734
        self.scheduler = get_cosine_schedule_with_warmup(self.optimizer, int(0.20 * self.t_total), self.t_total, config=config)  # :735-736
        self.current_batch, self.epoch = 0, 0
        self.beta, self.alpha = 0.0, 0.0    # :743-744
        self.pending_logs = []
        self.text_stream = None
        self.on_gradients = None    # optional callable(trainer): runs when the (all-reduced) gradients are final, in front of the optimiser step

    def step(self, images, captions, text_len=None):
        """text_len (optional): the batch's longest caption in tokens (EOT included), known on the host.  With `text_trim: True` in the
        config (an extra key, default False = every position as the reference) the text tower then runs over that many positions
        only - same embeddings, same gradients (ClipModel.text_forward); without a given length it is read from the tokens,
        which costs a host sync when they are already on the device."""
        cfg, m = self.config, self.model
        self.current_batch += 1                                                    # :755
        tokens = captions if isinstance(captions, torch.Tensor) else self.tokenizer(captions)   # :762
        if not cfg.get("text_trim", False):
            text_len = None
        elif text_len is None:
            text_len = caption_length(tokens)
        # the two towers are independent until the loss head: the text tower runs on its own HIP stream so that its HBM-bound
        # kernels (LayerNorm, attention, embedding) overlap the image tower's GEMMs and vice versa
        main = torch.cuda.current_stream()
        if self.text_stream is None:
            self.text_stream = torch.cuda.Stream(device=self.device)      # equal priority (a high-priority text / weight-gradient stream measured worse, round 2)
        self.text_stream.wait_stream(main)
        with torch.cuda.stream(self.text_stream):
            txt_e = m.text_forward(tokens, seq_len=text_len)                       # :769
        img_e = m.image_forward(images)                                            # :768
        main.wait_stream(self.text_stream)
        send = D.gather_send_buffer(img_e.shape[0], img_e.shape[1], img_e.device)[0] if D.active() else (None, None)
        img_n, inv_i = ops.l2norm_fwd(img_e, 0.0, out=send[0])                     # :772 (written straight into the gather's send buffer)
        txt_n, inv_t = ops.l2norm_fwd(txt_e, 0.0, out=send[1])                     # :773
        img_all, txt_all = D.all_gather_embeddings(img_n, txt_n)
        temp = float(self.temperature.detach()) if self.learnable_t else float(self.temperature)
        world, rank = D.sharding()
        rows = img_all.shape[0] // world
        sharded = world > 1 and cfg.get("shard_loss_head", True) and ops.loss_rows_supported(img_all.shape[0], img_all.shape[1], rank * rows, rows)
        if sharded:   # this rank's rows x all columns of the O(B^2) terms; one small all-gather of LSE statistics in the middle
            res = step_loss_rows(cfg, img_all, txt_all, temp, self.epoch, self.current_batch, self.t_total, rank * rows, rows, D.exchange_packets,
                                 want_dtemp=self.learnable_t)
            if res.d_temp is not None:
                D.all_reduce_sum_(res.d_temp)
        else:
            res = step_loss(cfg, img_all, txt_all, temp, self.epoch, self.current_batch, self.t_total, want_dtemp=self.learnable_t)   # :778-938
        if res.beta is not None:
            self.beta = res.beta
        if res.alpha is not None:
            self.alpha = res.alpha
        row = {"learning_rate": self.scheduler.get_last_lr()[0]}
        if self.learnable_t:
            row["constrantive_temperature_learnable"] = temp                      # [sic] :945
        else:
            row.update(beta=self.beta, alpha=self.alpha)
        self.pending_logs.append((res.loss, row))
        self.optimizer.zero_grad()                                                 # :957
        d_img_e = ops.l2norm_bwd(img_n, inv_i, res.d_img if sharded else D.local_rows(res.d_img))
        d_txt_e = ops.l2norm_bwd(txt_n, inv_t, res.d_txt if sharded else D.local_rows(res.d_txt))
        self.text_stream.wait_stream(main)          # loss.backward() (:965): the two towers' backward passes are independent too
        with torch.cuda.stream(self.text_stream):
            m.text_backward(d_txt_e)
        m.image_backward(d_img_e)
        main.wait_stream(self.text_stream)
        if self.learnable_t and res.d_temp is not None:
            self.temperature.grad = res.d_temp.detach().cpu().reshape(())
        self.sync.wait_all()
        if self.on_gradients is not None:
            self.on_gradients(self)
        self.optimizer.step()                                                      # :966
        self.scheduler.step()                                                      # :969
        if len(self.pending_logs) >= cfg.get("log_every", 10):
            self.flush_logs()
        return res.loss

    def _resident_sets(self, n_micro, limit):
        """How many micro-batches keep their activations on the card between the two passes of step_cached: as many as fit beside what is
        already allocated (one set is known after the first forward), at most `limit`; decided once per shape."""
        m = self.model
        key = (n_micro, m.visual.batch, limit)
        if getattr(self, "_sets_key", None) != key:
            set_bytes = max(1, m.activation_set_bytes())
            free, total = torch.cuda.mem_get_info(self.device)
            free += torch.cuda.memory_reserved(self.device) - torch.cuda.memory_allocated(self.device)      # torch's cached, unused blocks
            margin = max(8 << 30, total // 16)      # head-room: backward scratch of a first step, allocator fragmentation
            fit = 1 + max(0, int((free - margin) // set_bytes))
            self._sets_key, self._sets_n = key, max(1, min(n_micro, fit, limit or n_micro))
        return self._sets_n

    def step_cached(self, images, captions, micro_batch, resident_sets=None):
        """The SAME step for a batch whose saved activations do not fit the GPU (global batch 8192 of ViT-B/32 is ~280 GB of them):
        (1) both towers forward over micro-batches, keeping only the [B, E] embeddings; (2) the loss head ONCE over the whole batch -
        the O(B^2) terms see every pair, which plain gradient accumulation over micro-batches would not give; (3) per micro-batch the
        forward again (its activations saved this time) and the backward with its rows of dL/d(embedding), parameter gradients
        accumulating in the flat buffer; (4) one optimiser step.  As many micro-batches as the card's memory holds keep their activations
        between the passes (ClipModel.activation_set; `resident_sets` caps the number, 1 = only the last one) and skip the second forward.
        The result equals step() on the whole batch up to fp32 summation order (tests/test_gpu_model.py::test_step_cached_equals_step) at
        between 1x and 4/3 of its encoder work.  Under data parallelism `images` / `captions` are the rank's shard: embeddings gathered, loss head by rows,
        each gradient bucket all-reduced once (behind the last micro-batch's backward).  LayerNorm models only: BatchNorm statistics of a
        micro-batch are not the batch's."""
        cfg, m = self.config, self.model
        if m.rn is not None:
            raise ScError("step_cached: the ModifiedResNet tower normalises over the batch; micro-batches would change the result")
        batch = images.shape[0]
        if micro_batch <= 0 or batch % micro_batch:
            raise ScError(f"step_cached: batch {batch} is not a multiple of the micro-batch {micro_batch}")
        self.current_batch += 1
        tokens = captions if isinstance(captions, torch.Tensor) else self.tokenizer(captions)
        tokens = tokens.to(self.device)
        main = torch.cuda.current_stream()
        if self.text_stream is None:
            self.text_stream = torch.cuda.Stream(device=self.device)
        parts = [slice(k, k + micro_batch) for k in range(0, batch, micro_batch)]

        def towers(sl):
            self.text_stream.wait_stream(main)
            with torch.cuda.stream(self.text_stream):
                te = m.text_forward(tokens[sl])
            ie = m.image_forward(images[sl])
            main.wait_stream(self.text_stream)
            return ie, te

        e = m.cfg["embed_dim"]
        img_e = torch.empty(batch, e, dtype=torch.float32, device=self.device)
        txt_e = torch.empty(batch, e, dtype=torch.float32, device=self.device)
        n, keep = len(parts), 1
        for idx, sl in enumerate(parts):
            m.activation_set(idx % keep)
            ie, te = towers(sl)
            img_e[sl].copy_(ie)
            txt_e[sl].copy_(te)
            if idx == 0:
                keep = self._resident_sets(n, resident_sets)      # micro-batches n - keep .. n - 1 stay resident for the second pass
        # from here to the embedding gradients: exactly step() - under data parallelism `images` is this rank's shard, the embeddings of
        # all ranks are gathered and the loss head (by rows, or replicated) sees the global batch
        send = D.gather_send_buffer(batch, e, img_e.device)[0] if D.active() else (None, None)
        img_n, inv_i = ops.l2norm_fwd(img_e, 0.0, out=send[0])
        txt_n, inv_t = ops.l2norm_fwd(txt_e, 0.0, out=send[1])
        img_all, txt_all = D.all_gather_embeddings(img_n, txt_n)
        temp = float(self.temperature.detach()) if self.learnable_t else float(self.temperature)
        world, rank = D.sharding()
        rows = img_all.shape[0] // world
        sharded = world > 1 and cfg.get("shard_loss_head", True) and ops.loss_rows_supported(img_all.shape[0], img_all.shape[1], rank * rows, rows)
        if sharded:
            res = step_loss_rows(cfg, img_all, txt_all, temp, self.epoch, self.current_batch, self.t_total, rank * rows, rows, D.exchange_packets,
                                 want_dtemp=self.learnable_t)
            if res.d_temp is not None:
                D.all_reduce_sum_(res.d_temp)
        else:
            res = step_loss(cfg, img_all, txt_all, temp, self.epoch, self.current_batch, self.t_total, want_dtemp=self.learnable_t)
        if res.beta is not None:
            self.beta = res.beta
        if res.alpha is not None:
            self.alpha = res.alpha
        row = {"learning_rate": self.scheduler.get_last_lr()[0]}
        if self.learnable_t:
            row["constrantive_temperature_learnable"] = temp
        else:
            row.update(beta=self.beta, alpha=self.alpha)
        self.pending_logs.append((res.loss, row))
        self.optimizer.zero_grad()
        d_img_e = ops.l2norm_bwd(img_n, inv_i, res.d_img if sharded else D.local_rows(res.d_img))
        d_txt_e = ops.l2norm_bwd(txt_n, inv_t, res.d_txt if sharded else D.local_rows(res.d_txt))
        comm = m.comm
        try:
            for idx in reversed(range(n)):               # the resident micro-batches first (their sets are free for the others afterwards)
                sl = parts[idx]
                m.activation_set(idx % keep)
                if idx < n - keep:
                    towers(sl)                           # not resident: the forward again, activations saved for this backward
                di, dt = d_img_e[sl].contiguous(), d_txt_e[sl].contiguous()
                m.comm = comm if idx == 0 else None      # a bucket is all-reduced once, when the LAST micro-batch processed has added to it
                self.text_stream.wait_stream(main)
                with torch.cuda.stream(self.text_stream):
                    m.text_backward(dt)                  # the first micro-batch overwrites the gradient buffer, the others accumulate
                m.image_backward(di)
                main.wait_stream(self.text_stream)
        finally:
            m.comm = comm
        if self.learnable_t and res.d_temp is not None:
            self.temperature.grad = res.d_temp.detach().cpu().reshape(())
        self.sync.wait_all()
        if self.on_gradients is not None:
            self.on_gradients(self)
        self.optimizer.step()
        self.scheduler.step()
        if len(self.pending_logs) >= cfg.get("log_every", 10):
            self.flush_logs()
        return res.loss

    # ---- optional full-state sidecar (the reference resumes WEIGHTS only, :719-724; SURVEY 5.4 asks for an opt-in sidecar)
    def full_state(self) -> dict:
        """Everything besides the weights that a bit-exact continuation needs, as a flat {str: Tensor} (weights_only-loadable)."""
        opt = self.optimizer
        st = {"adam_m": opt.m.detach().cpu(), "adam_v": opt.v.detach().cpu(),
              "counters": torch.tensor([opt.steps, self.current_batch, self.epoch, self.scheduler.last_epoch], dtype=torch.int64),
              "run": torch.tensor([self.t_total, self.config["epochs"]], dtype=torch.int64),      # the schedule's span: a resume continues THIS run
              "weights": torch.tensor([self.beta, self.alpha], dtype=torch.float64)}
        if self.learnable_t:
            val, grad, em, ev = opt.extra_state[id(self.temperature)]
            st.update(temperature=self.temperature.detach().cpu().reshape(1), temperature_m=em.cpu(), temperature_v=ev.cpu(),
                      temperature_steps=torch.tensor([opt.extra_steps[id(self.temperature)]], dtype=torch.int64))
        return st

    def load_full_state(self, st: dict):
        opt = self.optimizer
        if st["adam_m"].numel() != opt.m.numel():
            raise ScError("full-state sidecar belongs to a different model (moment size mismatch)")
        opt.m.copy_(st["adam_m"]), opt.v.copy_(st["adam_v"])
        opt.steps, self.current_batch, self.epoch, last = (int(v) for v in st["counters"])
        self.beta, self.alpha = (float(v) for v in st["weights"])
        if "run" in st and int(st["run"][0]) != self.t_total:
            # the LR schedule and the alpha / beta ramps are functions of the ORIGINAL run's total step count (reference :734-736, :41-64):
            # rebuild them from it, whatever `epochs` the resuming config carries
            self.t_total = int(st["run"][0])
            self.scheduler = get_cosine_schedule_with_warmup(self.optimizer, int(0.20 * self.t_total), self.t_total, config=self.config)
        self.scheduler.last_epoch = last
        lrs = [base * lmbda(last) for lmbda, base in zip(self.scheduler.lr_lambdas, self.scheduler.base_lrs)]
        for g, lr in zip(opt.param_groups, lrs):
            g["lr"] = lr
        self.scheduler._last_lr = lrs
        if self.learnable_t and "temperature" in st:
            self.temperature.data.copy_(st["temperature"].reshape(()))
            val, grad, em, ev = opt.extra_state[id(self.temperature)]
            em.copy_(st["temperature_m"]), ev.copy_(st["temperature_v"])
            opt.extra_steps[id(self.temperature)] = int(st["temperature_steps"])

    def flush_logs(self):
        for loss, row in self.pending_logs:
            self.logger.log({"train_loss": float(loss.item()), **row})
        self.pending_logs = []


def recall_from_ranks(rank: torch.Tensor, prefix: str) -> dict:
    """R@1/5/10 and their mean as percentages rounded to 4 dp from the 0-based rank of each query's true match
    (reference compute_metric_ret, sparsify_clip.py:382-392, :404-414, which finds the rank with a Python list.index loop)."""
    n = rank.numel()
    r = [(rank < k).sum().item() / n for k in (1, 5, 10)]
    return {f"{prefix}_r1": round(r[0] * 100, 4), f"{prefix}_r5": round(r[1] * 100, 4), f"{prefix}_r10": round(r[2] * 100, 4),
            f"{prefix}_ravg": round(sum(r) / 3 * 100, 4)}


def evaluate_model(model, test_loader, device, plot_embeddings=False, logger=None, tokenizer=None):
    """Retrieval + geometry metrics with the reference's keys (:534-676).  Plots (UMAP/t-SNE/PCA, :599-620) are out of scope."""
    was_training = model.training
    model.eval()
    tokenizer = tokenizer or get_tokenizer(model.name, model.cfg["ctx"], model.cfg["vocab"])
    imgs, txts = [], []
    with torch.no_grad():
        for images, captions in test_loader:
            tokens = captions if isinstance(captions, torch.Tensor) else tokenizer(captions)
            imgs.append(model.image_forward(images).clone())
            txts.append(model.text_forward(tokens).clone())
    if not imgs:
        raise ScError("evaluation loader yielded no batches (the reference's drop_last loader does this when "
                      "batch_size > num_test_samples; set eval_batch_size)")
    img, _ = ops.l2norm_fwd(torch.cat(imgs), 0.0)       # :624-625
    txt, _ = ops.l2norm_fwd(torch.cat(txts), 0.0)
    score = ops.gemm_f32(txt, img, trans_b=True)        # [N_text, N_image]  :628
    rank_f, rank_b, top_f, top_b = ops.retrieval_ranks(score)
    n = score.shape[0]
    m = ops.eval_metrics(img, txt, rank_f, rank_b).tolist()      # ONE read-back for gap / angular / true-pair cosine / recall counts

    def recall(counts, prefix):      # reference :382-392 rounding
        r = [c / n for c in counts]
        return {f"{prefix}_r1": round(r[0] * 100, 4), f"{prefix}_r5": round(r[1] * 100, 4), f"{prefix}_r10": round(r[2] * 100, 4),
                f"{prefix}_ravg": round(sum(r) / 3 * 100, 4)}

    final_log = {**recall(m[4:7], "forward"), **recall(m[7:10], "backward"),
                 "gap": round(m[0], 4),
                 "mean_angular_value_image": round(m[1], 4),
                 "mean_angular_value_text": round(m[2], 4),
                 "uniformity": round(U.uniformity(img, txt), 4),
                 "mean_cosine_similarity_true_pairs": round(m[3], 4)}
    if logger is not None:
        logger.log(final_log)
    model.train(was_training)
    return final_log


def dataset_loader(config, device):
    """The reference's dataset_loader(config) (:992-1065): (train, test) loaders with drop_last semantics.
      * synthetic: False  -> COCO captions from ./data/coco/... (the reference's paths), decoded on the host, cropped / resized /
        flipped / normalised ON THE DEVICE (input_pipeline.DeviceAugLoader); every rank reads its rank-major share of each batch;
      * input_pipeline: "device" -> the same pipeline fed by a synthetic uint8 dataset of COCO-like image sizes;
      * otherwise (default) -> pre-normalised fp32 batches resident in HBM (SyntheticLoader)."""
    world, rank = D.world_size(), D.get_rank()
    gb = config["batch_size"]
    if gb % world:
        raise ScError(f"batch_size {gb} is not divisible by the {world} data-parallel ranks")
    n_train = config["num_train_samples"] if config["num_train_samples"] != -1 else 118287   # COCO train2017
    if config.get("steps_per_epoch"):
        n_train = config["steps_per_epoch"] * gb
    n_test = config["num_test_samples"] if config["num_test_samples"] != -1 else 5000
    from .model import CONFIGS, canonical_name
    c = CONFIGS[canonical_name(config["model"])]
    if not config.get("synthetic", True) or config.get("input_pipeline") == "device":
        from . import input_pipeline as IP
        if not config.get("synthetic", True):
            troot, tann = IP.coco_paths("train")
            vroot, vann = IP.coco_paths("test")
            if not (os.path.exists(tann) and os.path.exists(vann)):
                raise ScError(f"synthetic: False needs the reference's COCO layout ({tann}, {vann}); it is not on this machine")
            train_ds = IP.CocoCaptionsDataset(troot, tann, config["num_train_samples"])
            test_ds = IP.CocoCaptionsDataset(vroot, vann, config["num_test_samples"])
        else:
            train_ds, test_ds = IP.SyntheticCocoDataset(n_train, config["seed"]), IP.SyntheticCocoDataset(n_test, config["seed"] + 777)
        eb = config.get("eval_batch_size") or min(gb, len(test_ds))
        train = IP.DeviceAugLoader(_RankShard(train_ds, gb, rank, world), gb // world, device, train=True, seed=config["seed"], size=c["image_size"])
        test = IP.DeviceAugLoader(test_ds, eb, device, train=False, seed=config["seed"] + 777, size=c["image_size"])
        return train, test
    eb = config.get("eval_batch_size") or min(gb, n_test)
    train = SyntheticLoader(n_train // world, gb // world, config["seed"] + 1000 * rank, device, c["image_size"], c["ctx"], c["vocab"])
    test = SyntheticLoader(n_test, eb, config["seed"] + 777, device, c["image_size"], c["ctx"], c["vocab"], distinct=max(1, n_test // eb))
    return train, test


class _RankShard:
    """View of a dataset for one data-parallel rank: of every global batch of `gb` consecutive samples the rank sees its
    rank-major slice (world = 1: the dataset itself).  Shuffling then happens inside the rank's share with the same seed on every
    rank, so a global batch is never split differently on two ranks."""

    def __init__(self, ds, gb, rank, world):
        self.ds, self.per, self.gb, self.rank, self.world = ds, gb // world, gb, rank, world

    def __len__(self):
        return (len(self.ds) // self.gb) * self.per

    def __getitem__(self, i):
        return self.ds[(i // self.per) * self.gb + self.rank * self.per + i % self.per]


def train_model(config, train_loader, test_loader, device, logger=None):
    """Reference :682-986."""
    trainer = Trainer(config, device, len(train_loader), logger)
    model = trainer.model
    start_epoch, end_epoch = 0, config["epochs"]
    if config["resume_checkpoint"]:   # :719-724 (weights only, `module.`-prefixed keys accepted)
        model.load_state_dict(torch.load(config["resume_checkpoint"], map_location="cpu", weights_only=True))
        start_epoch = config.get("resume_epoch", 0)
        end_epoch = start_epoch + config["epochs"]      # the reference runs a further full `epochs` (:749)
        sidecar = sidecar_path(config["resume_checkpoint"])
        if config.get("full_state_checkpoint"):   # opt-in: optimiser moments, step counters, schedule position, temperature - continue the ORIGINAL run
            if not os.path.exists(sidecar):
                raise ScError(f"full_state_checkpoint: {sidecar} is missing next to the checkpoint")
            st = torch.load(sidecar, map_location="cpu", weights_only=True)
            trainer.load_full_state(st)
            trainer.optimizer.model.refresh_shadows(full=True)
            start_epoch = trainer.epoch + 1
            end_epoch = int(st["run"][1]) if "run" in st else start_epoch + config["epochs"]      # the epochs the original run had left
    if len(train_loader) == 0:
        raise ScError("training loader yields no batches (num_train_samples < batch_size with drop_last, SURVEY 0.10)")
    micro = int(config.get("micro_batch") or 0)
    if micro > 0:   # checked before the first evaluation, not at the first training step
        if model.rn is not None:
            raise ScError("micro_batch: the ModifiedResNet tower normalises over the batch (BatchNorm); micro-batches would change the result")
        per_rank = config["batch_size"] // max(1, D.world_size())
        if per_rank % micro:
            raise ScError(f"micro_batch {micro} does not divide the {per_rank} pairs a rank processes per step")
        if config.get("text_trim"):
            raise ScError("micro_batch and text_trim cannot be combined (step_cached runs every position)")
    evaluate_model(model, test_loader, device, logger=logger)   # :740
    for epoch in range(start_epoch, end_epoch):
        trainer.epoch = epoch
        for images, captions in train_loader:
            if 0 < micro < images.shape[0]:
                trainer.step_cached(images, captions, micro)
            else:
                trainer.step(images, captions)
        trainer.flush_logs()
        evaluate_model(model, test_loader, device, logger=logger)   # :980
        if (epoch + 1) % config["save_checkpoint_every_n_epochs"] == 0 and D.get_rank() == 0:   # :982-984
            os.makedirs("models", exist_ok=True)
            path = f"models/{config['run_name']}_epoch_{epoch + 1}.pt"
            torch.save(model.state_dict(prefix="module."), path)
            if config.get("full_state_checkpoint"):
                torch.save(trainer.full_state(), sidecar_path(path))
    return model


def sidecar_path(checkpoint_path: str) -> str:
    """models/run_epoch_3.pt -> models/run_epoch_3.state.pt"""
    root, ext = os.path.splitext(checkpoint_path)
    return root + ".state" + ext


def main(config):
    """Reference :1084-1121 (wandb.init/save/finish -> JSONL under logs/)."""
    logger = JsonlLogger(os.path.join("logs", f"{config['run_name']}.jsonl")) if D.get_rank() == 0 else JsonlLogger(None)
    set_seed(config["seed"])
    if not torch.cuda.is_available():
        raise ScError("no GPU visible: this runner has no CPU path (the reference falls back to CPU at :1100; the oracle/ "
                      "directory holds the CPU restatement used for parity tests)")
    device = torch.device("cuda", config["device_id"])
    torch.cuda.set_device(device)
    train_loader, test_loader = dataset_loader(config, device)
    t0 = time.time()
    model = train_model(config, train_loader, test_loader, device, logger)
    final_log = evaluate_model(model, test_loader, device, logger=logger)
    if D.get_rank() == 0:
        os.makedirs("models", exist_ok=True)
        torch.save(model.state_dict(prefix="module."), "models/" + config["run_name"] + ".pt")   # :1118
    return {"final": final_log, "seconds": time.time() - t0}
