#!/usr/bin/env python
"""Headline benchmark: image-text pairs/sec of the full CLIP training step (both encoders fwd+bwd, global-batch loss
head, AdamW, DP collectives) on N MI355X of one node.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...)

Workload (BASELINE.json): experiment_6 loss stack (anchor + lalign + lunif(centroids), main phase) on ViT-B/32 in bf16,
synthetic 3x224x224 images + 77-token captions, LOCAL batch 1024 per GPU - weak scaling, so that N = 8 is the metric's
global batch 8192.  Prints ONE JSON line on rank 0.  `roofline` is measured live with HIP events on the dominant kernel
(the bf16 NT MFMA GEMM) by replaying one step's GEMM launches; `cpu_baseline` times the oracle's CPU step (fp32, plain torch)
on a bounded sample on the host cores (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as tdist  # noqa: E402

T_START = time.perf_counter()


def progress(msg):
    """Stage marker on stderr (the JSON line on stdout stays the only stdout output)."""
    print(f"[bench +{time.perf_counter() - T_START:7.1f}s] {msg}", file=sys.stderr, flush=True)


def host_threads():
    """Threads for the CPU baseline: the cores this process may actually use, capped at the GPU box's per-GPU share (16)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, int(os.environ.get("BENCH_CPU_THREADS", "16"))))


GF_PER_PAIR = {"ViT-B-32": 44.33, "ViT-L-14": 525.98}     # fwd+bwd = 3x fwd GFLOP per (image, caption) pair, SURVEY 8d
PEAK_BF16_TFLOPS = 2500.0                                   # dense bf16 MFMA peak, MI355X_MICROARCH.md


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=8)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--local-batch", type=int, default=None, help="pairs per forward/backward pass of a rank (default 1024).  Given WITHOUT --global-batch it "
                   "selects weak scaling: this many pairs per GPU and step (the per-GPU shards of BASELINE's other configurations, RN50, fp32)")
    p.add_argument("--model", type=str, default="ViT-B-32")
    p.add_argument("--experiment", type=str, default="experiment_6", help="which reference YAML (by prefix) supplies loss_type & co.")
    p.add_argument("--precision", type=str, default="bf16", choices=["bf16", "fp32"])
    p.add_argument("--cpu-baseline", type=int, default=1)
    p.add_argument("--cpu-batch", type=int, default=32)
    p.add_argument("--input-pipeline", type=int, default=0, help="N = 1 only: after the timed region, time the step again with every batch "
                   "coming through the host assembly + H2D copy + device augmentation of sparsify_clip_amd/input_pipeline.py (uint8 COCO-sized images)")
    p.add_argument("--text-trim", type=int, default=1, help="N = 1 only: after the timed region, time the step again with `text_trim: True` (the "
                   "text tower runs over the batch's longest caption instead of all 77 positions; same results) and report it as 'with_text_trim'; "
                   "the headline value always computes every position, as the reference does")
    p.add_argument("--global-batch", type=int, default=None, help="the step's global batch, split over the N ranks (default: 8192, BASELINE.json's metric, at "
                   "every N = strong scaling); a rank whose share exceeds --local-batch runs it by micro-batches of --local-batch through "
                   "Trainer.step_cached (the same step; activations of as many micro-batches as fit stay resident, the rest is forwarded twice).  "
                   "0 = weak scaling (--local-batch pairs per GPU)")
    p.add_argument("--global-batch-one-gpu", type=int, default=8192, help="N = 1 only: after the timed region, time the metric's global batch as ONE step on "
                   "this GPU through Trainer.step_cached (micro-batches of --local-batch); 0 = off")
    p.add_argument("--simulate-dp", type=int, default=8, help="N = 1 only: after the timed region, time the step again with the loss head fed a "
                   "global batch of simulate_dp x local_batch rows (filler rows for the absent ranks) = the per-GPU work of that DP job; 1 = off")
    args = p.parse_args()
    if args.global_batch is None:      # the metric's configuration unless the caller asked for a per-GPU batch
        args.global_batch = 8192 if args.local_batch is None else 0
    if args.local_batch is None:
        args.local_batch = 1024
    return args


def reference_config(prefix, model, global_batch, precision):
    """The parsed reference YAML (tests/golden/configs.json, produced from /root/reference by oracle/make_golden.py) + overrides."""
    from sparsify_clip_amd.config import finalize_config
    with open(os.path.join(ROOT, "tests", "golden", "configs.json")) as f:
        table = json.load(f)
    key = [k for k in table if os.path.basename(k).startswith(prefix + "-")][0]
    return key, finalize_config(table[key], 0, {"model": model, "batch_size": global_batch, "precision": precision, "log_every": 1 << 30})


def step_gemm_launches(cfg, batch, k_patch, grid):
    """((M, N, K), epilogue, repetitions) of every bf16 NT GEMM launch of one training step: the four forward linears of a
    block with the epilogues the step fuses into them, and the four activation-gradient GEMMs."""
    launches = []
    for rows, w, layers in ((batch * (grid ** 2 + 1), cfg["v_width"], cfg["v_layers"]), (batch * cfg["ctx"], cfg["t_width"], cfg["t_layers"])):
        shapes = [((rows, 3 * w, w), "bias"), ((rows, w, w), "bias+resid"), ((rows, 4 * w, w), "bias+gelu+pre"), ((rows, w, 4 * w), "bias+resid"),   # fwd: qkv, out, fc1, fc2
                  ((rows, w, 3 * w), "plain"), ((rows, w, w), "plain"), ((rows, w, 4 * w), "plain"), ((rows, 4 * w, w), "dgelu")]                    # dX of the same four
        launches += [(s, e, layers) for s, e in shapes]
    if k_patch % 64 == 0:
        launches.append(((batch * grid ** 2, cfg["v_width"], k_patch), "plain", 1))
    return launches


def gemm_launch_operands(m, n, k, kind, device):
    """Operands of one replayed NT GEMM launch with the epilogue the training step fuses into it."""
    from sparsify_clip_amd import ops
    a = torch.randn(m, k, device=device).to(torch.bfloat16)
    b = torch.randn(n, k, device=device).to(torch.bfloat16)
    bias = torch.randn(n, device=device)
    epi, out_dtype, keep = None, torch.bfloat16, None
    if kind == "bias":
        epi = ops.make_epilogue(bias=bias, ld_aux=n)
    elif kind == "bias+resid":
        keep = torch.randn(m, n, device=device)
        epi, out_dtype = ops.make_epilogue(bias=bias, resid=keep, ld_aux=n), torch.float32
    elif kind == "bias+gelu+pre":
        keep = torch.empty(m, n, dtype=torch.bfloat16, device=device)
        epi = ops.make_epilogue(bias=bias, pre_out=keep, act=1, ld_aux=n)
    elif kind == "dgelu":
        keep = torch.randn(m, n, device=device).to(torch.bfloat16)
        epi = ops.make_epilogue(dgelu_pre=keep, ld_aux=n)
    out = torch.empty(m, n, dtype=out_dtype, device=device)
    return a, b, out, epi, keep


def gemm_launch_bytes(m, n, k, kind):
    """Algorithmic HBM bytes of one launch: A + B once, C once, plus the epilogue's own operands."""
    base = 2 * (m * k + n * k)
    if kind == "bias+resid":
        return base + 4 * m * n + 4 * m * n            # fp32 residual in, fp32 C out
    if kind in ("bias+gelu+pre", "dgelu"):
        return base + 2 * m * n + 2 * m * n            # bf16 C out + saved pre-activation out / in
    return base + 2 * m * n


def gemm_roofline(model, device):
    """Replay the bf16 NT GEMM launches of one training step - same shapes, same fused epilogues (bias, GELU + saved
    pre-activation, fp32 residual add, GELU' multiply) - with a HIP event pair around each launch on the launch stream;
    algorithmic FLOPs = 2*M*N*K per launch.  `traffic` = HBM bytes per launch from the committed rocprofv3 PMC passes over the
    same launch set (tools/gpu_pmc_traffic.sh: FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE), or null when absent."""
    from sparsify_clip_amd import ops
    launches = step_gemm_launches(model.cfg, model.visual.batch, model.k_patch, model.grid)
    total_flops, total_ms, count = 0.0, 0.0, 0
    for (m, n, k), kind, reps in launches:
        a, b, out, epi, keep = gemm_launch_operands(m, n, k, kind, device)
        ops.gemm_bf16_nt(a, b, out=out, epi=epi)
        # three samples of FOUR back-to-back launches each (the event pair then brackets kernel time, not the ~5 us between an event and the
        # launch behind it - with one launch per pair the figure read 3-5 % above rocprofv3's kernel durations); the median sample counts
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
        for s, e in evs:
            s.record()
            for _ in range(4):
                ops.gemm_bf16_nt(a, b, out=out, epi=epi)
            e.record()
        torch.cuda.synchronize()
        ms = sorted(s.elapsed_time(e) for s, e in evs)[1] / 4.0
        total_flops += 2.0 * m * n * k * reps
        total_ms += ms * reps
        count += reps
        del a, b, out, keep, epi
    achieved = total_flops / (total_ms * 1e-3) / 1e12
    traffic = None
    import glob
    tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_gemm_traffic*.json")))   # the latest committed PMC pass (tools/gpu.sh pmc_traffic)
    tpath = tfiles[-1] if tfiles else ""
    if tpath:
        with open(tpath) as f:
            t = json.load(f)
        if t.get("launches") == count and t.get("local_batch") == model.visual.batch:
            traffic = t["hbm_bytes_per_launch"]
    return {"bound": "mfma", "kernel": "gemm_bf16_nt_pers_kernel (persistent 256x256 AGPR kernel; every NT launch of one forward + backward pass over "
                                     f"{model.visual.batch} pairs with its fused epilogue - a step is one such pass per micro-batch, plus the second forwards)",
            "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "launches_per_pass": count,
            "avg_launch_us": round(total_ms * 1e3 / count, 1), "flops_per_launch_avg": total_flops / count,
            "algorithmic_bytes_per_launch_avg": sum(gemm_launch_bytes(m, n, k, kind) * reps for (m, n, k), kind, reps in launches) / count}


def cpu_baseline(cfg, model_name, batch):
    """Oracle CPU step (plain torch fp32, the restatement pinned to the reference by tests/golden) on a bounded sample."""
    from oracle.clip_model import create_model, synthetic_batch
    from oracle.train_step import CpuTrainer
    torch.set_num_threads(host_threads())
    ref = create_model(model_name, seed=0)
    c = dict(cfg, batch_size=batch)
    tr = CpuTrainer(c, 100, model=ref)
    tr.epoch = 1
    images, tokens = synthetic_batch(1, batch, ref.cfg)
    images, tokens = torch.tensor(images), torch.tensor(tokens)
    progress(f"cpu baseline: {torch.get_num_threads()} threads, warm-up step")
    tr.step(images, tokens)   # untimed first step (thread pool / allocator warm-up)
    progress("cpu baseline: timed steps")
    n = 2
    t0 = time.perf_counter()
    for _ in range(n):
        tr.step(images, tokens)
    dt = time.perf_counter() - t0
    return {"value": round(batch * n / dt, 3), "unit": "pairs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} full training steps of {model_name} fp32 at {batch} pairs/step (oracle/train_step.py, plain torch on the host cores)"}


class simulated_world:
    """Bench-only (N = 1): what ONE rank of a w-GPU job computes per step.  While active, the four places where the trainer meets
    sparsify_clip_amd.dist see a world of w ranks with this process as rank 0: the embedding gather returns a [w * Bl, E] batch whose first
    Bl rows are this rank's and whose other rows are fixed unit-norm filler, the loss head is sharded w ways (this rank's rows against
    all columns), the statistics exchange returns w copies of this rank's packet.  No collective is simulated.  The patch lives here, not in the
    product module."""

    def __init__(self, w):
        self.w, self.fill, self.saved = int(w), {}, {}

    def _gather(self, local, which):
        bl, e = local.shape
        key = (bl, e, which, str(local.device))
        full = self.fill.get(key)
        if full is None:
            g = torch.Generator(device="cpu").manual_seed(977 + which)
            full = self.fill[key] = torch.nn.functional.normalize(torch.randn(self.w * bl, e, generator=g), dim=-1).to(local.device)
        full[:bl].copy_(local)
        return full

    def __enter__(self):
        from sparsify_clip_amd import dist as D
        assert not D.active(), "simulated_world is for a process without a process group"
        w = self.w
        patch = {"all_gather_embeddings": lambda img, txt: (self._gather(img, 0), self._gather(txt, 1)),
                 "sharding": lambda: (w, 0),
                 "exchange_packets": lambda packet: packet.unsqueeze(0).expand(w, -1).contiguous(),
                 "local_rows": lambda full: full[: full.shape[0] // w].contiguous()}
        for name, fn in patch.items():
            self.saved[name] = getattr(D, name)
            setattr(D, name, fn)
        return self

    def __exit__(self, *exc):
        from sparsify_clip_amd import dist as D
        for name, fn in self.saved.items():
            setattr(D, name, fn)
        self.fill.clear()
        return False


def self_launch(args):
    """`python bench.py --gpus N` outside torchrun: start the N ranks as CHILD processes (torch.distributed.run on 127.0.0.1) and
    relay rank 0's JSON line.  Decided before this process touches the GPU (no torch.cuda call, no library load); the parent
    never initialises HIP and never exec()s - it only waits for the launcher and exits with its return code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    progress(f"--gpus {args.gpus} without WORLD_SIZE: launching {args.gpus} ranks as child processes (port {port})")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    for ln in lines[-1:]:
        print(ln, flush=True)
    if proc.returncode == 0 and not lines:
        progress("the ranks exited 0 but printed no JSON line")
        return 1
    return proc.returncode


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    from sparsify_clip_amd import dist as D
    rank, local_rank, world = D.init_process_group()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    device = torch.device("cuda", local_rank % max(1, torch.cuda.device_count()))   # ranks > devices only in the gloo rehearsal
    torch.cuda.set_device(device)
    from sparsify_clip_amd.data import synthetic_batch
    from sparsify_clip_amd.train import Trainer
    strong = args.global_batch > 0
    if strong and args.global_batch % (world * args.local_batch) and args.global_batch % world:
        raise SystemExit(f"bench.py: --global-batch {args.global_batch} is not divisible by {world} ranks")
    rank_batch = args.global_batch // world if strong else args.local_batch      # pairs per rank and step
    global_batch = rank_batch * world
    key, cfg = reference_config(args.experiment, args.model, global_batch, args.precision)
    progress(f"rank {rank}/{world}: building {args.model} ({args.precision}) and synthetic batches")
    trainer = Trainer(cfg, device, steps_per_epoch=1000)
    trainer.epoch = max(1, cfg["only_lunif_epochs"])       # main phase: the full loss stack, not the warm-up branch
    c = trainer.model.cfg
    if strong and rank_batch > args.local_batch and trainer.model.rn is not None:
        raise SystemExit(f"bench.py: {args.model} normalises over the batch (BatchNorm): a {rank_batch}-pair shard cannot run by micro-batches; "
                         f"pass --local-batch (weak scaling) for it")
    if strong:      # big shards: token rows from the seeded generator, pixels drawn on the device (i.i.d. N(0, 1) as SURVEY 8d, another stream);
        gen = torch.Generator(device=device).manual_seed(42 + rank)      # the two batches share the 4.9 GB of pixels, in another order
        pixels = torch.randn(rank_batch, 3, c["image_size"], c["image_size"], device=device, generator=gen)
        batches = [(pixels, synthetic_batch(42 + rank + 100 * k, rank_batch, 8, c["ctx"], c["vocab"])[1].to(device)) for k in range(2)]
        del pixels
    else:
        batches = [tuple(t.to(device) for t in synthetic_batch(42 + rank + 100 * k, args.local_batch, c["image_size"], c["ctx"], c["vocab"])) for k in range(2)]
    cached = strong and rank_batch > args.local_batch and rank_batch % args.local_batch == 0

    resident_cap = [None]      # None = as many micro-batches' activations as Trainer._resident_sets finds room for

    def run_step(b):
        return trainer.step_cached(*b, args.local_batch, resident_sets=resident_cap[0]) if cached else trainer.step(*b)

    def barrier():
        if world > 1:
            tdist.barrier()
        torch.cuda.synchronize()

    for i in range(max(args.warmup, 1 if cached else 0)):
        step_state = (trainer.current_batch, list(trainer.pending_logs))
        try:
            run_step(batches[i % 2])
            torch.cuda.synchronize()
        except torch.cuda.OutOfMemoryError:
            # the memory estimate was too optimistic for this card's state (another context, fragmentation): park fewer micro-batches and go on;
            # an interrupted step leaves half-accumulated gradients behind, which the next step overwrites.  ONE rank only: with several
            # ranks the interrupted step may already have issued its embedding all-gather / statistics exchange while the others sit in
            # their bucket all-reduces - re-running it would mismatch the collective sequences, so the run fails instead
            if not cached or world > 1:
                raise
            trainer.current_batch, trainer.pending_logs = step_state      # the retried step is the same step: same schedule position, one log row
            kept = getattr(trainer, "_sets_n", 2)
            resident_cap[0] = max(1, kept // 2)
            progress(f"out of device memory with {kept} resident micro-batches: retrying with {resident_cap[0]}")
            trainer.model.activation_set(0)
            trainer.model.drop_activation_sets()
            torch.cuda.empty_cache()
            run_step(batches[i % 2])
            torch.cuda.synchronize()
        progress(f"warm-up step {i + 1}/{args.warmup} done")
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = run_step(batches[i % 2])
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        elapsed = t.item()
    last_loss = float(loss.item())
    progress(f"timed {args.steps} steps in {elapsed:.3f}s")
    pairs_per_s = global_batch * args.steps / elapsed
    out = {"metric": "image-text pairs/sec", "value": round(pairs_per_s, 1), "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong" if strong else "weak",
           "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
           "config": {"workload": f"{os.path.basename(key)} (loss_type {cfg['loss_type']!r}, main phase), {args.model}, global batch {global_batch}, "
                                  f"{rank_batch} pairs per GPU" + (f" run by micro-batches of {args.local_batch} (Trainer.step_cached: embeddings of every "
                                  f"micro-batch, ONE loss head over the global batch, backward per micro-batch from resident activations where the card's "
                                  f"memory holds them, after a second forward otherwise; same result as the plain step - the saved activations of 8192 pairs, "
                                  f"~280 GB, do not fit one GPU)" if cached else "") + ", AdamW, random-init weights; "
                                  + ("strong scaling: BASELINE.json's global batch at every N" if strong else
                                     f"weak scaling: {args.local_batch} pairs per GPU, the per-GPU shard of an N-GPU job"),
                      "global_batch": global_batch, "local_batch": rank_batch, "micro_batch": min(rank_batch, args.local_batch), "parallelism": f"dp{world}"},
           "last_loss": last_loss, "peak_device_memory_gib": round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 1),
           "step_mfma_frac": round(pairs_per_s * GF_PER_PAIR.get(args.model, 0.0) / 1e3 / (world * PEAK_BF16_TFLOPS), 4)}
    if world == 1 and strong and rank_batch > args.local_batch:
        # N = 1 extras below describe ONE forward/backward pass of --local-batch pairs (the per-GPU shard of the 8-GPU job): free the parked
        # activations and cut the batches down to that size
        out["resident_micro_batches"] = getattr(trainer, "_sets_n", 1)
        trainer.model.drop_activation_sets()
        batches = [(b[0][: args.local_batch].clone(), b[1][: args.local_batch].clone()) for b in batches]
        torch.cuda.empty_cache()
        for i in range(2):
            trainer.step(*batches[i % 2])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        nsh = min(args.steps, 8)
        for i in range(nsh):
            trainer.step(*batches[i % 2])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / nsh
        out["shard_step"] = {"local_batch": args.local_batch, "ms_per_step": round(dt * 1e3, 3), "pairs_per_s": round(args.local_batch / dt, 1),
                             "note": "the plain step over --local-batch pairs (what one rank of the 8-GPU job runs, Trainer.step; the line bench.py printed as "
                                     "`value` until the global batch ran on one GPU)"}
    if world == 1 and args.simulate_dp > 1:
        # what ONE rank of a simulate_dp-GPU job computes per step (its row block of the global-batch loss head included; no collective)
        with simulated_world(args.simulate_dp):
            trainer.step(*batches[0])
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            nsim = min(args.steps, 8)
            for i in range(nsim):
                trainer.step(*batches[i % 2])
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / nsim
        out["dp_rank_equivalent"] = {"simulated_world": args.simulate_dp, "loss_head_batch": args.simulate_dp * args.local_batch,
                                     "ms_per_step": round(dt * 1e3, 3), "pairs_per_s_per_gpu": round(args.local_batch / dt, 1),
                                     "note": "compute of one rank of the DP job: the loss head takes this rank's 1/world of the rows against the whole gathered batch "
                                             "(step_loss_rows); the statistics exchange and the other collectives are not simulated"}
    if world == 1 and not strong and args.global_batch_one_gpu > args.local_batch and trainer.model.rn is None:
        # the metric's own global batch on ONE GPU: Trainer.step_cached = towers over micro-batches, loss head once over all pairs, forward again +
        # backward per micro-batch (the saved activations of 8192 pairs, ~280 GB, do not fit; 4/3 of the encoder work instead)
        gb = args.global_batch_one_gpu - args.global_batch_one_gpu % args.local_batch
        big_tokens = synthetic_batch(4242, gb, 8, c["ctx"], c["vocab"])[1].to(device)
        big_images = torch.randn(gb, 3, c["image_size"], c["image_size"], device=device, generator=torch.Generator(device=device).manual_seed(4242))
        trainer.step_cached(big_images, big_tokens, args.local_batch)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(2):
            trainer.step_cached(big_images, big_tokens, args.local_batch)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / 2
        del big_images, big_tokens
        resident = getattr(trainer, "_sets_n", 1)
        trainer.model.drop_activation_sets()      # up to ~240 GB of parked activations: give them back before the next measurements
        torch.cuda.empty_cache()
        out["global_batch_one_gpu"] = {"global_batch": gb, "micro_batch": args.local_batch, "resident_micro_batches": resident, "ms_per_step": round(dt * 1e3, 3), "pairs_per_s": round(gb / dt, 1),
                                       "note": "the whole global-batch step on one GPU (Trainer.step_cached: embeddings of all micro-batches, ONE loss head over "
                                               "every pair, then backward per micro-batch - from its resident activations where the card's memory holds them, after "
                                               "a second forward otherwise - one optimiser step); equal to step() up to fp32 summation "
                                               "order (tests/test_gpu_model.py::test_step_cached_equals_step); not the headline value"}
    if world == 1 and args.text_trim:
        # opt-in: text tower over the longest caption of the batch only (padding behind EOT cannot influence the result under the causal mask)
        from sparsify_clip_amd.data import caption_length
        lens = [caption_length(b[1]) for b in batches]
        trainer.config["text_trim"] = True
        for i in range(2):
            trainer.step(*batches[i % 2], text_len=lens[i % 2])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        ntrim = min(args.steps, 8)
        for i in range(ntrim):
            trainer.step(*batches[i % 2], text_len=lens[i % 2])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / ntrim
        trainer.config["text_trim"] = False
        out["with_text_trim"] = {"ms_per_step": round(dt * 1e3, 3), "pairs_per_s": round(args.local_batch / dt, 1), "longest_caption_tokens": max(lens),
                                 "note": "config key text_trim: the text tower runs over the batch's longest caption (synthetic captions: 5-30 words + SOT/EOT, "
                                         "SURVEY 8d) rounded up to a multiple of 8 instead of all 77 positions; embeddings and gradients are the same "
                                         "(tests/test_gpu_model.py::test_text_trim_equivalence); not the headline value"}
    if world == 1 and args.input_pipeline:
        # the same step fed by the real input path: uint8 pixels -> pinned staging -> H2D on a side stream -> device crop/resize/flip/normalise
        from sparsify_clip_amd.input_pipeline import DeviceAugLoader, SyntheticCocoDataset
        nsteps = min(args.steps, 8)
        loader = DeviceAugLoader(SyntheticCocoDataset((nsteps + 2) * args.local_batch, seed=7), args.local_batch, device, train=True, seed=7,
                                 size=c["image_size"], workers=host_threads(), tokenizer=trainer.tokenizer)
        it = iter(loader)
        for _ in range(2):
            trainer.step(*next(it))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(nsteps):
            trainer.step(*next(it))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / nsteps
        del it
        out["with_input_pipeline"] = {"ms_per_step": round(dt * 1e3, 3), "pairs_per_s": round(args.local_batch / dt, 1), "host_threads": host_threads(),
                                      "note": "uint8 640x480-class images: host crop-box sampling + packing, H2D (PCIe) on a copy stream, device "
                                              "RandomResizedCrop/HFlip/Normalize and host tokenisation, all inside the timed loop"}
    if world > 1:
        # the ranks part here: rank 0's roofline replay runs alone, nobody sits in a collective with a timeout meanwhile
        tdist.barrier()
        tdist.destroy_process_group()
    if rank == 0:
        if args.precision == "bf16" and trainer.model.rn is None:      # the replay set is the ViT towers' (the headline configuration)
            progress("roofline: replaying the step's NT GEMM launches under HIP events")
            out["roofline"] = gemm_roofline(trainer.model, device)
        if args.cpu_baseline and world == 1:
            del trainer
            torch.cuda.empty_cache()
            out["cpu_baseline"] = cpu_baseline(cfg, args.model, args.cpu_batch)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
