"""CLIP encoders (ViT image tower + text transformer) on the HIP library.

Host-side mirror of what the reference obtains from ``open_clip.create_model_and_transforms(name, pretrained=None,
device=device)`` (reference sparsify_clip.py:685-703) and drives through ``encode_image`` / ``encode_text``
(:768-769): same parameter names (open_clip state_dict keys), same call surface, but every FLOP runs in
libsparsify_hip.so.  There is no torch.nn module and no CPU path in here.

Memory layout (sized for 288 GB of HBM per GPU)
  * one flat fp32 buffer holds every parameter (256-byte aligned slots), a second one their gradients, so the
    optimiser is ONE launch and a data-parallel gradient bucket is a contiguous slice;
  * bf16 mode keeps a bf16 shadow of the flat buffer (written by the AdamW kernel) plus [in,out] copies of the
    block GEMM weights for the activation-gradient GEMMs;
  * activations saved for the backward are preallocated per layer for the current batch and reused every step.
Parameter order = reverse of the order gradients become final in the backward, per tower:
  [tower head | block L-1 ... block 0 | tower stem], so bucket k of the all-reduce is ready before bucket k+1.
"""
from __future__ import annotations

import os

import math
from collections import OrderedDict

import torch

from . import ops
from ._lib import BlockDesc, EventSet, ScError, SC_BF16, SC_F32, sc_dtype

CONFIGS = {
    "ViT-B-32": dict(embed_dim=512, image_size=224, patch=32, v_width=768, v_layers=12, v_heads=12,
                     ctx=77, vocab=49408, t_width=512, t_layers=12, t_heads=8),
    "ViT-L-14": dict(embed_dim=768, image_size=224, patch=14, v_width=1024, v_layers=24, v_heads=16,
                     ctx=77, vocab=49408, t_width=768, t_layers=12, t_heads=12),
    # open_clip "RN50" (the `model:` of every reference YAML): ModifiedResNet image tower (resnet.py) + the 512-wide text tower
    "RN50": dict(embed_dim=1024, image_size=224, v_kind="resnet", v_layers=(3, 4, 6, 3), v_width=64,
                 ctx=77, vocab=49408, t_width=512, t_layers=12, t_heads=8),
    "test-rn": dict(embed_dim=64, image_size=64, v_kind="resnet", v_layers=(1, 2, 1, 1), v_width=16,
                    ctx=16, vocab=512, t_width=64, t_layers=1, t_heads=1),
    "test-rn64": dict(embed_dim=128, image_size=64, v_kind="resnet", v_layers=(1, 1, 1, 1), v_width=128,
                      ctx=16, vocab=512, t_width=64, t_layers=1, t_heads=1),
    "test-rn32": dict(embed_dim=64, image_size=64, v_kind="resnet", v_layers=(1, 1, 1, 1), v_width=64,      # RN50's channel counts (32-channel stem)
                      ctx=16, vocab=512, t_width=64, t_layers=1, t_heads=1),
    "tiny": dict(embed_dim=64, image_size=64, patch=32, v_width=128, v_layers=2, v_heads=2,
                 ctx=16, vocab=512, t_width=64, t_layers=2, t_heads=1),
    "test-small": dict(embed_dim=128, image_size=224, patch=32, v_width=128, v_layers=1, v_heads=2,
                       ctx=77, vocab=1000, t_width=64, t_layers=1, t_heads=1),
    "test-l14": dict(embed_dim=64, image_size=224, patch=14, v_width=128, v_layers=1, v_heads=2,
                     ctx=77, vocab=1000, t_width=64, t_layers=1, t_heads=1),
    # 101 image tokens / 100 text tokens: the sequence lengths between the short (<= 80) and the long (> 128) attention kernels
    "test-s101": dict(embed_dim=64, image_size=320, patch=32, v_width=128, v_layers=1, v_heads=2,
                      ctx=100, vocab=1000, t_width=64, t_layers=1, t_heads=1),
}
BLOCK_PARAMS = ["ln_1.weight", "ln_1.bias", "attn.in_proj_weight", "attn.in_proj_bias", "attn.out_proj.weight",
                "attn.out_proj.bias", "ln_2.weight", "ln_2.bias", "mlp.c_fc.weight", "mlp.c_fc.bias",
                "mlp.c_proj.weight", "mlp.c_proj.bias"]
GEMM_WEIGHTS = ["attn.in_proj_weight", "attn.out_proj.weight", "mlp.c_fc.weight", "mlp.c_proj.weight"]


def canonical_name(name: str) -> str:
    return name.replace("/", "-")


def _block_shapes(w):
    return OrderedDict([("ln_1.weight", (w,)), ("ln_1.bias", (w,)), ("attn.in_proj_weight", (3 * w, w)),
                        ("attn.in_proj_bias", (3 * w,)), ("attn.out_proj.weight", (w, w)), ("attn.out_proj.bias", (w,)),
                        ("ln_2.weight", (w,)), ("ln_2.bias", (w,)), ("mlp.c_fc.weight", (4 * w, w)),
                        ("mlp.c_fc.bias", (4 * w,)), ("mlp.c_proj.weight", (w, 4 * w)), ("mlp.c_proj.bias", (w,))])


class _Tower:
    """Static description + per-batch buffers of one tower."""

    def __init__(self, kind, prefix, width, layers, heads, seq, causal):
        self.kind, self.prefix, self.width, self.layers, self.heads, self.seq, self.causal = kind, prefix, width, layers, heads, seq, causal
        self.batch = 0
        self.descs = []
        self.bufs = {}
        self.events = None
        self.tickets = None    # tile tickets of the tower's NT GEMMs (one stream at a time: the tower's main stream)
        self.run_seq = seq     # sequence length of the current forward / backward (text tower: <= seq when the batch is trimmed)
        self.run_descs = None  # the block descriptors in use (tower.descs, or their copies with seq = run_seq)
        self.trimmed = {}      # run_seq -> descriptor copies
        self.sets = {}         # parked activation sets: id -> (bufs, descs, batch, run_seq), see ClipModel.activation_set


class ClipModel:
    """open_clip-compatible CLIP (ViT towers only) whose arithmetic is the HIP library."""

    def __init__(self, name="ViT-B-32", device="cuda:0", precision="bf16", seed=0):
        cname = canonical_name(name)
        if cname not in CONFIGS:
            raise ScError(f"model {name!r} is not implemented natively (RN50, ViT-B-32, ViT-L-14 and the test geometries are)")
        self.name, self.cfg = cname, CONFIGS[cname]
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ScError("ClipModel needs a GPU device: there is no CPU path in this package")
        if precision not in ("bf16", "fp32"):
            raise ScError(f"precision must be 'bf16' or 'fp32', got {precision!r}")
        self.dtype = torch.bfloat16 if precision == "bf16" else torch.float32
        self.training = True
        c = self.cfg
        self.buffers = {}     # non-trainable state that belongs in the state_dict (BatchNorm running statistics of the ResNet tower)
        self.rn = None
        if c.get("v_kind") == "resnet":
            from .resnet import ResNetVisual
            self.rn = ResNetVisual(self, c)
            self.visual = None
        else:
            self.grid = c["image_size"] // c["patch"]
            self.k_patch = 3 * c["patch"] ** 2
            self.k_pad = ((self.k_patch + 63) // 64) * 64
            self.visual = _Tower("image", "visual.transformer.resblocks.", c["v_width"], c["v_layers"], c["v_heads"], self.grid ** 2 + 1, 0)
        self.text = _Tower("text", "transformer.resblocks.", c["t_width"], c["t_layers"], c["t_heads"], c["ctx"], 1)
        self._layout()
        self.flat = torch.zeros(self.n_total, dtype=torch.float32, device=self.device)
        self.flat_grad = torch.zeros(self.n_total, dtype=torch.float32, device=self.device)
        self.flat_bf16 = torch.zeros(self.n_total, dtype=torch.bfloat16, device=self.device) if self.dtype == torch.bfloat16 else None
        self.wt = {}          # name -> bf16 [in,out] copy
        self._grads_fresh = True   # next backward overwrites (True) or accumulates (False)
        self.comm = None      # optional data-parallel hook (dist.GradSync)
        self._act_set = 0     # id of the towers' current activation set
        self._scratch = {}
        self._aux_stream, self.wt_ready = None, None
        self._side = None      # side stream of the weight-gradient GEMMs (bf16 path)
        self._wt_batch = None  # batched launcher of the [in,out] weight copies
        if self.rn is not None:
            for bname, shape, fill, bdt in self.rn.buffer_specs():
                self.buffers[bname] = torch.full(shape, fill, dtype=bdt, device=self.device)
        self.init_parameters(seed)

    # ------------------------------------------------------------------------------------------ layout
    def _layout(self):
        c = self.cfg
        order = []
        vw, tw, e = c["v_width"], c["t_width"], c["embed_dim"]
        # image tower: head, blocks L-1..0, stem
        if self.rn is not None:
            order += self.rn.param_order()
        else:
            order += [("visual.ln_post.weight", (vw,)), ("visual.ln_post.bias", (vw,)), ("visual.proj", (vw, e))]
            for i in reversed(range(c["v_layers"])):
                order += [(f"visual.transformer.resblocks.{i}.{k}", s) for k, s in _block_shapes(vw).items()]
            order += [("visual.ln_pre.weight", (vw,)), ("visual.ln_pre.bias", (vw,)), ("visual.positional_embedding", (self.grid ** 2 + 1, vw)),
                      ("visual.class_embedding", (vw,)), ("visual.conv1.weight", (vw, 3, c["patch"], c["patch"]))]
        # text tower
        order += [("ln_final.weight", (tw,)), ("ln_final.bias", (tw,)), ("text_projection", (tw, e))]
        for i in reversed(range(c["t_layers"])):
            order += [(f"transformer.resblocks.{i}.{k}", s) for k, s in _block_shapes(tw).items()]
        order += [("positional_embedding", (c["ctx"], tw)), ("token_embedding.weight", (c["vocab"], tw))]
        self.slots = OrderedDict()
        off = 0
        for name, shape in order:
            n = math.prod(shape)
            self.slots[name] = (off, shape)
            off += ((n + 63) // 64) * 64
        self.n_trainable = off
        # logit_scale exists in open_clip but never receives a gradient in this training loop (reference never calls
        # model.forward); torch's AdamW skips grad-less parameters, so it sits outside the optimised range.
        self.slots["logit_scale"] = (off, ())
        self.n_total = off + 64
        # bucket boundaries (contiguous slices that become final together during the backward)
        self.buckets = []
        names = list(self.slots)

        def span(first, last):
            a = self.slots[first][0]
            o, s = self.slots[last]
            return (a, o + ((math.prod(s) + 63) // 64) * 64)

        if self.rn is not None:
            for bname, first, last in self.rn.bucket_names():
                self.buckets.append((bname, span(first, last)))
        else:
            self.buckets.append(("visual.head", span("visual.ln_post.weight", "visual.proj")))
            for i in reversed(range(c["v_layers"])):
                p = f"visual.transformer.resblocks.{i}."
                self.buckets.append((p, span(p + BLOCK_PARAMS[0], p + BLOCK_PARAMS[-1])))
            self.buckets.append(("visual.stem", span("visual.ln_pre.weight", "visual.conv1.weight")))
        self.buckets.append(("text.head", span("ln_final.weight", "text_projection")))
        for i in reversed(range(c["t_layers"])):
            p = f"transformer.resblocks.{i}."
            self.buckets.append((p, span(p + BLOCK_PARAMS[0], p + BLOCK_PARAMS[-1])))
        self.buckets.append(("text.stem", span("positional_embedding", "token_embedding.weight")))
        assert names[-1] == "logit_scale"

    def _view(self, flat, name):
        off, shape = self.slots[name]
        n = math.prod(shape)
        return flat[off:off + n].view(shape)

    def param(self, name):
        return self._view(self.flat, name)

    def grad(self, name):
        return self._view(self.flat_grad, name)

    def gemm_weight(self, name):
        """GEMM operand view in the compute dtype (bf16 shadow or the fp32 master)."""
        return self._view(self.flat_bf16 if self.flat_bf16 is not None else self.flat, name)

    # ------------------------------------------------------------------------------------------ parameters
    def init_parameters(self, seed=0):
        """open_clip's initialisers drawn on the host from a seeded generator, then uploaded."""
        g = torch.Generator().manual_seed(seed)
        c = self.cfg

        def normal(shape, std):
            return torch.randn(shape, generator=g) * std

        def uniform(shape, bound):
            return (torch.rand(shape, generator=g) * 2 - 1) * bound

        sd = {}
        vw, tw = c["v_width"], c["t_width"]
        if self.rn is not None:
            self.rn.init_parameters(sd, g)
            sd["ln_final.weight"], sd["ln_final.bias"] = torch.ones(tw), torch.zeros(tw)
        else:
            self._init_vit_visual(sd, normal, uniform)
        sd["token_embedding.weight"] = normal((c["vocab"], tw), 0.02)
        sd["positional_embedding"] = normal((c["ctx"], tw), 0.01)
        proj_std, attn_std, fc_std = (tw ** -0.5) * ((2 * c["t_layers"]) ** -0.5), tw ** -0.5, (2 * tw) ** -0.5
        for i in range(c["t_layers"]):
            p = f"transformer.resblocks.{i}."
            sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"] = normal((3 * tw, tw), attn_std), torch.zeros(3 * tw)
            sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"] = normal((tw, tw), proj_std), torch.zeros(tw)
            sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"] = normal((4 * tw, tw), fc_std), uniform((4 * tw,), 1 / math.sqrt(tw))
            sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"] = normal((tw, 4 * tw), proj_std), uniform((tw,), 1 / math.sqrt(4 * tw))
            for ln in ["ln_1", "ln_2"]:
                sd[p + ln + ".weight"], sd[p + ln + ".bias"] = torch.ones(tw), torch.zeros(tw)
        sd["text_projection"] = normal((tw, c["embed_dim"]), tw ** -0.5)
        sd["logit_scale"] = torch.tensor(math.log(1 / 0.07))
        self.load_state_dict(sd)

    def _init_vit_visual(self, sd, normal, uniform):
        c = self.cfg
        vw, tw = c["v_width"], c["t_width"]
        sd["visual.class_embedding"] = normal((vw,), vw ** -0.5)
        sd["visual.positional_embedding"] = normal((self.grid ** 2 + 1, vw), vw ** -0.5)
        sd["visual.proj"] = normal((vw, c["embed_dim"]), vw ** -0.5)
        fan = self.k_patch
        sd["visual.conv1.weight"] = uniform((vw, 3, c["patch"], c["patch"]), 1 / math.sqrt(fan))   # kaiming_uniform(a=sqrt 5)
        for ln in ["visual.ln_pre", "visual.ln_post", "ln_final"]:
            w = vw if ln.startswith("visual") else tw
            sd[ln + ".weight"], sd[ln + ".bias"] = torch.ones(w), torch.zeros(w)
        for i in range(c["v_layers"]):     # vision blocks: torch defaults (xavier in_proj, kaiming-uniform linears)
            p = f"visual.transformer.resblocks.{i}."
            sd[p + "attn.in_proj_weight"] = uniform((3 * vw, vw), math.sqrt(6.0 / (3 * vw + vw)))
            sd[p + "attn.in_proj_bias"] = torch.zeros(3 * vw)
            sd[p + "attn.out_proj.weight"] = uniform((vw, vw), 1 / math.sqrt(vw))
            sd[p + "attn.out_proj.bias"] = torch.zeros(vw)
            sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"] = uniform((4 * vw, vw), 1 / math.sqrt(vw)), uniform((4 * vw,), 1 / math.sqrt(vw))
            sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"] = uniform((vw, 4 * vw), 1 / math.sqrt(4 * vw)), uniform((vw,), 1 / math.sqrt(4 * vw))
            for ln in ["ln_1", "ln_2"]:
                sd[p + ln + ".weight"], sd[p + ln + ".bias"] = torch.ones(vw), torch.zeros(vw)

    def named_parameters(self):
        for name in self.slots:
            yield name, self.param(name)

    def parameters(self):
        return [p for _, p in self.named_parameters()]

    def state_dict(self, prefix=""):
        """Flat {open_clip key: tensor}; pass prefix='module.' for the reference's DataParallel checkpoints (:983, :1118)."""
        sd = OrderedDict((prefix + n, p.detach().clone()) for n, p in self.named_parameters())
        for n, b in self.buffers.items():
            sd[prefix + n] = b.detach().clone()
        return sd

    def load_state_dict(self, sd, strict=True):
        seen = set()
        for key, val in sd.items():
            name = key[len("module."):] if key.startswith("module.") else key
            if name in self.buffers:
                self.buffers[name].copy_(val.to(device=self.device, dtype=self.buffers[name].dtype))
                continue
            if name not in self.slots:
                if strict:
                    raise ScError(f"unexpected key {key!r} in state_dict")
                continue
            dst = self.param(name)
            if tuple(val.shape) != tuple(dst.shape):
                raise ScError(f"shape mismatch for {key!r}: {tuple(val.shape)} vs {tuple(dst.shape)}")
            dst.copy_(val.to(device=self.device, dtype=torch.float32))
            seen.add(name)
        missing = [n for n in self.slots if n not in seen]
        if strict and missing:
            raise ScError(f"missing keys in state_dict: {missing[:4]}{'...' if len(missing) > 4 else ''}")
        self.refresh_shadows(full=True)

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def zero_grad(self):
        """The next backward overwrites the gradient buffer instead of accumulating (saves a 605 MB memset)."""
        self._grads_fresh = True

    def refresh_shadows(self, full=False, overlap=False):
        """bf16 mode: rebuild the [in,out] weight copies (and, if `full`, the bf16 shadow itself) from the masters.
        `overlap`: the ~100 small transposes are only read by the NEXT backward pass, so they go to an auxiliary stream (ordered
        behind everything enqueued so far) and run under the next forward pass; `_blocks_bwd` and the next optimiser step wait
        for `wt_ready`."""
        if self.rn is not None:
            self.rn.refresh_weights()      # GEMM-layout copies of the convolution weights (both precisions)
        if self.flat_bf16 is None:
            return
        if full:
            ops.cast_bf16(self.flat, self.flat_bf16)
        if overlap and self.device.type == "cuda":
            if self._aux_stream is None:
                self._aux_stream = torch.cuda.Stream(device=self.device)
            self._aux_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._aux_stream):
                self._rebuild_wt()
                self.wt_ready = torch.cuda.Event()
                self.wt_ready.record(self._aux_stream)
            return
        self._rebuild_wt()

    def wait_wt(self):
        """Order the current stream behind the last overlapped rebuild of the [in,out] weight copies."""
        if self.wt_ready is not None:
            torch.cuda.current_stream().wait_event(self.wt_ready)

    def _rebuild_wt(self):
        """bf16 [in,out] copies of every block GEMM weight from the fp32 masters: one batched launch over a pointer table."""
        if self._wt_batch is None:
            pairs = []
            for tower in (self.visual, self.text):
                if tower is None:
                    continue
                for i in range(tower.layers):
                    for k in GEMM_WEIGHTS:
                        name = f"{tower.prefix}{i}.{k}"
                        src = self.param(name)
                        dst = self.wt[name] = torch.empty(src.shape[1], src.shape[0], dtype=torch.bfloat16, device=self.device)
                        pairs.append((src, dst))
            self._wt_batch = ops.transpose_cast_bf16_batch(pairs)
        self._wt_batch()

    # ------------------------------------------------------------------------------------------ buffers
    def _buf(self, key, shape, dtype):
        t = self._scratch.get(key)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            t = self._scratch[key] = torch.empty(shape, dtype=dtype, device=self.device)
        return t

    def _prepare(self, tower: _Tower, batch: int):
        if tower.batch == batch and tower.descs:
            return
        T, dev = self.dtype, self.device
        rows, w, mlp = batch * tower.seq, tower.width, 4 * tower.width
        f32 = torch.float32
        b = tower.bufs = {}
        b["x"] = [torch.empty(rows, w, dtype=f32, device=dev) for _ in range(tower.layers + 1)]
        for key, cols, dt in [("ln1_out", w, T), ("qkv", 3 * w, T), ("attn_out", w, T), ("ln2_out", w, T), ("h_pre", mlp, T), ("h_act", mlp, T),
                              ("x_mid", w, f32)]:
            b[key] = [torch.empty(rows, cols, dtype=dt, device=dev) for _ in range(tower.layers)]
        for key in ["ln1_mean", "ln1_rstd", "ln2_mean", "ln2_rstd"]:
            b[key] = [torch.empty(rows, dtype=f32, device=dev) for _ in range(tower.layers)]
        # softmax statistics of every (image, head, query) for the flash-attention form of the backward (long sequences: ViT-L/14's 257 tokens)
        b["attn_lse"] = ([torch.empty(batch * tower.heads * tower.seq, dtype=f32, device=dev) for _ in range(tower.layers)]
                         if ops.attention_uses_stats(T, tower.seq) else None)
        # backward scratch shared by both towers (sized for the larger request)
        ws_bytes = ops.block_workspace_bytes(rows, w, mlp, T)
        # the weight-gradient side stream of block k may still read d_h / d_qkv / d_res_t / dx_t while block k+1 runs on the main
        # stream: those live in two (dx_t: three) rotating sets, the rest is shared
        sc = {"d_ln": (rows * w, T), "d_attn": (rows * w, T), "dx_mid": (rows * w, f32), "dx": (rows * w, f32), "ws": (ws_bytes, torch.uint8),
              "ws_side": (ws_bytes, torch.uint8)}
        for k in range(2):
            sc.update({f"d_h.{k}": (rows * mlp, T), f"d_qkv.{k}": (rows * 3 * w, T), f"d_res_t.{k}": (rows * w, T)})
        for k in range(3):
            sc[f"dx_t.{k}"] = (rows * w, T)
        for key, (n, dt) in sc.items():       # one scratch family per tower: their backward passes run on different streams
            cur = self._scratch.get(f"bwd.{tower.kind}.{key}")
            if cur is None or cur.numel() < n or cur.dtype != dt:
                self._scratch[f"bwd.{tower.kind}.{key}"] = torch.empty(n, dtype=dt, device=dev)
        tower.batch = batch
        tower.descs = []
        tower.trimmed = {}
        tower.run_descs, tower.run_seq = None, tower.seq
        if self.dtype == torch.bfloat16 and tower.events is None:
            tower.events = EventSet()    # the blocks of one tower are enqueued one after the other from this thread: one set per tower
            tower.tickets = torch.zeros(16, dtype=torch.int32, device=dev)   # sc_block_desc.tile_tickets: zeroed once, the kernels leave it zero
        for i in range(tower.layers):
            d = BlockDesc()
            if tower.events is not None:
                tower.events.bind(d)
                d.tile_tickets = tower.tickets.data_ptr()
            d.batch, d.seq, d.width, d.heads, d.mlp_width = batch, tower.seq, w, tower.heads, mlp
            d.dtype, d.causal = sc_dtype(T), tower.causal
            p = f"{tower.prefix}{i}."
            for field, pname in [("ln1_g", "ln_1.weight"), ("ln1_b", "ln_1.bias"), ("b_qkv", "attn.in_proj_bias"), ("b_o", "attn.out_proj.bias"),
                                 ("ln2_g", "ln_2.weight"), ("ln2_b", "ln_2.bias"), ("b_fc1", "mlp.c_fc.bias"), ("b_fc2", "mlp.c_proj.bias")]:
                setattr(d, field, self.param(p + pname).data_ptr())
            for field, pname in [("w_qkv", "attn.in_proj_weight"), ("w_o", "attn.out_proj.weight"), ("w_fc1", "mlp.c_fc.weight"),
                                 ("w_fc2", "mlp.c_proj.weight")]:
                setattr(d, field, self.gemm_weight(p + pname).data_ptr())
                setattr(d, "wt_" + field[2:], self.wt[p + pname].data_ptr() if self.flat_bf16 is not None else None)
            d.x_in, d.x_out, d.x_mid = b["x"][i].data_ptr(), b["x"][i + 1].data_ptr(), b["x_mid"][i].data_ptr()
            for field in ["ln1_out", "qkv", "attn_out", "ln2_out", "h_pre", "h_act", "ln1_mean", "ln1_rstd", "ln2_mean", "ln2_rstd"]:
                setattr(d, field, b[field][i].data_ptr())
            d.attn_lse = b["attn_lse"][i].data_ptr() if b["attn_lse"] is not None else None
            for field, pname in [("g_ln1_g", "ln_1.weight"), ("g_ln1_b", "ln_1.bias"), ("g_w_qkv", "attn.in_proj_weight"),
                                 ("g_b_qkv", "attn.in_proj_bias"), ("g_w_o", "attn.out_proj.weight"), ("g_b_o", "attn.out_proj.bias"),
                                 ("g_ln2_g", "ln_2.weight"), ("g_ln2_b", "ln_2.bias"), ("g_w_fc1", "mlp.c_fc.weight"), ("g_b_fc1", "mlp.c_fc.bias"),
                                 ("g_w_fc2", "mlp.c_proj.weight"), ("g_b_fc2", "mlp.c_proj.bias")]:
                setattr(d, field, self.grad(p + pname).data_ptr())
            tower.descs.append(d)
        # the c_proj bias gradient of block i-1 is the column sum of block i's dx_in: fused into block i's ln_1 backward
        for i in range(1, tower.layers):
            tower.descs[i].g_below_b_fc2 = self.grad(f"{tower.prefix}{i - 1}.mlp.c_proj.bias").data_ptr()
            tower.descs[i - 1].b_fc2_done = 1
        self._bind_scratch()

    def activation_set(self, j: int):
        """Switch both ViT towers to activation buffer set j (allocated by the next forward when new).  A forward leaves everything its
        backward needs in the current set, so several micro-batches can be forwarded one after the other and back-propagated later, each
        from its own set (Trainer.step_cached keeps as many resident as the card's memory allows: ~35 GB per 1024 pairs of ViT-B/32)."""
        if self.rn is not None:
            raise ScError("activation sets exist for the ViT towers only")
        if j == self._act_set:
            return
        for tower in (self.visual, self.text):
            tower.sets[self._act_set] = (tower.bufs, tower.descs, tower.batch, tower.run_seq)
            bufs, descs, batch, run_seq = tower.sets.get(j, ({}, [], 0, tower.seq))
            tower.bufs, tower.descs, tower.batch = bufs, descs, batch
            tower.trimmed, tower.run_descs, tower.run_seq = {}, None, tower.seq
        self._act_set = j
        self._bind_scratch()      # the backward scratch may have been re-allocated since this set's descriptors were bound
        for tower in (self.visual, self.text):
            state = tower.sets.get(j)
            if state is not None and tower.descs:
                self._use_seq(tower, state[3])

    def drop_activation_sets(self):
        """Free every activation set but the current one."""
        for tower in (self.visual, self.text):
            if tower is not None:
                tower.sets = {}

    def activation_set_bytes(self):
        """Device bytes of the current activation set (both towers), 0 before the first forward."""
        total = 0
        for tower in (self.visual, self.text):
            for v in (tower.bufs.values() if tower is not None else ()):
                for t in (v if isinstance(v, (list, tuple)) else [v]):
                    if torch.is_tensor(t):
                        total += t.numel() * t.element_size()
        return total

    def _bind_scratch(self):
        for tower in (self.visual, self.text):
            if tower is None or not tower.descs:
                continue
            tower.trimmed = {}
            for i, d in enumerate(tower.descs):
                k = (tower.layers - 1 - i) % 2          # position in the backward order selects the scratch set
                fam = f"bwd.{tower.kind}."
                for key in ["d_ln", "d_attn", "dx_mid", "ws", "ws_side"]:
                    setattr(d, key, self._scratch[fam + key].data_ptr())
                for key in ["d_h", "d_qkv", "d_res_t"]:
                    setattr(d, key, self._scratch[f"{fam}{key}.{k}"].data_ptr())
                d.ws_bytes = self._scratch[fam + "ws"].numel()
                d.ws_side_bytes = self._scratch[fam + "ws_side"].numel()

    # ------------------------------------------------------------------------------------------ towers
    def _linear(self, x, w):
        """x [M,K] (compute dtype) times torch-layout weight w [N,K]."""
        if self.dtype == torch.bfloat16:
            return ops.gemm_bf16_nt(x, w)
        return ops.gemm_f32(x, w, trans_b=True)

    def _use_seq(self, tower, seq):
        """Select the descriptors of a forward / backward over `seq` <= tower.seq positions per sample: the same buffers (a
        [batch * seq, W] prefix of each), copies of the descriptors with the shorter sequence length."""
        if seq == tower.seq:
            tower.run_descs, tower.run_seq = tower.descs, seq
            return
        if seq not in tower.trimmed:
            import ctypes
            copies = []
            for d in tower.descs:
                c = BlockDesc()
                ctypes.memmove(ctypes.byref(c), ctypes.byref(d), ctypes.sizeof(BlockDesc))
                c.seq = seq
                copies.append(c)
            tower.trimmed[seq] = copies
        tower.run_descs, tower.run_seq = tower.trimmed[seq], seq

    def _blocks_fwd(self, tower):
        for d in (tower.run_descs or tower.descs):
            ops.block_fwd(d)

    def _blocks_bwd(self, tower, dx, acc):
        """dx: fp32 [rows,W] gradient w.r.t. the last block's output; returns gradient w.r.t. the first block's input (in place).

        bf16: the weight-gradient GEMMs of block k run on a side stream (sc_block_bwd_async) and overlap the HBM-bound kernels
        of block k+1 on the main stream; block k+2 reuses block k's scratch set, so it first waits for block k's side work
        (that is also the moment block k's gradient bucket is handed to the all-reduce)."""
        bf = self.dtype == torch.bfloat16
        if not bf:
            for i in reversed(range(tower.layers)):
                d = (tower.run_descs or tower.descs)[i]
                d.accumulate = int(acc)
                ops.block_bwd(d, dx, None, dx, None)
                if self.comm is not None:
                    self.comm.bucket_ready(f"{tower.prefix}{i}.")
            return dx
        main = torch.cuda.current_stream()
        self.wait_wt()
        if self._side is None:
            self._side = {}
        if tower.kind not in self._side:
            self._side[tower.kind] = (torch.cuda.Stream(device=self.device), [torch.cuda.Event(), torch.cuda.Event()])
        side, side_done = self._side[tower.kind]
        n = dx.numel()
        dx_t = [self._scratch[f"bwd.{tower.kind}.dx_t.{k}"][:n] for k in range(3)]
        ops.cast_bf16(dx, dx_t[0])
        side.wait_stream(main)                            # everything the side stream will read exists
        pending = []                                      # (step, layer) whose side work has not been joined yet
        for step, i in enumerate(reversed(range(tower.layers))):
            if step >= 2:
                main.wait_event(side_done[step % 2])                 # block (step-2) is done with this scratch set
                s0, layer0 = pending.pop(0)
                if self.comm is not None:
                    self.comm.bucket_ready(f"{tower.prefix}{layer0}.")
            d = (tower.run_descs or tower.descs)[i]
            d.accumulate = int(acc)
            ops.block_bwd(d, dx, dx_t[step % 3], dx, dx_t[(step + 1) % 3], side_stream=side)
            side_done[step % 2].record(side)
            pending.append((step, i))
        main.wait_stream(side)
        for _, layer0 in pending:
            if self.comm is not None:
                self.comm.bucket_ready(f"{tower.prefix}{layer0}.")
        return dx

    def image_forward(self, images):
        c, tw = self.cfg, self.visual
        if images.dim() != 4 or images.shape[1] != 3 or images.shape[2] != c["image_size"] or images.shape[3] != c["image_size"]:
            raise ScError(f"encode_image expects [B,3,{c['image_size']},{c['image_size']}], got {tuple(images.shape)}")
        images = images.to(device=self.device, dtype=torch.float32).contiguous()
        if self.rn is not None:
            return self.rn.forward(images)
        batch = images.shape[0]
        self._prepare(tw, batch)
        b = tw.bufs
        b["patches"] = ops.im2col(images, c["patch"], self.k_pad, self.dtype)
        b["patch_out"] = self._linear(b["patches"], self._conv_weight())
        b["x_pre"] = ops.vit_tokens_fwd(b["patch_out"], self.param("visual.class_embedding"), self.param("visual.positional_embedding"), batch, tw.seq)
        _, b["pre_mean"], b["pre_rstd"] = ops.layernorm_fwd(b["x_pre"], self.param("visual.ln_pre.weight"), self.param("visual.ln_pre.bias"),
                                                            torch.float32, out=b["x"][0])
        self._blocks_fwd(tw)
        b["pooled"] = ops.pool_gather(b["x"][-1], None, batch, tw.seq)
        b["pooled_ln"], b["post_mean"], b["post_rstd"] = ops.layernorm_fwd(b["pooled"], self.param("visual.ln_post.weight"),
                                                                           self.param("visual.ln_post.bias"), torch.float32)
        return ops.gemm_f32(b["pooled_ln"], self.param("visual.proj"))

    def _conv_weight(self):
        w = self.gemm_weight("visual.conv1.weight").view(self.cfg["v_width"], self.k_patch)
        if self.k_pad == self.k_patch:
            return w
        padded = self._buf("conv_pad", (self.cfg["v_width"], self.k_pad), w.dtype)
        padded.zero_()
        padded[:, : self.k_patch].copy_(w)
        return padded

    def image_backward(self, d_emb):
        if self.rn is not None:
            self.wait_wt()
            return self.rn.backward(d_emb, not self._grads_fresh_for("image"))
        tw, b = self.visual, self.visual.bufs
        acc = not self._grads_fresh_for("image")
        batch = tw.batch
        d_emb = d_emb.to(torch.float32).contiguous()
        ops.gemm_f32(b["pooled_ln"], d_emb, trans_a=True, out=self.grad("visual.proj"), epi=ops.make_epilogue(beta=1.0 if acc else 0.0))
        d_pln = ops.gemm_f32(d_emb, self.param("visual.proj"), trans_b=True)
        d_pooled, _, _, _ = ops.layernorm_bwd(d_pln, b["pooled"], b["post_mean"], b["post_rstd"], self.param("visual.ln_post.weight"),
                                              dgamma=self.grad("visual.ln_post.weight"), dbeta=self.grad("visual.ln_post.bias"), accumulate=acc)
        if self.comm is not None:
            self.comm.bucket_ready("visual.head")
        dx = self._scratch["bwd.image.dx"][: batch * tw.seq * tw.width].view(batch * tw.seq, tw.width)
        dx.zero_()
        ops.pool_scatter(d_pooled, None, batch, tw.seq, dx)
        self._blocks_bwd(tw, dx, acc)
        d_pre, _, _, _ = ops.layernorm_bwd(dx, b["x_pre"], b["pre_mean"], b["pre_rstd"], self.param("visual.ln_pre.weight"),
                                           dgamma=self.grad("visual.ln_pre.weight"), dbeta=self.grad("visual.ln_pre.bias"), accumulate=acc)
        d_patch = ops.vit_tokens_bwd(d_pre, batch, tw.seq, self.dtype, self.grad("visual.class_embedding"), self.grad("visual.positional_embedding"), acc)
        gw = self.grad("visual.conv1.weight").view(self.cfg["v_width"], self.k_patch)
        bf, beta = self.dtype == torch.bfloat16, 1.0 if acc else 0.0
        if self.k_pad == self.k_patch:
            if bf:
                ops.gemm_bf16_tn(d_patch, b["patches"], out=gw, beta=beta)
            else:
                ops.gemm_f32(d_patch, b["patches"], trans_a=True, out=gw, epi=ops.make_epilogue(beta=beta))
        else:   # ViT-L/14: K = 588 is padded to 640 for the GEMM; drop the padding columns (index bookkeeping only)
            full = ops.gemm_bf16_tn(d_patch, b["patches"]) if bf else ops.gemm_f32(d_patch, b["patches"], trans_a=True)
            gw.add_(full[:, : self.k_patch]) if acc else gw.copy_(full[:, : self.k_patch])
        if self.comm is not None:
            self.comm.bucket_ready("visual.stem")

    def text_forward(self, tokens, seq_len=None):
        """seq_len (optional, host int): the batch's longest caption in tokens, EOT included.  Under the causal mask nothing at or
        before a caption's EOT sees the positions behind it, the tower's output is taken AT the EOT and the positions behind it
        receive an exactly-zero gradient, so running the tower over the first `seq_len` positions only gives the same embeddings and
        the same parameter gradients (up to fp32 summation order) as running it over all 77 - with 77 / seq_len times fewer rows in
        every text GEMM.  The caller must know the length on the host (the tokenizer does); None = all positions, as the reference."""
        c, tw = self.cfg, self.text
        if tokens.dim() != 2 or tokens.shape[1] != c["ctx"]:
            raise ScError(f"encode_text expects int64 [B,{c['ctx']}], got {tuple(tokens.shape)}")
        tokens = tokens.to(device=self.device, dtype=torch.int64).contiguous()
        batch = tokens.shape[0]
        self._prepare(tw, batch)
        seq = tw.seq
        if seq_len is not None:
            if not 1 <= int(seq_len) <= tw.seq:
                raise ScError(f"encode_text: seq_len {seq_len} outside 1..{tw.seq}")
            seq = min(tw.seq, (int(seq_len) + 7) // 8 * 8)      # a few distinct lengths only (descriptor copies are cached per length)
            if seq < tw.seq:
                tokens = tokens[:, :seq].contiguous()
        self._use_seq(tw, seq)
        b = tw.bufs
        b["tokens"] = tokens
        ops.text_embed_fwd(tokens, self.param("token_embedding.weight"), self.param("positional_embedding"), out=b["x"][0])
        self._blocks_fwd(tw)
        b["eot"] = ops.argmax_tokens(tokens)
        if self.training:   # bookkeeping of the backward pass, done here so that it runs under the forward GEMMs instead of at the tail of the step
            b["sorted_keys"], b["sort_order"] = self._token_sort(tokens, b["eot"], seq)
        else:
            b.pop("sorted_keys", None)
        b["pooled"] = ops.pool_gather(b["x"][-1], b["eot"], batch, seq)
        b["pooled_ln"], b["post_mean"], b["post_rstd"] = ops.layernorm_fwd(b["pooled"], self.param("ln_final.weight"), self.param("ln_final.bias"),
                                                                           torch.float32)
        return ops.gemm_f32(b["pooled_ln"], self.param("text_projection"))

    def _token_sort(self, tokens, eot, seq):
        """Sorted token ids + permutation for the token-embedding scatter-add (sc_token_sort: keys and stable radix sort on the device,
        no host sync - a .nonzero() would stall the enqueue of the other tower).  Positions after EOT carry an exactly-zero gradient
        under the causal mask and get the key `vocab` (sorts to the end, ignored by the scatter kernel)."""
        return ops.token_sort(tokens, eot, self.cfg["vocab"])

    def text_backward(self, d_emb):
        tw, b = self.text, self.text.bufs
        acc = not self._grads_fresh_for("text")
        batch = tw.batch
        d_emb = d_emb.to(torch.float32).contiguous()
        ops.gemm_f32(b["pooled_ln"], d_emb, trans_a=True, out=self.grad("text_projection"), epi=ops.make_epilogue(beta=1.0 if acc else 0.0))
        d_pln = ops.gemm_f32(d_emb, self.param("text_projection"), trans_b=True)
        d_pooled, _, _, _ = ops.layernorm_bwd(d_pln, b["pooled"], b["post_mean"], b["post_rstd"], self.param("ln_final.weight"),
                                              dgamma=self.grad("ln_final.weight"), dbeta=self.grad("ln_final.bias"), accumulate=acc)
        if self.comm is not None:
            self.comm.bucket_ready("text.head")
        seq = tw.run_seq
        dx = self._scratch["bwd.text.dx"][: batch * seq * tw.width].view(batch * seq, tw.width)
        dx.zero_()
        ops.pool_scatter(d_pooled, b["eot"], batch, seq, dx)
        self._blocks_bwd(tw, dx, acc)
        # token-embedding scatter-add in a fixed order (the index sort is bookkeeping done in text_forward, the fp32 sums are the
        # HIP kernel's)
        if "sorted_keys" not in b:   # forward ran in eval mode
            b["sorted_keys"], b["sort_order"] = self._token_sort(b["tokens"], b["eot"], seq)
        st, order = b["sorted_keys"], b["sort_order"]
        d_pos = self.grad("positional_embedding")
        if seq < tw.seq and not acc:
            d_pos[seq:].zero_()      # positions the trimmed batch never reaches: gradient exactly zero
        ops.text_embed_bwd(dx, st, order, batch, seq, self.grad("token_embedding.weight"), d_pos, acc)
        if self.comm is not None:
            self.comm.bucket_ready("text.stem")

    def _grads_fresh_for(self, which):
        """True if this tower's gradients should overwrite; flips to accumulate once both towers ran."""
        if self._grads_fresh is True:
            self._grads_fresh = {"image": True, "text": True}
        if isinstance(self._grads_fresh, dict):
            fresh = self._grads_fresh.get(which, False)
            self._grads_fresh[which] = False
            return fresh
        return False

    # ------------------------------------------------------------------------------------------ reference call surface
    def encode_image(self, images):
        """[B,3,224,224] -> [B,E] fp32, differentiable through torch autograd (reference :768)."""
        return _TowerFn.apply(self, "image", images, _anchor(self.device))

    def encode_text(self, tokens):
        """int64 [B,77] -> [B,E] fp32 (reference :769)."""
        return _TowerFn.apply(self, "text", tokens, _anchor(self.device))


_anchors = {}


def _anchor(device):
    a = _anchors.get(str(device))
    if a is None:
        a = _anchors[str(device)] = torch.zeros((), device=device, requires_grad=True)
    return a


class _TowerFn(torch.autograd.Function):
    """Autograd bridge: forward runs the HIP tower, backward runs the HIP backward into model.flat_grad."""

    @staticmethod
    def forward(ctx, model, which, inp, anchor):
        ctx.model, ctx.which = model, which
        return model.image_forward(inp) if which == "image" else model.text_forward(inp)

    @staticmethod
    def backward(ctx, d_emb):
        (ctx.model.image_backward if ctx.which == "image" else ctx.model.text_backward)(d_emb)
        return None, None, None, None


def create_model_and_transforms(model_name, pretrained=None, device="cuda:0", precision="bf16", seed=0):
    """Mirror of the reference's factory call (sparsify_clip.py:685-689): returns (model, None, None) - the image
    transforms belong to the input pipeline, which is synthetic in this build."""
    if pretrained:
        raise ScError("pretrained weights cannot be fetched offline; load a state_dict instead")
    return ClipModel(model_name, device=device, precision=precision, seed=seed), None, None
