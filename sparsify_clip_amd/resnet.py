"""open_clip ModifiedResNet ("RN50") image tower on the HIP library (reference: every experiments_configs/*.yaml says `model: "RN50"`,
instantiated at sparsify_clip.py:685-689 and called at :768).  open_clip is absent offline: the architecture is the published one
(oracle/clip_model.py restates it in torch.nn; PARITY UNPINNED like the ViT towers).

Layout: activations are NHWC, i.e. row-major [B*H*W, C] matrices in the compute dtype, so
  * a 1x1 convolution is the NT GEMM of the library as it stands (weight [Cout, Cin]),
  * a 3x3 convolution is sc_im2col3x3 + the same GEMM against the weight permuted to [Cout, 3, 3, Cin],
  * BatchNorm runs in training mode (batch statistics, fused ReLU / residual join, running statistics updated as torch does);
    under data parallelism the statistics are exchanged, so that DP = k reproduces the single-process batch statistics of the
    reference (which is one process over the whole batch),
  * the attention pool runs the library's attention kernel over the HW + 1 tokens and keeps the mean token's row.
The backward is written out by hand (no autograd), like the ViT towers'.
"""
from __future__ import annotations

import math
import os

import torch

from . import dist as D
from . import ops
from ._lib import ScError


def _pad64(k):
    return (k + 63) // 64 * 64


class _Conv:
    """One bias-free convolution + its BatchNorm: parameter names, geometry."""

    def __init__(self, name, bn, cin, cout, ksize, stride=1):
        self.name, self.bn, self.cin, self.cout, self.k, self.stride = name, bn, cin, cout, ksize, stride
        self.kdim = cin * ksize * ksize


class ResNetVisual:
    def __init__(self, model, cfg):
        self.m = model
        self.width, self.layers, self.image_size, self.embed_dim = cfg["v_width"], tuple(cfg["v_layers"]), cfg["image_size"], cfg["embed_dim"]
        w = self.width
        self.heads = w * 32 // 64
        if self.image_size % 32 != 0:
            raise ScError("ModifiedResNet needs an image size that is a multiple of 32")
        self.stem = [_Conv("visual.conv1", "visual.bn1", 3, w // 2, 3, 2), _Conv("visual.conv2", "visual.bn2", w // 2, w // 2, 3),
                     _Conv("visual.conv3", "visual.bn3", w // 2, w, 3)]
        self.blocks = []     # (prefix, conv1, conv2, conv3, downsample conv or None, stride)
        inplanes = w
        for li, (planes, nblk) in enumerate(zip((w, 2 * w, 4 * w, 8 * w), self.layers)):
            for bi in range(nblk):
                stride = 2 if (li > 0 and bi == 0) else 1
                p = f"visual.layer{li + 1}.{bi}."
                down = None
                if stride > 1 or inplanes != planes * 4:
                    down = _Conv(p + "downsample.0", p + "downsample.1", inplanes, planes * 4, 1)
                self.blocks.append((p, _Conv(p + "conv1", p + "bn1", inplanes, planes, 1), _Conv(p + "conv2", p + "bn2", planes, planes, 3),
                                    _Conv(p + "conv3", p + "bn3", planes, planes * 4, 1), down, stride))
                inplanes = planes * 4
        self.cfeat = inplanes                      # 32 * width
        self.hw = (self.image_size // 32) ** 2
        self.saved = None
        self._halos = {}
        self.gw = {}                               # conv name -> (GEMM-layout weight [Cout, kpad], its transpose [kpad, Cout]) in the compute dtype

    # ------------------------------------------------------------------------------------------ parameters
    def convs(self):
        out = list(self.stem)
        for _, c1, c2, c3, down, _ in self.blocks:
            out += [c1, c2, c3] + ([down] if down is not None else [])
        return out

    def _conv_params(self, cv):
        return [(cv.name + ".weight", (cv.cout, cv.cin, cv.k, cv.k)), (cv.bn + ".weight", (cv.cout,)), (cv.bn + ".bias", (cv.cout,))]

    def param_order(self):
        """(name, shape) in reverse backward order: attention pool, blocks last to first, stem."""
        c = self.cfeat
        order = [("visual.attnpool.positional_embedding", (self.hw + 1, c))]
        for n in ("q_proj", "k_proj", "v_proj"):
            order += [(f"visual.attnpool.{n}.weight", (c, c)), (f"visual.attnpool.{n}.bias", (c,))]
        order += [("visual.attnpool.c_proj.weight", (self.embed_dim, c)), ("visual.attnpool.c_proj.bias", (self.embed_dim,))]
        for _, c1, c2, c3, down, _ in reversed(self.blocks):
            for cv in ([down] if down is not None else []) + [c3, c2, c1]:
                order += self._conv_params(cv)
        for cv in reversed(self.stem):
            order += self._conv_params(cv)
        return order

    def bucket_names(self):
        """(bucket name, first parameter, last parameter) in the order the backward completes them."""
        out = [("visual.head", "visual.attnpool.positional_embedding", "visual.attnpool.c_proj.bias")]
        for p, c1, c2, c3, down, _ in reversed(self.blocks):
            first = (down if down is not None else c3).name + ".weight"
            out.append((p, first, c1.bn + ".bias"))
        out.append(("visual.stem", self.stem[-1].name + ".weight", self.stem[0].bn + ".bias"))
        return out

    def buffer_specs(self):
        out = []
        for cv in self.convs():
            out += [(cv.bn + ".running_mean", (cv.cout,), 0.0, torch.float32), (cv.bn + ".running_var", (cv.cout,), 1.0, torch.float32),
                    (cv.bn + ".num_batches_tracked", (), 0, torch.int64)]
        return out

    def init_parameters(self, sd, g):
        """torch defaults for Conv2d / BatchNorm2d + open_clip's ModifiedResNet.init_parameters (attention-pool projections ~ N(0, C^-1),
        the last BatchNorm weight of every bottleneck zero)."""
        def uniform(shape, bound):
            return (torch.rand(shape, generator=g) * 2 - 1) * bound
        for cv in self.convs():
            sd[cv.name + ".weight"] = uniform((cv.cout, cv.cin, cv.k, cv.k), 1 / math.sqrt(cv.kdim))      # kaiming_uniform(a = sqrt 5)
            sd[cv.bn + ".weight"], sd[cv.bn + ".bias"] = torch.ones(cv.cout), torch.zeros(cv.cout)
        for _, _, _, c3, _, _ in self.blocks:
            sd[c3.bn + ".weight"] = torch.zeros(c3.cout)
        c = self.cfeat
        sd["visual.attnpool.positional_embedding"] = torch.randn(self.hw + 1, c, generator=g) / c ** 0.5
        for n, o in (("q_proj", c), ("k_proj", c), ("v_proj", c), ("c_proj", self.embed_dim)):
            sd[f"visual.attnpool.{n}.weight"] = torch.randn(o, c, generator=g) * c ** -0.5
            sd[f"visual.attnpool.{n}.bias"] = uniform((o,), 1 / math.sqrt(c))

    # ------------------------------------------------------------------------------------------ derived weights
    def refresh_weights(self):
        """GEMM-layout copies of the convolution weights in the compute dtype ([Cout, taps * Cin] padded to a multiple of 64 columns for the
        bf16 kernels) and their transposes (for the input gradients), and the packed attention-pool projection.  Data movement only."""
        m, dt = self.m, self.m.dtype
        for cv in self.convs():
            w = m.param(cv.name + ".weight")
            g = w.permute(0, 2, 3, 1).reshape(cv.cout, cv.kdim) if cv.k == 3 else w.reshape(cv.cout, cv.cin)
            kpad = self.kpad(cv)
            if kpad != cv.kdim:
                gp = torch.zeros(cv.cout, kpad, dtype=torch.float32, device=w.device)
                gp[:, : cv.kdim].copy_(g)
                g = gp
            g = g.to(dt).contiguous()
            if self.implicit(cv):   # tap-major over the bordered image's channels (zero columns for the padding channels); the input
                ci, co = _pad64(cv.cin), _pad64(cv.cout)      # gradient is the same implicit GEMM over the bordered output gradient
                g = torch.zeros(cv.cout, 3, 3, ci, dtype=torch.float32, device=w.device)
                g[..., : cv.cin] = w.permute(0, 2, 3, 1)
                gdx = torch.zeros(cv.cin, 3, 3, co, dtype=torch.float32, device=w.device)      # [cin][(8 - tap) * co + cout]
                gdx[..., : cv.cout] = w.flip(2, 3).permute(1, 2, 3, 0)
                self.gw[cv.name] = (g.reshape(cv.cout, 9 * ci).to(dt).contiguous(), gdx.reshape(cv.cin, 9 * co).to(dt).contiguous())
            else:
                self.gw[cv.name] = (g, g.t().contiguous())
        c = self.cfeat
        qkv = torch.cat([m.param(f"visual.attnpool.{n}.weight") for n in ("q_proj", "k_proj", "v_proj")], dim=0)
        self.gw["attnpool.qkv"] = (qkv.to(dt).contiguous(), qkv.t().to(dt).contiguous())
        self.gw["attnpool.qkv_bias"] = torch.cat([m.param(f"visual.attnpool.{n}.bias") for n in ("q_proj", "k_proj", "v_proj")]).contiguous()

    def kpad(self, cv):
        return _pad64(cv.kdim) if self.m.dtype == torch.bfloat16 else cv.kdim

    def implicit(self, cv):
        """A 3x3, stride-1 convolution of a bottleneck or of the stem (conv2, conv3) in bf16 whose channel counts, rounded up to 64, are
        64 * 2^j (a 32-channel stem activation travels inside a 64-channel image, the other half zero): the implicit-GEMM convolution
        (sc_conv3x3_bf16) on bordered activations - forward and input gradient without a patch matrix, the weight gradient as one TN
        GEMM over shifted views of the bordered input.  SC_RN_IM2COL=1 keeps the im2col path (A/B)."""
        def ok(c):
            p = _pad64(c) // 64
            return c % 8 == 0 and 2 * c >= 64 * p and p & (p - 1) == 0
        placed = (cv.name.endswith(".conv2") and ".layer" in cv.name) or cv.name in ("visual.conv2", "visual.conv3")
        return (placed and cv.k == 3 and cv.stride == 1 and self.m.dtype == torch.bfloat16 and ok(cv.cin) and ok(cv.cout)
                and ops.bn_mask_from_x(cv.cin) and os.environ.get("SC_RN_IM2COL", "0") != "1")
    def _nt(self, x, w, resid=None):            # x [R, K] @ w [N, K]^T (+ resid [R, N])
        epi = None
        if self.m.dtype == torch.bfloat16:
            k = x.shape[1]
            if k % 64 != 0:      # the stem's 32-channel gradients: zero columns up to the kernels' K granularity (data movement only)
                xp = torch.zeros(x.shape[0], _pad64(k), dtype=x.dtype, device=x.device)
                xp[:, :k].copy_(x)
                wp = torch.zeros(w.shape[0], _pad64(k), dtype=w.dtype, device=w.device)
                wp[:, :k].copy_(w)
                x, w = xp, wp
            if resid is not None:
                epi = ops.make_epilogue(resid=resid, ld_aux=w.shape[0])
            return ops.gemm_bf16_nt(x, w, epi=epi)
        if resid is not None:
            epi = ops.make_epilogue(resid=resid, ld_aux=w.shape[0])
        return ops.gemm_f32(x, w, trans_b=True, epi=epi)

    def _tn(self, dy, x):           # dy [R, M]^T @ x [R, N] -> fp32 [M, N]
        if self.m.dtype == torch.bfloat16:
            return ops.gemm_bf16_tn(dy, x)
        return ops.gemm_f32(dy, x, trans_a=True)

    # ------------------------------------------------------------------------------------------ one convolution + BatchNorm (+ ReLU / residual)
    def _conv_fwd(self, cv, x, batch, h, w, images=None):
        """x: NHWC rows [B*h*w, cin] (or `images` for the stem's first convolution).  -> conv output rows, output size."""
        g, _ = self.gw[cv.name]
        if cv.k == 3:
            cols = ops.im2col3x3(images if images is not None else x, batch, h, w, cv.cin, cv.stride, self.kpad(cv), self.m.dtype, nchw_images=images is not None)
            if self.m.training:
                self.saved["cols"][cv.name] = cols      # the weight gradient reads the same patch matrix (9x the activation: 6 GB at batch 256)
            ho, wo = (h - 1) // cv.stride + 1, (w - 1) // cv.stride + 1
            return self._nt(cols, g), ho, wo
        if self.m.dtype == torch.bfloat16 and cv.cin % 64 != 0:
            raise ScError(f"{cv.name}: the bf16 path needs channel counts that are multiples of 64 (got {cv.cin}); use --precision fp32")
        return self._nt(x, g), h, w

    def _bn_fwd(self, cv, z, relu, res=None, halo=None):
        rows, c = z.shape
        m = self.m
        if m.training:
            stats = ops.bn_stats(z)
            world = D.world_size() if D.active() else 1
            if world > 1:
                stats = D.exchange_packets(stats).reshape(-1).contiguous()  # synchronised BatchNorm: the reference normalises over the whole batch
            mean, rstd = ops.bn_finish(stats, world, c, rows, m.buffers[cv.bn + ".running_mean"], m.buffers[cv.bn + ".running_var"])
            m.buffers[cv.bn + ".num_batches_tracked"] += 1
        else:   # evaluation: the running statistics (sparsify_clip.py:540 model.eval())
            mean = m.buffers[cv.bn + ".running_mean"]
            rstd = torch.rsqrt(m.buffers[cv.bn + ".running_var"] + 1e-5)
        y = ops.bn_apply(z, mean, rstd, m.param(cv.bn + ".weight"), m.param(cv.bn + ".bias"), relu, res, halo=halo)
        return y, mean, rstd

    def _bn_bwd(self, cv, dy, y, z, mean, rstd, relu, acc, want_dres=False, halo=None):
        m = self.m
        rows = z.shape[0]
        gamma, beta = m.param(cv.bn + ".weight"), m.param(cv.bn + ".bias")
        if relu and not want_dres and (y is None or ops.bn_mask_from_x(cv.cout)):
            y = None         # no residual entered this BatchNorm: the ReLU mask is recomputed from z, one tensor less to read in both passes
        sums = ops.bn_bwd_stats(dy, y, z, mean, rstd, relu, gamma, beta)      # [2 C]: sum g, sum g xhat over this rank's rows
        world = D.world_size() if D.active() else 1
        g_w, g_b = m.grad(cv.bn + ".weight"), m.grad(cv.bn + ".bias")
        if world > 1:
            # The input gradient needs the WHOLE batch's sums; the affine parameters' gradients must stay this rank's share, because the
            # bucket all-reduce adds the ranks' gradient buffers afterwards (rounds 1-2 wrote the all-reduced sums into them: d gamma and
            # d beta came out `world` times too large under data parallelism - found by the gradient-level check of tests/test_gpu_dp.py).
            c = cv.cout
            if acc:
                ops.axpy_(g_b, 1.0, sums[:c])
                ops.axpy_(g_w, 1.0, sums[c:])
            else:
                g_b.copy_(sums[:c])
                g_w.copy_(sums[c:])
            D.all_reduce_sum_(sums)
            g_w = g_b = self._bn_grad_sink(c, sums.device)      # the kernel's own (whole-batch) parameter sums go nowhere
        return ops.bn_bwd_apply(dy, y, z, mean, rstd, gamma, sums, rows * world, relu, g_w, g_b, False if world > 1 else acc,
                                want_dres, beta=beta, halo=halo)

    def _bn_grad_sink(self, c, device):
        sink = getattr(self, "_sink", None)
        if sink is None or sink.numel() < c or sink.device != device:
            sink = self._sink = torch.empty(max(c, 2048), dtype=torch.float32, device=device)
        return sink[:c]

    def _conv_bwd(self, cv, dz, x_in, batch, h, w, acc, need_dx=True, images=None, resid=None):
        """dz: gradient of the convolution output rows; x_in: the convolution's input rows (or images).  Writes the weight gradient;
        returns the input gradient (+ resid, added in the GEMM epilogue: the shortcut's gradient at a bottleneck's entry)."""
        m = self.m
        g, gt = self.gw[cv.name]
        kpad = self.kpad(cv)
        if cv.k == 3:
            cols = self.saved["cols"].pop(cv.name, None)
            if cols is None:
                cols = ops.im2col3x3(images if images is not None else x_in, batch, h, w, cv.cin, cv.stride, kpad, m.dtype, nchw_images=images is not None)
            dwg = self._tn(dz, cols)[:, : cv.kdim].reshape(cv.cout, 3, 3, cv.cin).permute(0, 3, 1, 2)
            del cols
        else:
            dwg = self._tn(dz, x_in).reshape(cv.cout, cv.cin, 1, 1)
        gw = m.grad(cv.name + ".weight")
        if acc:
            ops.axpy_(gw, 1.0, dwg.contiguous())
        else:
            gw.copy_(dwg)                   # layout change only ([Cout, taps, Cin] -> [Cout, Cin, 3, 3])
        if not need_dx:
            return None
        dcols = self._nt(dz, gt, resid=resid if cv.k == 1 else None)            # [rows_out, kpad]
        if cv.k == 3:
            return ops.col2im3x3(dcols, batch, h, w, cv.cin, cv.stride, kpad)
        return dcols

    def _halo(self, tag, batch, h, w, c, device):
        """The bordered image (ops.halo_buffer) a BatchNorm apply writes for an implicit convolution, kept across steps: the kernels
        only ever write interior pixels and the first channels, so border, slack rows and padding channels stay zero and the
        110 MB memset per layer and step is paid once.  tag: the producing layer (its image lives until that layer's backward) or
        "grad<channels>" (an output-gradient image is consumed by the two launches right behind it on the same stream, so one per shape
        and real channel count - a 32-channel gradient must not inherit the upper channels of a 64-channel one)."""
        key = (tag, batch, h, w, c, self.m.dtype, str(device))
        buf = self._halos.get(key)
        if buf is None:
            if len(self._halos) > 256:      # a stream of different batch sizes: start over rather than grow
                self._halos.clear()
            buf = self._halos[key] = ops.halo_buffer(batch, h, w, c, self.m.dtype, device)
        return buf

    def _conv3x3_dw(self, cv, dz_img, x_flat, batch, h, w, acc):
        """Weight gradient of an implicit convolution, one TN GEMM whose B operand is the bordered input read through the nine tap shifts
        (sc_conv3x3_dw_bf16; the zero border of dz makes the border rows contribute nothing)."""
        co, ci = dz_img.shape[-1], x_flat.shape[-1]       # the images' channel counts (>= the convolution's)
        dwg = ops.conv3x3_dw_bf16(dz_img, x_flat, batch, h, w).view(co, 3, 3, ci)[: cv.cout, :, :, : cv.cin].permute(0, 3, 1, 2)
        gw = self.m.grad(cv.name + ".weight")
        if acc:
            ops.axpy_(gw, 1.0, dwg.contiguous())
        else:
            gw.copy_(dwg)                   # layout change only ([Cout, taps, Cin] -> [Cout, Cin, 3, 3])

    # ------------------------------------------------------------------------------------------ tower
    def forward(self, images):
        m = self.m
        batch = images.shape[0]
        if not self.gw:
            self.refresh_weights()
        S = self.saved = {"batch": batch, "images": images, "stem": [], "blocks": [], "cols": {}}
        h = w = self.image_size
        x = x_halo = None
        for i, cv in enumerate(self.stem):
            if self.implicit(cv):      # its input arrived as a bordered image
                z, ho, wo = ops.conv3x3_bf16(x_halo[1], self.gw[cv.name][0], batch, h, w), h, w
            else:
                z, ho, wo = self._conv_fwd(cv, x, batch, h, w, images=images if i == 0 else None)
            if i + 1 < len(self.stem) and self.implicit(self.stem[i + 1]):
                out_halo = self._halo(cv.name, batch, ho, wo, _pad64(cv.cout), z.device)
                y, (_, mean, rstd) = None, self._bn_fwd(cv, z, True, halo=(out_halo[1], ho, wo))
            else:
                out_halo = None
                y, mean, rstd = self._bn_fwd(cv, z, True)
            S["stem"].append((x, x_halo, z, y, mean, rstd, h, w))
            x, x_halo, h, w = y, out_halo, ho, wo
        S["stem_hw"] = (h, w)
        x = ops.avgpool_fwd(x, batch, h, w, self.width, 2)
        h, w = h // 2, w // 2
        for p, c1, c2, c3, down, stride in self.blocks:
            rec = {"x": x, "h": h, "w": w}
            z1, _, _ = self._conv_fwd(c1, x, batch, h, w)
            if self.implicit(c2):      # bn1 writes straight into the bordered image conv2 reads; no compact y1, no patch matrix
                flat1, img1 = self._halo(c1.name, batch, h, w, _pad64(c1.cout), z1.device)
                _, m1, r1 = self._bn_fwd(c1, z1, True, halo=(img1, h, w))
                y1 = None
                rec["y1_halo"] = flat1
                z2 = ops.conv3x3_bf16(img1, self.gw[c2.name][0], batch, h, w)
            else:
                y1, m1, r1 = self._bn_fwd(c1, z1, True)
                z2, _, _ = self._conv_fwd(c2, y1, batch, h, w)
            y2, m2, r2 = self._bn_fwd(c2, z2, True)
            a2 = ops.avgpool_fwd(y2, batch, h, w, c2.cout, stride) if stride > 1 else y2
            ho, wo = h // stride, w // stride
            z3, _, _ = self._conv_fwd(c3, a2, batch, ho, wo)
            if down is not None:
                xi = ops.avgpool_fwd(x, batch, h, w, c1.cin, stride) if stride > 1 else x
                zd, _, _ = self._conv_fwd(down, xi, batch, ho, wo)
                idn, md, rd = self._bn_fwd(down, zd, False)
                rec.update(xi=xi, zd=zd, md=md, rd=rd)
            else:
                idn = x
            y3, m3, r3 = self._bn_fwd(c3, z3, True, res=idn)
            rec.update(z1=z1, y1=y1, m1=m1, r1=r1, z2=z2, y2=y2, m2=m2, r2=r2, a2=a2, z3=z3, y3=y3, m3=m3, r3=r3)
            S["blocks"].append(rec)
            x, h, w = y3, ho, wo
        # attention pool
        c, hw, hd = self.cfeat, self.hw, 64
        if h * w != hw:
            raise ScError("ModifiedResNet: unexpected feature map size")
        tokens = ops.attnpool_tokens_fwd(x, m.param("visual.attnpool.positional_embedding"), batch, hw)
        qkv = self._linear_bias(tokens, self.gw["attnpool.qkv"][0], self.gw["attnpool.qkv_bias"])
        att = ops.attention_fwd(qkv, batch, hw + 1, self.heads, False)
        pooled = att.view(batch, hw + 1, c)[:, 0].to(torch.float32).contiguous()
        S.update(feat=x, tokens=tokens, qkv=qkv, pooled=pooled)
        return ops.gemm_f32(pooled, m.param("visual.attnpool.c_proj.weight"), trans_b=True,
                            epi=ops.make_epilogue(bias=m.param("visual.attnpool.c_proj.bias")))

    def _linear_bias(self, x, w, bias):
        if self.m.dtype == torch.bfloat16:
            return ops.gemm_bf16_nt(x, w, epi=ops.make_epilogue(bias=bias, ld_aux=w.shape[0]))
        return ops.gemm_f32(x, w, trans_b=True, epi=ops.make_epilogue(bias=bias))

    def backward(self, d_emb, acc):
        m, S = self.m, self.saved
        batch, c, hw = S["batch"], self.cfeat, self.hw
        dt = m.dtype
        d_emb = d_emb.to(torch.float32).contiguous()

        def put(name, value):
            g = m.grad(name)
            if acc:
                ops.axpy_(g, 1.0, value.to(torch.float32).contiguous())
            else:
                g.copy_(value.reshape(g.shape))
        # c_proj
        put("visual.attnpool.c_proj.weight", ops.gemm_f32(d_emb, S["pooled"], trans_a=True))
        put("visual.attnpool.c_proj.bias", ops.colsum(d_emb))
        d_pooled = ops.gemm_f32(d_emb, m.param("visual.attnpool.c_proj.weight"))
        d_att = torch.zeros(batch, hw + 1, c, dtype=dt, device=d_emb.device)
        d_att[:, 0].copy_(d_pooled)
        d_qkv = ops.attention_bwd(S["qkv"], d_att.view(batch * (hw + 1), c), batch, hw + 1, self.heads, False)
        # only the mean token's query enters the output (AttentionPool2d queries with x[:1]): the other rows' query gradients are not the model's
        dq = d_qkv.view(batch, hw + 1, 3 * c)
        dq[:, 1:, :c].zero_()
        dwqkv = self._tn(d_qkv, S["tokens"])                      # [3c, c] fp32
        dbqkv = ops.colsum(d_qkv)
        for k, n in enumerate(("q_proj", "k_proj", "v_proj")):
            put(f"visual.attnpool.{n}.weight", dwqkv[k * c:(k + 1) * c])
            put(f"visual.attnpool.{n}.bias", dbqkv[k * c:(k + 1) * c])
        d_tokens = self._nt(d_qkv, self.gw["attnpool.qkv"][1])    # [B*(hw+1), c]
        put("visual.attnpool.positional_embedding", ops.colsum(d_tokens.view(batch, (hw + 1) * c)))      # sum over the batch
        dx = ops.attnpool_tokens_bwd(d_tokens, batch, hw)
        if m.comm is not None:
            m.comm.bucket_ready("visual.head")
        # bottlenecks, last to first
        for (p, c1, c2, c3, down, stride), rec in zip(reversed(self.blocks), reversed(S["blocks"])):
            h, w = rec["h"], rec["w"]
            ho, wo = h // stride, w // stride
            dz3, didn = self._bn_bwd(c3, dx, rec["y3"], rec["z3"], rec["m3"], rec["r3"], True, acc, want_dres=True)
            da2 = self._conv_bwd(c3, dz3, rec["a2"], batch, ho, wo, acc)
            dy2 = ops.avgpool_bwd(da2, batch, h, w, c2.cout, stride) if stride > 1 else da2
            if self.implicit(c2):
                _, imgd = self._halo(f"grad{c2.cout}", batch, h, w, _pad64(c2.cout), dx.device)
                self._bn_bwd(c2, dy2, rec["y2"], rec["z2"], rec["m2"], rec["r2"], True, acc, halo=(imgd, h, w))      # dz2 into the bordered image
                dy1 = ops.conv3x3_bf16(imgd, self.gw[c2.name][1], batch, h, w)
                self._conv3x3_dw(c2, imgd, rec["y1_halo"], batch, h, w, acc)
            else:
                dz2, _ = self._bn_bwd(c2, dy2, rec["y2"], rec["z2"], rec["m2"], rec["r2"], True, acc)
                dy1 = self._conv_bwd(c2, dz2, rec["y1"], batch, h, w, acc)
            dz1, _ = self._bn_bwd(c1, dy1, rec["y1"], rec["z1"], rec["m1"], rec["r1"], True, acc)
            if down is not None:
                dzd, _ = self._bn_bwd(down, didn, None, rec["zd"], rec["md"], rec["rd"], False, acc)
                dxi = self._conv_bwd(down, dzd, rec["xi"], batch, ho, wo, acc)
                dx_id = ops.avgpool_bwd(dxi, batch, h, w, c1.cin, stride) if stride > 1 else dxi
            else:
                dx_id = didn
            dx = self._conv_bwd(c1, dz1, rec["x"], batch, h, w, acc, resid=dx_id)      # main path + shortcut, joined in the GEMM epilogue
            if m.comm is not None:
                m.comm.bucket_ready(p)
        # stem
        h, w = S["stem_hw"]
        dx = ops.avgpool_bwd(dx, batch, h, w, self.width, 2)
        for i in reversed(range(len(self.stem))):
            cv = self.stem[i]
            x_in, x_halo, z, y, mean, rstd, hi, wi = S["stem"][i]
            if self.implicit(cv):
                _, imgd = self._halo(f"grad{cv.cout}", batch, hi, wi, _pad64(cv.cout), dx.device)
                self._bn_bwd(cv, dx, y, z, mean, rstd, True, acc, halo=(imgd, hi, wi))
                dx = ops.conv3x3_bf16(imgd, self.gw[cv.name][1], batch, hi, wi)
                self._conv3x3_dw(cv, imgd, x_halo[0], batch, hi, wi, acc)
                continue
            dz, _ = self._bn_bwd(cv, dx, y, z, mean, rstd, True, acc)
            dx = self._conv_bwd(cv, dz, x_in, batch, hi, wi, acc, need_dx=i > 0, images=S["images"] if i == 0 else None)
        if m.comm is not None:
            m.comm.bucket_ready("visual.stem")
