"""Oracle: plain-torch.nn CPU restatement of the open_clip CLIP encoders (test infrastructure).

PARITY UNPINNED: open-clip-torch==2.29.0 (reference environment.yml:191) is not installed and the
reference holds no fixture for its encoders; this file restates the published architecture that the
reference instantiates at sparsify_clip.py:685-689 and calls at :768-769 (ViT-B-32 / ViT-L-14 configs,
pre-LN residual attention blocks, nn.MultiheadAttention with packed in_proj, exact-erf GELU, LayerNorm
eps 1e-5, cls-token pooling after ln_post, argmax(EOT) text pooling, bias-free projections, causal text
mask).  Parameter names follow open_clip's state_dict so that a real checkpoint could settle parity later.
Cross-checks that do hold: 151.28 M / 427.62 M parameters (tests/test_host_logic.py).
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch
from torch import nn

CONFIGS = {
    "ViT-B-32": dict(embed_dim=512, image_size=224, patch=32, v_width=768, v_layers=12, v_heads=12,
                     ctx=77, vocab=49408, t_width=512, t_layers=12, t_heads=8),
    "ViT-L-14": dict(embed_dim=768, image_size=224, patch=14, v_width=1024, v_layers=24, v_heads=16,
                     ctx=77, vocab=49408, t_width=768, t_layers=12, t_heads=12),
    # open_clip "RN50": ModifiedResNet (three-conv stem, anti-aliasing average pools, attention pooling) + the 512-wide text tower.
    # This is the `model:` every shipped reference YAML names.  v_layers = bottlenecks per stage, v_width = stem width.
    "RN50": dict(embed_dim=1024, image_size=224, v_kind="resnet", v_layers=(3, 4, 6, 3), v_width=64,
                 ctx=77, vocab=49408, t_width=512, t_layers=12, t_heads=8),
    # small shapes for tests (same architecture, not an open_clip config)
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


def canonical_name(name: str) -> str:
    return name.replace("/", "-")


class ResidualAttentionBlock(nn.Module):
    def __init__(self, width, heads, mlp_ratio=4):
        super().__init__()
        self.ln_1 = nn.LayerNorm(width)
        self.attn = nn.MultiheadAttention(width, heads, batch_first=True)
        self.ln_2 = nn.LayerNorm(width)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(width, width * mlp_ratio)), ("gelu", nn.GELU()),
                                              ("c_proj", nn.Linear(width * mlp_ratio, width))]))

    def forward(self, x, attn_mask=None):
        h = self.ln_1(x)
        x = x + self.attn(h, h, h, need_weights=False, attn_mask=attn_mask)[0]
        return x + self.mlp(self.ln_2(x))


class Transformer(nn.Module):
    def __init__(self, width, layers, heads):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.ModuleList([ResidualAttentionBlock(width, heads) for _ in range(layers)])

    def forward(self, x, attn_mask=None):
        for blk in self.resblocks:
            x = blk(x, attn_mask)
        return x


class VisionTransformer(nn.Module):
    def __init__(self, image_size, patch, width, layers, heads, embed_dim):
        super().__init__()
        self.grid = image_size // patch
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch, stride=patch, bias=False)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn(self.grid ** 2 + 1, width))
        self.ln_pre = nn.LayerNorm(width)
        self.transformer = Transformer(width, layers, heads)
        self.ln_post = nn.LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, embed_dim))

    def forward(self, images):
        x = self.conv1(images)                                   # [B, W, g, g]
        x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)  # [B, g*g, W]
        cls = self.class_embedding.to(x.dtype).expand(x.shape[0], 1, -1)
        x = torch.cat([cls, x], dim=1) + self.positional_embedding
        x = self.ln_pre(x)
        x = self.transformer(x)
        x = self.ln_post(x)
        return x[:, 0] @ self.proj


class Bottleneck(nn.Module):
    """open_clip ModifiedResNet bottleneck: 1x1 - 3x3 - (average pool for the stride) - 1x1, all convolutions at stride 1;
    the shortcut is average pool + 1x1 convolution when the shape changes."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.avgpool = nn.AvgPool2d(stride) if stride > 1 else nn.Identity()
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = None
        if stride > 1 or inplanes != planes * 4:
            self.downsample = nn.Sequential(OrderedDict([("-1", nn.AvgPool2d(stride)), ("0", nn.Conv2d(inplanes, planes * 4, 1, bias=False)),
                                                         ("1", nn.BatchNorm2d(planes * 4))]))

    def forward(self, x):
        out = torch.relu(self.bn1(self.conv1(x)))
        out = torch.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(self.avgpool(out)))
        identity = x if self.downsample is None else self.downsample(x)
        return torch.relu(out + identity)


class AttentionPool2d(nn.Module):
    """QKV attention over the HW feature positions plus their mean; the output is the attended mean token."""

    def __init__(self, spacial_dim, embed_dim, num_heads, output_dim):
        super().__init__()
        self.positional_embedding = nn.Parameter(torch.randn(spacial_dim ** 2 + 1, embed_dim) / embed_dim ** 0.5)
        self.k_proj = nn.Linear(embed_dim, embed_dim)
        self.q_proj = nn.Linear(embed_dim, embed_dim)
        self.v_proj = nn.Linear(embed_dim, embed_dim)
        self.c_proj = nn.Linear(embed_dim, output_dim)
        self.num_heads = num_heads

    def forward(self, x):
        b, c = x.shape[0], x.shape[1]
        x = x.reshape(b, c, -1).permute(0, 2, 1)                               # [B, HW, C]
        x = torch.cat([x.mean(dim=1, keepdim=True), x], dim=1) + self.positional_embedding
        hd = c // self.num_heads
        q = self.q_proj(x[:, :1]).reshape(b, 1, self.num_heads, hd).transpose(1, 2)
        k = self.k_proj(x).reshape(b, -1, self.num_heads, hd).transpose(1, 2)
        v = self.v_proj(x).reshape(b, -1, self.num_heads, hd).transpose(1, 2)
        att = torch.softmax(q @ k.transpose(-1, -2) / hd ** 0.5, dim=-1)
        return self.c_proj((att @ v).transpose(1, 2).reshape(b, c))


class ModifiedResNet(nn.Module):
    def __init__(self, layers, output_dim, heads, image_size, width):
        super().__init__()
        self.conv1 = nn.Conv2d(3, width // 2, 3, stride=2, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(width // 2)
        self.conv2 = nn.Conv2d(width // 2, width // 2, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(width // 2)
        self.conv3 = nn.Conv2d(width // 2, width, 3, padding=1, bias=False)
        self.bn3 = nn.BatchNorm2d(width)
        self.avgpool = nn.AvgPool2d(2)
        self._inplanes = width
        self.layer1 = self._make_layer(width, layers[0])
        self.layer2 = self._make_layer(width * 2, layers[1], stride=2)
        self.layer3 = self._make_layer(width * 4, layers[2], stride=2)
        self.layer4 = self._make_layer(width * 8, layers[3], stride=2)
        self.attnpool = AttentionPool2d(image_size // 32, width * 32, heads, output_dim)
        self._init()

    def _make_layer(self, planes, blocks, stride=1):
        layers = [Bottleneck(self._inplanes, planes, stride)]
        self._inplanes = planes * 4
        layers += [Bottleneck(self._inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def _init(self):     # open_clip ModifiedResNet.init_parameters
        std = self.attnpool.c_proj.in_features ** -0.5
        for proj in (self.attnpool.q_proj, self.attnpool.k_proj, self.attnpool.v_proj, self.attnpool.c_proj):
            nn.init.normal_(proj.weight, std=std)
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for name, param in layer.named_parameters():
                if name.endswith("bn3.weight"):
                    nn.init.zeros_(param)

    def forward(self, x):
        x = torch.relu(self.bn1(self.conv1(x)))
        x = torch.relu(self.bn2(self.conv2(x)))
        x = torch.relu(self.bn3(self.conv3(x)))
        x = self.avgpool(x)
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.attnpool(x)


class CLIP(nn.Module):
    def __init__(self, name="ViT-B-32"):
        super().__init__()
        c = CONFIGS[canonical_name(name)]
        self.cfg = c
        if c.get("v_kind") == "resnet":
            self.visual = ModifiedResNet(c["v_layers"], c["embed_dim"], c["v_width"] * 32 // 64, c["image_size"], c["v_width"])
        else:
            self.visual = VisionTransformer(c["image_size"], c["patch"], c["v_width"], c["v_layers"], c["v_heads"], c["embed_dim"])
        self.transformer = Transformer(c["t_width"], c["t_layers"], c["t_heads"])
        self.token_embedding = nn.Embedding(c["vocab"], c["t_width"])
        self.positional_embedding = nn.Parameter(torch.empty(c["ctx"], c["t_width"]))
        self.ln_final = nn.LayerNorm(c["t_width"])
        self.text_projection = nn.Parameter(torch.empty(c["t_width"], c["embed_dim"]))
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))  # present in open_clip, unused by this training loop
        mask = torch.full((c["ctx"], c["ctx"]), float("-inf")).triu_(1)
        self.register_buffer("attn_mask", mask, persistent=False)
        self._init_text()

    def _init_text(self):
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        nn.init.normal_(self.positional_embedding, std=0.01)
        w, layers = self.transformer.width, self.transformer.layers
        proj_std = (w ** -0.5) * ((2 * layers) ** -0.5)
        attn_std, fc_std = w ** -0.5, (2 * w) ** -0.5
        for blk in self.transformer.resblocks:
            nn.init.normal_(blk.attn.in_proj_weight, std=attn_std)
            nn.init.normal_(blk.attn.out_proj.weight, std=proj_std)
            nn.init.normal_(blk.mlp.c_fc.weight, std=fc_std)
            nn.init.normal_(blk.mlp.c_proj.weight, std=proj_std)
        nn.init.normal_(self.text_projection, std=w ** -0.5)

    def encode_image(self, images):
        return self.visual(images)

    def encode_text(self, tokens):
        x = self.token_embedding(tokens) + self.positional_embedding
        x = self.transformer(x, attn_mask=self.attn_mask[: tokens.shape[1], : tokens.shape[1]])
        x = self.ln_final(x)
        return x[torch.arange(x.shape[0]), tokens.argmax(dim=-1)] @ self.text_projection


def create_model(name="ViT-B-32", seed=0):
    g = torch.random.get_rng_state()
    torch.manual_seed(seed)
    m = CLIP(name)
    torch.random.set_rng_state(g)
    return m


def synthetic_batch(seed: int, batch: int, cfg: dict):
    """Images N(0,1) fp32 [B,3,R,R]; captions [SOT, r_1..r_L, EOT, 0...], L ~ U{5..min(30, ctx-3)}."""
    rng = np.random.Generator(np.random.Philox(seed))
    r = cfg["image_size"]
    images = rng.standard_normal((batch, 3, r, r), dtype=np.float32)
    sot, eot = cfg["vocab"] - 2, cfg["vocab"] - 1
    tokens = np.zeros((batch, cfg["ctx"]), dtype=np.int64)
    hi = min(30, cfg["ctx"] - 3)
    lens = rng.integers(min(5, hi), hi + 1, size=batch)
    for i, n in enumerate(lens):
        tokens[i, 0] = sot
        tokens[i, 1:1 + n] = rng.integers(1, sot, size=n)
        tokens[i, 1 + n] = eot
    return images, tokens
