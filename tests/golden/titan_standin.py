"""Stand-in for the TITAN slide encoder (HF MahmoodLab/TITAN @ b2fb4f47..., utils/constants.py:22-23) -- TEST INFRASTRUCTURE.

The TITAN snapshot's source and weights are not in the reference tree ("TITAN backbone: parity unpinned"),
but the reference's adapter (models/aggregators/titan_adapter.py) only touches a narrow surface of its
`VisionTransformer` base class:
    ctor kwargs (TA:88-104), `pos_encode_type`, `masked_im_modeling`, `get_alibi(w, h, bg_mask)`, `patch_embed`,
    `_pos_embed(x, coords, w, h)`, `norm_pre` (TA:253-293), `blocks.modules_list[i](x, attn_bias, bg_mask)` (TA:359-361,394),
    `norm`, `forward_attn_pool(x, bg_mask=)` (TA:401-402).
This module is OUR OWN small ViT with exactly that surface (pre-LN blocks, dense attention with an additive 2-D ALiBi
bias, one-query attentional pooling).  It is what the golden generator plugs into the fake snapshot package so that the
REFERENCE's TITAN adapter code runs end to end, and what the GPU test plugs into modaltune_amd.titan as the
"bring your own backbone" -- so the adapter-side flow (feature gridding, background masking, interaction blocks, cat
head on the pooled image token) is pinned against the reference; the real backbone's arithmetic stays unpinned.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class VisionConfig:
    grid_size = 14
    global_pool = False
    embed_dim = 768
    depth = 6
    num_heads = 12
    mlp_ratio = 1.0              # (stand-in: narrow MLP keeps the fixture generation fast)
    qkv_bias = True
    mlp_patch_embed_dim = 768
    pos_encode_type = "alibi"
    attentional_pool = True
    attn_pooler_queries = 1
    attn_pooler_heads = 12


class TitanConfig:
    def __init__(self):
        self.vision_config = VisionConfig()


class _Block(nn.Module):
    def __init__(self, dim, heads, mlp_ratio, qkv_bias):
        super().__init__()
        self.heads = heads
        self.norm1 = nn.LayerNorm(dim)
        self.qkv = nn.Linear(dim, 3 * dim, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.norm2 = nn.LayerNorm(dim)
        self.fc1 = nn.Linear(dim, int(dim * mlp_ratio))
        self.fc2 = nn.Linear(int(dim * mlp_ratio), dim)

    def forward(self, x, attn_bias=None, bg_mask=None):
        B, N, D = x.shape
        # the reference passes the cls-prefixed background mask of the WHOLE grid to every block (adapter_modules.py:535):
        # [1, 1 + w * h] bool, cls kept, exactly N entries set (the tokens were already filtered with it, TA:282-291)
        assert bg_mask is not None and bg_mask.dtype == torch.bool and bg_mask.dim() == 2 and bool(bg_mask[0, 0])
        assert int(bg_mask.sum()) == N, (int(bg_mask.sum()), N)
        q, k, v = self.qkv(self.norm1(x)).view(B, N, 3, self.heads, D // self.heads).permute(2, 0, 3, 1, 4)
        s = q @ k.transpose(-1, -2) * (D // self.heads) ** -0.5
        if attn_bias is not None:
            s = s + attn_bias
        a = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, N, D)
        x = x + self.proj(a)
        return x + self.fc2(F.gelu(self.fc1(self.norm2(x))))


class _Blocks(nn.Module):
    def __init__(self, blocks):
        super().__init__()
        self.modules_list = nn.ModuleList(blocks)


class VisionTransformer(nn.Module):
    def __init__(self, grid_size=14, global_pool=False, embed_dim=768, depth=6, num_heads=12, mlp_ratio=1.0, qkv_bias=True,
                 mlp_patch_embed_dim=768, pos_encode_type="alibi", attentional_pool=True, attn_pooler_queries=1,
                 attn_pooler_heads=12, **kwargs):
        super().__init__()
        self.pos_encode_type, self.masked_im_modeling = pos_encode_type, False
        self.local_alibi_status = self.global_alibi_status = False
        self.num_heads = num_heads
        self.patch_embed = nn.Sequential(nn.Linear(mlp_patch_embed_dim, embed_dim), nn.GELU(), nn.Linear(embed_dim, embed_dim))
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.norm_pre = nn.LayerNorm(embed_dim)
        self.blocks = _Blocks([_Block(embed_dim, num_heads, mlp_ratio, qkv_bias) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim)
        self.pool_query = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pool_attn = nn.MultiheadAttention(embed_dim, attn_pooler_heads, batch_first=True)
        self.pool_norm = nn.LayerNorm(embed_dim)

    def get_alibi(self, w, h, bg_mask=None):
        """[1, heads, T, T] additive bias: -slope_h * euclidean grid distance between patch tokens; 0 to / from cls.
        With bg_mask ([1, w, h] bool) only the foreground cells (+ cls) are kept, in row-major order."""
        dev = self.cls_token.device
        ii, jj = torch.meshgrid(torch.arange(w, device=dev), torch.arange(h, device=dev), indexing="ij")
        pos = torch.stack([ii.reshape(-1), jj.reshape(-1)], 1).to(self.cls_token.dtype)
        if bg_mask is not None:
            pos = pos[bg_mask.reshape(-1)]
        dist = torch.cdist(pos, pos)
        T = pos.shape[0] + 1
        bias = torch.zeros(self.num_heads, T, T, dtype=pos.dtype, device=dev)
        slopes = torch.tensor([2.0 ** (-8.0 * (i + 1) / self.num_heads) for i in range(self.num_heads)], dtype=pos.dtype, device=dev)
        bias[:, 1:, 1:] = -slopes.view(-1, 1, 1) * dist
        return bias.unsqueeze(0)

    def _pos_embed(self, x, coords, w, h):
        return torch.cat((self.cls_token.expand(x.shape[0], -1, -1), x), dim=1)      # ALiBi: no additive position table

    def forward_attn_pool(self, x, bg_mask=None):
        assert bg_mask is not None and bg_mask.dtype == torch.bool and int(bg_mask.sum()) == x.shape[1]      # TA:402
        q = self.pool_query.expand(x.shape[0], -1, -1)
        out, _ = self.pool_attn(q, x, x, need_weights=False)
        return self.pool_norm(out[:, 0]), x


def init_standin(model: nn.Module, seed: int):
    """Deterministic weights for every stand-in backbone tensor (keys sorted, one generator stream per key)."""
    import zlib
    import numpy as np
    own = {k for k, _ in VisionTransformer().state_dict().items()}
    sd = model.state_dict()
    with torch.no_grad():
        for k in sorted(own):
            t = sd[k]
            r = np.random.Generator(np.random.PCG64([int(seed), zlib.crc32(("titan_standin/" + k).encode())]))
            if t.dim() >= 2 and "norm" not in k:
                a = r.standard_normal(tuple(t.shape)) * (0.5 / math.sqrt(t.shape[-1]))
            elif k.endswith("weight"):
                a = 1.0 + 0.05 * r.standard_normal(tuple(t.shape))
            else:
                a = 0.02 * r.standard_normal(tuple(t.shape))
            t.copy_(torch.from_numpy(a.astype(np.float32)).to(t.dtype))
