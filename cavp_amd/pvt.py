"""PVTv2-B5 visual backbone on MI355X (reference models/visual/backbones/pvt/pvt.py, selected by seg_model="PVT",
cavp_model.py:106-115; config #4).  Parameter containers with the reference's state_dict names + the HIP forward.

Tokens [B, N, C] are NHWC pixels, so the reference's permute/reshape round trips (pvt.py:108,114-115,320-325,301)
disappear.  Per block: LN -> q Linear; spatial-reduction conv (run as a KH=sr, KW=1 conv over the input viewed as
[B, H, W/sr, sr*C] with stride (sr, 1): its k x k patch is sr contiguous runs of sr*C channels) -> LN -> kv Linear;
MFMA softmax attention (cavp_sra_attention); proj Linear + residual; LN -> fc1 -> depth-wise 3x3 + GELU -> fc2 + residual.
"""
from __future__ import annotations

from functools import partial
from typing import Dict, List

import torch
import torch.nn as nn

from . import ops
from ._lib import ACT_GELU, ACT_NONE, CavpError


class _Container(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise CavpError(f"{type(self).__name__} is a parameter container of the HIP path")


class DWConv(_Container):
    def __init__(self, dim):
        super().__init__()
        self.dwconv = nn.Conv2d(dim, dim, 3, 1, 1, bias=True, groups=dim)


class PvtMlp(_Container):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.dwconv = DWConv(hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)
        self.drop = nn.Dropout(0.0)


class PvtAttention(_Container):
    def __init__(self, dim, num_heads, sr_ratio):
        super().__init__()
        self.dim, self.num_heads, self.sr_ratio = dim, num_heads, sr_ratio
        self.scale = (dim // num_heads) ** -0.5
        self.q = nn.Linear(dim, dim, bias=True)
        self.kv = nn.Linear(dim, dim * 2, bias=True)
        self.attn_drop = nn.Dropout(0.0)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(0.0)
        if sr_ratio > 1:
            self.sr = nn.Conv2d(dim, dim, kernel_size=sr_ratio, stride=sr_ratio)
            self.norm = nn.LayerNorm(dim)          # default eps 1e-5 (pvt.py:79), unlike the 1e-6 block norms


class PvtBlock(_Container):
    def __init__(self, dim, num_heads, mlp_ratio, sr_ratio, norm_layer, drop_path: float = 0.0):
        super().__init__()
        self.drop_prob = float(drop_path)         # timm DropPath probability (pvt.py:144); training pass only
        self.norm1 = norm_layer(dim)
        self.attn = PvtAttention(dim, num_heads, sr_ratio)
        self.drop_path = nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = PvtMlp(dim, int(dim * mlp_ratio))


class OverlapPatchEmbed(_Container):
    def __init__(self, patch_size, stride, in_chans, embed_dim):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=stride, padding=patch_size // 2)
        self.norm = nn.LayerNorm(embed_dim)       # default eps 1e-5 (pvt.py:189)


class PyramidVisionTransformerV2(_Container):
    def __init__(self, embed_dims=(64, 128, 320, 512), num_heads=(1, 2, 5, 8), mlp_ratios=(4, 4, 4, 4),
                 depths=(3, 6, 40, 3), sr_ratios=(8, 4, 2, 1), drop_path_rate: float = 0.1):
        super().__init__()
        norm_layer = partial(nn.LayerNorm, eps=1e-6)
        self.depths, self.embed_dims, self.num_stages = tuple(depths), tuple(embed_dims), 4
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]   # stochastic depth decay rule (pvt.py:229)
        cur = 0
        for i in range(4):
            setattr(self, f"patch_embed{i + 1}", OverlapPatchEmbed(7 if i == 0 else 3, 4 if i == 0 else 2,
                                                                  3 if i == 0 else embed_dims[i - 1], embed_dims[i]))
            setattr(self, f"block{i + 1}", nn.ModuleList([PvtBlock(embed_dims[i], num_heads[i], mlp_ratios[i], sr_ratios[i],
                                                                   norm_layer, dpr[cur + j]) for j in range(depths[i])]))
            setattr(self, f"norm{i + 1}", norm_layer(embed_dims[i]))
            cur += depths[i]


def pvt_v2_b5():
    """pvt.py:413-421 (drop_path_rate = 0.1)."""
    return PyramidVisionTransformerV2()


# ---------------------------------------------------------------------------------------------------------------
class _LinP:
    __slots__ = ("w", "b", "cout", "cin")


def _lin(mod, dtype):
    p = _LinP()
    p.w = ops.pack_weight(mod.weight, dtype)
    p.b = mod.bias.detach() if mod.bias is not None else None
    p.cout, p.cin = mod.weight.shape[0], mod.weight.numel() // mod.weight.shape[0]
    return p


def pack_pvt(bb: PyramidVisionTransformerV2, dtype: torch.dtype) -> Dict[str, object]:
    P: Dict[str, object] = {}
    for i in range(4):
        pe = getattr(bb, f"patch_embed{i + 1}")
        P[f"pe{i}"] = pe.proj.weight.detach() if i == 0 else _lin(pe.proj, dtype)   # stage-1 conv runs on raw OIHW f32
        for j, blk in enumerate(getattr(bb, f"block{i + 1}")):
            k = f"b{i}.{j}."
            P[k + "q"], P[k + "kv"], P[k + "proj"] = _lin(blk.attn.q, dtype), _lin(blk.attn.kv, dtype), _lin(blk.attn.proj, dtype)
            if blk.attn.sr_ratio > 1:
                P[k + "sr"] = _lin(blk.attn.sr, dtype)       # OHWI [Cout][sr][sr][C] == [Cout][sr][1][sr*C]
            P[k + "fc1"], P[k + "fc2"] = _lin(blk.mlp.fc1, dtype), _lin(blk.mlp.fc2, dtype)
            P[k + "dw"] = ops.pack_dwconv_weight(blk.mlp.dwconv.dwconv.weight)
    return P


def _ln(x, ln):
    return ops.layernorm(x, ln.weight.detach(), ln.bias.detach(), torch.empty_like(x), ln.eps)


def pvt_forward_hip(bb: PyramidVisionTransformerV2, image: torch.Tensor, P: Dict[str, object], dtype: torch.dtype) -> List[torch.Tensor]:
    """forward_features (pvt.py:291-306): returns the 4 stage maps as NHWC tensors."""
    dev = image.device
    B = image.shape[0]
    feats = []
    x4 = None
    for i in range(4):
        pe = getattr(bb, f"patch_embed{i + 1}")
        cout = pe.proj.out_channels
        if i == 0:
            H, W = (image.shape[2] + 6 - 7) // 4 + 1, (image.shape[3] + 6 - 7) // 4 + 1
            t = torch.empty((B, H, W, cout), dtype=dtype, device=dev)
            ops.conv_smallcin_kxk(image, P["pe0"], pe.proj.bias.detach(), t, 7, 4, 3)
        else:
            p = P[f"pe{i}"]
            H, W = (x4.shape[1] - 1) // 2 + 1, (x4.shape[2] - 1) // 2 + 1
            t = torch.empty((B, H, W, cout), dtype=dtype, device=dev)
            ops.conv2d(x4, p.w, t, kh=3, kw=3, stride=2, pad=1, shift=p.b)
        C = cout
        N = H * W
        x = _ln(t.view(B, N, C), pe.norm)
        for j, blk in enumerate(getattr(bb, f"block{i + 1}")):
            k = f"b{i}.{j}."
            at = blk.attn
            n1 = _ln(x, blk.norm1)
            q = ops.linear(n1, P[k + "q"].w, torch.empty_like(n1), bias=P[k + "q"].b)
            if at.sr_ratio > 1:
                sr = at.sr_ratio
                if H % sr or W % sr:
                    raise CavpError("PVT spatial-reduction conv needs H, W divisible by sr_ratio")
                xin = n1.view(B, H, W // sr, sr * C)
                xs = torch.empty((B, H // sr, W // sr, C), dtype=dtype, device=dev)
                ops.conv2d(xin, P[k + "sr"].w, xs, kh=sr, kw=1, stride=sr, stride_w=1, shift=P[k + "sr"].b)
                xs = _ln(xs.view(B, -1, C), at.norm)
            else:
                xs = n1
            kv = ops.linear(xs, P[k + "kv"].w, torch.empty((B, xs.shape[1], 2 * C), dtype=dtype, device=dev), bias=P[k + "kv"].b)
            o = ops.sra_attention(q, kv, torch.empty_like(q), at.num_heads, at.scale)
            x = ops.linear(o, P[k + "proj"].w, torch.empty_like(x), bias=P[k + "proj"].b, residual=x)
            n2 = _ln(x, blk.norm2)
            hdim = P[k + "fc1"].cout
            h1 = ops.linear(n2, P[k + "fc1"].w, torch.empty((B, N, hdim), dtype=dtype, device=dev), bias=P[k + "fc1"].b)
            h2 = ops.dwconv3x3(h1.view(B, H, W, hdim), P[k + "dw"], blk.mlp.dwconv.dwconv.bias.detach(),
                               torch.empty((B, H, W, hdim), dtype=dtype, device=dev), act=ACT_GELU)
            x = ops.linear(h2.view(B, N, hdim), P[k + "fc2"].w, torch.empty_like(x), bias=P[k + "fc2"].b, residual=x)
        x = _ln(x, getattr(bb, f"norm{i + 1}"))
        x4 = x.view(B, H, W, C)
        feats.append(x4)
    return feats
