"""CPU oracle for the CAVP forward hot path — TEST INFRASTRUCTURE, NOT PRODUCT.

A plain-PyTorch, fp32, CPU restatement of `models/cavp_model.py::CAVP.forward` of the reference
(cyh-0/CAVP @ 2025-03-07).  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this file; the product path (`cavp_amd/`) never does and fails loudly without its HIP library.

It is written functionally over a `state_dict` with the reference's key tree so it cannot share
code with the product's module tree.  Every function cites the reference lines it restates.

Pinning: the reference has no tests / golden vectors.  This oracle is pinned by
`tests/golden/*.npz`, produced by `tools/make_golden.py`, which imports the *reference itself* in the authoring
container, feeds it `cavp_amd.synth` weights/inputs and stores its outputs; `tests/test_oracle_golden.py`
checks this file against them (max abs err <= 1e-5).  timm==0.4.9's `Mlp` (not vendored in the reference) is
restated from its published definition: fc1 -> GELU(erf) -> fc2, dropout p=0.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # models/visual/deeplabv3/encoder_decoder.py:10, resnet.py bn_eps default
LN_EPS = 1e-5  # nn.LayerNorm default (attn.py:130,229)
NUM_HEADS = 4  # attn.py:178


# --------------------------------------------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------------------------------------------
def _bn(x, sd, p, train):
    """nn.BatchNorm2d(eps=1e-5): eval -> running stats; train -> batch stats (biased var)."""
    if train:
        return F.batch_norm(x, None, None, sd[p + ".weight"], sd[p + ".bias"], True, 0.1, BN_EPS)
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        False, 0.1, BN_EPS)


def _conv(x, sd, p, stride=1, padding=0, dilation=1):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride, padding, dilation)


def _linear(x, sd, p):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _ln(x, sd, p):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], LN_EPS)


def _mlp(x, sd, p):
    """timm 0.4.9 Mlp: fc2(GELU_erf(fc1(x))), drop=0 (cavp_model.py:123-128, attn.py:138-143)."""
    return _linear(F.gelu(_linear(x, sd, p + ".fc1")), sd, p + ".fc2")


# --------------------------------------------------------------------------------------------------------------
# ResNet-50, deep stem, dilated (resnet.py:101-201; encoder_decoder.py:14-59)
# --------------------------------------------------------------------------------------------------------------
def resnet50_block_table(last_three_dilation_stride: Sequence[bool]) -> List[List[Tuple[int, int, int, bool]]]:
    """Per layer, per block: (planes, stride, dilation_of_3x3, has_downsample).

    resnet.py:159-184 `_make_layer` (+ resnet.py:142-157 call sites) then encoder_decoder.py:36-55: every block i
    of layer4 gets stride 2 -> 1 and its 3x3 dilation/padding set to 2, 4, 8."""
    layers = [3, 4, 6, 3]
    planes = [64, 128, 256, 512]
    strides = [1, 2, 2, 2]
    dil_flags = [False] + list(last_three_dilation_stride)
    inplanes, dilation = 128, 1
    table = []
    for li in range(4):
        prev_dil, stride = dilation, strides[li]
        if dil_flags[li]:
            dilation *= stride
            stride = 1
        blocks = []
        for bi in range(layers[li]):
            s = stride if bi == 0 else 1
            d = prev_dil if bi == 0 else dilation
            ds = bi == 0 and (stride != 1 or inplanes != planes[li] * 4)
            blocks.append((planes[li], s, d, ds))
        inplanes = planes[li] * 4
        table.append(blocks)
    # Backbone._nostride_dilate on layer4 (encoder_decoder.py:36-55)
    dil = 2
    new4 = []
    for (pl, s, d, ds) in table[3]:
        new4.append((pl, 1, dil, ds))
        dil *= 2
    table[3] = new4
    return table


def _bottleneck(x, sd, p, stride, dilation, has_ds, train):
    """resnet.py:75-98."""
    out = F.relu(_bn(_conv(x, sd, p + ".conv1"), sd, p + ".bn1", train))
    out = F.relu(_bn(_conv(out, sd, p + ".conv2", stride, dilation, dilation), sd, p + ".bn2", train))
    out = _bn(_conv(out, sd, p + ".conv3"), sd, p + ".bn3", train)
    res = x
    if has_ds:
        res = _bn(_conv(x, sd, p + ".downsample.0", stride), sd, p + ".downsample.1", train)
    return F.relu(out + res)


def backbone_forward(image, sd, last_three_dilation_stride, train=False, p="backbone.backbone"):
    """resnet.py:186-201 (deep stem resnet.py:107-121)."""
    x = F.relu(_bn(_conv(image, sd, p + ".conv1.0", 2, 1), sd, p + ".conv1.1", train))
    x = F.relu(_bn(_conv(x, sd, p + ".conv1.3", 1, 1), sd, p + ".conv1.4", train))
    x = _conv(x, sd, p + ".conv1.6", 1, 1)
    x = F.relu(_bn(x, sd, p + ".bn1", train))
    x = F.max_pool2d(x, 3, 2, 1)
    feats = []
    for li, blocks in enumerate(resnet50_block_table(last_three_dilation_stride)):
        for bi, (_, s, d, ds) in enumerate(blocks):
            x = _bottleneck(x, sd, f"{p}.layer{li + 1}.{bi}", s, d, ds, train)
        feats.append(x)
    return feats


# --------------------------------------------------------------------------------------------------------------
# PVTv2-B5 (models/visual/backbones/pvt/pvt.py:413-421: dims 64/128/320/512, heads 1/2/5/8, depths 3/6/40/3, sr 8/4/2/1)
# --------------------------------------------------------------------------------------------------------------
PVT_DIMS, PVT_HEADS, PVT_DEPTHS, PVT_SR = (64, 128, 320, 512), (1, 2, 5, 8), (3, 6, 40, 3), (8, 4, 2, 1)


def _lne(x, sd, p, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def pvt_forward(image, sd, p="backbone", drop_scales=None):
    """PyramidVisionTransformerV2.forward_features (pvt.py:291-306); Block :166-170; Attention :102-130 (softmax);
    Mlp :46-55 with DWConv :320-326; OverlapPatchEmbed :209-215.  Block / stage norms use eps 1e-6, the patch-embed
    norm and the attention's sr norm the LayerNorm default 1e-5.
    drop_scales (training, pvt.py:144,167-168): timm DropPath as one per-sample factor mask / keep_prob [B] (or None) per
    residual branch in forward order - x + drop_path(f(x)) == x + factor[:, None, None] * f(x)."""
    def dp(t):
        s_ = None if drop_scales is None else drop_scales[dp.i]
        dp.i += 1
        return t if s_ is None else t * s_.to(t.dtype).view(-1, 1, 1)
    dp.i = 0
    x = image
    feats = []
    B = image.shape[0]
    for i in range(4):
        k, s_, pad = (7, 4, 3) if i == 0 else (3, 2, 1)
        x = F.conv2d(x, sd[f"{p}.patch_embed{i + 1}.proj.weight"], sd[f"{p}.patch_embed{i + 1}.proj.bias"], s_, pad)
        H, W = x.shape[-2:]
        x = _lne(x.flatten(2).transpose(1, 2), sd, f"{p}.patch_embed{i + 1}.norm", 1e-5)
        C, nh, sr = PVT_DIMS[i], PVT_HEADS[i], PVT_SR[i]
        for j in range(PVT_DEPTHS[i]):
            b = f"{p}.block{i + 1}.{j}"
            n1 = _lne(x, sd, b + ".norm1", 1e-6)
            N = n1.shape[1]
            q = _linear(n1, sd, b + ".attn.q").reshape(B, N, nh, C // nh).permute(0, 2, 1, 3)
            if sr > 1:
                x_ = n1.permute(0, 2, 1).reshape(B, C, H, W)
                x_ = F.conv2d(x_, sd[b + ".attn.sr.weight"], sd[b + ".attn.sr.bias"], sr).reshape(B, C, -1).permute(0, 2, 1)
                x_ = _lne(x_, sd, b + ".attn.norm", 1e-5)
            else:
                x_ = n1
            kv = _linear(x_, sd, b + ".attn.kv").reshape(B, -1, 2, nh, C // nh).permute(2, 0, 3, 1, 4)
            attn = ((q @ kv[0].transpose(-2, -1)) * (C // nh) ** -0.5).softmax(dim=-1)
            o = (attn @ kv[1]).transpose(1, 2).reshape(B, N, C)
            x = x + dp(_linear(o, sd, b + ".attn.proj"))
            n2 = _lne(x, sd, b + ".norm2", 1e-6)
            h = _linear(n2, sd, b + ".mlp.fc1")
            hc = h.shape[-1]
            h = F.conv2d(h.transpose(1, 2).reshape(B, hc, H, W), sd[b + ".mlp.dwconv.dwconv.weight"],
                         sd[b + ".mlp.dwconv.dwconv.bias"], 1, 1, 1, hc).flatten(2).transpose(1, 2)
            x = x + dp(_linear(F.gelu(h), sd, b + ".mlp.fc2"))
        x = _lne(x, sd, f"{p}.norm{i + 1}", 1e-6)
        x = x.reshape(B, H, W, -1).permute(0, 3, 1, 2).contiguous()
        feats.append(x)
    return feats


# --------------------------------------------------------------------------------------------------------------
# DeepLabV3+ encoder side (encoder_decoder.py:97-105, ASPP :137-156)
# --------------------------------------------------------------------------------------------------------------
def aspp_forward(x, sd, train=False, p="segment.aspp", rates=(6, 12, 18)):
    outs = [_conv(x, sd, p + ".map_convs.0")]
    for i, r in enumerate(rates):
        outs.append(_conv(x, sd, f"{p}.map_convs.{i + 1}", 1, r, r))
    out = torch.cat(outs, 1)
    out = F.leaky_relu(_bn(out, sd, p + ".map_bn", train), 0.01)
    out = _conv(out, sd, p + ".red_conv")
    pool = x.view(x.size(0), x.size(1), -1).mean(-1).view(x.size(0), x.size(1), 1, 1)  # :159-161
    pool = _conv(pool, sd, p + ".global_pooling_conv")
    pool = F.leaky_relu(_bn(pool, sd, p + ".global_pooling_bn", train), 0.01)
    pool = _conv(pool, sd, p + ".pool_red_conv")
    out = out + pool  # repeat-broadcast (:150-153)
    return F.leaky_relu(_bn(out, sd, p + ".red_bn", train), 0.01)


def forward_feature(feats, sd, train=False, taps=None):
    f = aspp_forward(feats[-1], sd, train)
    if taps is not None:
        taps["aspp"] = f
    low = feats[0]
    low = F.relu(_bn(_conv(low, sd, "segment.reduce.0"), sd, "segment.reduce.1", train))
    f = F.interpolate(f, size=low.shape[-2:], mode="bilinear", align_corners=True)
    return torch.cat((f, low), 1)


# --------------------------------------------------------------------------------------------------------------
# VGGish audio encoder (vgg.py:5-36, audio_network.py:33-34)
# --------------------------------------------------------------------------------------------------------------
def audio_forward(audio, sd, p="audio_backbone.backbone"):
    x = audio
    idx = 0
    for v in [64, "M", 128, "M", 256, 256, "M", 512, 512, "M"]:
        if v == "M":
            x = F.max_pool2d(x, 2, 2)
            idx += 1
        else:
            x = F.relu(_conv(x, sd, f"{p}.features.{idx}", 1, 1))
            idx += 2
    x = x.permute(0, 2, 3, 1).contiguous().view(x.size(0), -1)  # NHWC flatten, vgg.py:19-22
    x = F.relu(_linear(x, sd, p + ".embeddings.0"))
    x = F.relu(_linear(x, sd, p + ".embeddings.2"))
    return F.relu(_linear(x, sd, p + ".embeddings.4"))


# --------------------------------------------------------------------------------------------------------------
# fusion: projector + cross-modal attention (cavp_model.py:143-154; attn.py:232-244,152-162,146-150,73-106)
# --------------------------------------------------------------------------------------------------------------
def _heads(x, B, N, C):
    return x.reshape(B, N, NUM_HEADS, C // NUM_HEADS).permute(0, 2, 1, 3)  # attn.py:64-71


def _attention(xq, xk, xv, sd, p):
    """attn.py:73-106 — sigmoid-gated (not softmax) attention, qkv_bias=False, no mask, dropout 0."""
    B, N, C = xq.shape
    q = _heads(_linear(xq, sd, p + ".q"), B, N, C)
    k = _heads(_linear(xk, sd, p + ".k"), B, xk.shape[1], C)
    v = _heads(_linear(xv, sd, p + ".v"), B, xv.shape[1], C)
    attn = torch.sigmoid((q @ k.transpose(-2, -1)) * (C // NUM_HEADS) ** -0.5)
    x = (attn @ v).transpose(1, 2).reshape(B, N, C)
    return _linear(x, sd, p + ".proj"), attn


def _sdp(q, k, v, sd, p):
    """Block.SDPAttention attn.py:146-150: residual is the (already normed) q."""
    out, attn = _attention(q, k, v, sd, p + ".attn")
    q = q + out
    q = q + _mlp(_ln(q, sd, p + ".norm2"), sd, p + ".mlp")
    return q, attn


def cross_attention(fv_nchw, fa_nc11, sd, p="cross_att", compute_dead_branch=False):
    """CROSS_ATTENTION.forward attn.py:232-244 with depth=1 Block.forward_ca attn.py:152-162."""
    B, C, H, W = fv_nchw.shape
    f_v = _linear(fv_nchw.flatten(2).transpose(1, 2), sd, p + ".patch_embed_v.proj")
    f_a = _linear(fa_nc11.flatten(2).transpose(1, 2), sd, p + ".patch_embed_a.proj")
    blk = p + ".blocks.0"
    f_v = _ln(f_v, sd, blk + ".norm1")
    f_a = _ln(f_a, sd, blk + ".norm1")
    f_v, attn_v = _sdp(f_v, f_a, f_a, sd, blk)
    if compute_dead_branch:  # attn.py:161; result dropped at cavp_model.py:151
        f_a, _ = _sdp(f_a, f_v, f_v, sd, blk)
    f_v = _ln(f_v, sd, p + ".norm")
    return f_v, f_a, attn_v


def forward_fusion(fea_v, fea_a, sd):
    """cavp_model.py:143-154."""
    b, c, h, w = fea_v.shape
    tok = fea_v.flatten(2).transpose(1, 2)
    proj = _mlp(tok, sd, "visual_projector")
    fea_v_proj = proj.transpose(1, 2).reshape(b, c, h, w)
    fa = fea_a[:, :, None, None]
    f_v, _, attn_v = cross_attention(fea_v_proj, fa, sd)
    out = f_v.transpose(1, 2).reshape(b, c, h, w)
    return out, {"audio": fa, "visual": fea_v_proj.clone(), "attn_v": attn_v}


# --------------------------------------------------------------------------------------------------------------
# decoder head (encoder_decoder.py:62-75, cavp_model.py:138-141)
# --------------------------------------------------------------------------------------------------------------
def forward_cls(x, sd, input_shape, train=False, taps=None):
    p = "segment.upsample"
    f = F.relu(_bn(_conv(x, sd, p + ".last_conv.0", 1, 1), sd, p + ".last_conv.1", train))
    f = F.relu(_bn(_conv(f, sd, p + ".last_conv.3", 1, 1), sd, p + ".last_conv.4", train))
    if taps is not None:
        taps["last_conv"] = f
    lo = _conv(f, sd, p + ".classifier")
    if taps is not None:
        taps["logits_lowres"] = lo
    return F.interpolate(lo, size=input_shape, mode="bilinear", align_corners=False)


# --------------------------------------------------------------------------------------------------------------
# CAVP.forward (cavp_model.py:199-205; forward_inference :190-197; forward_train :175-188)
# --------------------------------------------------------------------------------------------------------------
def cavp_forward(sd: Dict[str, torch.Tensor], image, audio, last_three_dilation_stride=(False, False, False),
                 eval_mode: bool = True, bn_train: Optional[bool] = None, taps: Optional[dict] = None,
                 seg_model: str = "DeepLabV3Plus", drop_scales=None):
    """Returns (out_pred, out_fusion, {"audio","visual","attn_v"}).

    eval_mode=True  -> forward_inference: image [B], audio [B].
    eval_mode=False -> forward_train: visual features duplicated to 2B, audio [2B] (cavp_model.py:181).
    bn_train: BatchNorm uses batch statistics (module.train()); defaults to `not eval_mode`."""
    if bn_train is None:
        bn_train = not eval_mode
    input_shape = tuple(image.shape[-2:])
    if seg_model == "PVT":   # cavp_model.py:106-115
        feats = pvt_forward(image, sd, drop_scales=drop_scales)
    else:
        feats = backbone_forward(image, sd, last_three_dilation_stride, bn_train)
    if taps is not None:
        for i, f in enumerate(feats):
            taps[("stage" if seg_model == "PVT" else "layer") + str(i + 1)] = f
    fea_v = forward_feature(feats, sd, bn_train, taps)
    if not eval_mode:
        fea_v = torch.cat((fea_v, fea_v.clone()), 0)
    fea_a = audio_forward(audio, sd)
    if taps is not None:
        taps["fea_v"] = fea_v
        taps["fea_a"] = fea_a
    out_fusion, pack = forward_fusion(fea_v, fea_a, sd)
    out_pred = forward_cls(out_fusion, sd, input_shape, bn_train, taps)
    return out_pred, out_fusion, pack


def ce_loss_train(out_pred, label, B):
    """trainer_cavp_vpo_mono.py:171,187 + loss/losser.py:60-62: CE(ignore_index=255) on out[:B] + out[B:]*0."""
    output = out_pred[:B] + out_pred[B:] * 0.0
    return F.cross_entropy(output, label, ignore_index=255)
