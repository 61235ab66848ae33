"""MI355X-native drop-in for the reference `models/cavp_model.py` (class `CAVP`, `SoundBank`).

Boundary: same constructor, same `forward(image, audio, shuffle_info, ow_flag, eval_mode,
audio_func)` signature and return triple, same attribute tree (`.backbone`, `.segment.business_layer`,
`.visual_projector`, `.cross_att`, `.audio_backbone`, `.memory`) and the same 417-entry state_dict key tree, so the
reference's trainers / eval scripts / checkpoints work unchanged.  Parameters live in stock torch.nn leaf modules
(Conv2d / BatchNorm2d / Linear / LayerNorm) that act purely as *containers*: none of their forwards is ever called.
All arithmetic runs in hand-written gfx950 kernels (libcavp_hip.so) driven by `CAVP._forward_hip`; there is no
PyTorch / CPU fallback — without the library or on CPU tensors the forward raises.

Layout: activations are NHWC in HBM (channels-last), weights are packed OHWI once per weight version; BN (eval)
is folded to a per-channel scale/shift applied in the conv epilogue; torch.cat along channels is a strided write.
Returned tensors keep the reference's logical shapes: `out_pred` is a contiguous NCHW f32 tensor, `out_fusion` /
`pack["visual"]` are NCHW-shaped views of NHWC memory (torch channels_last), values identical.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn

from . import ops
from ._lib import ACT_GELU, ACT_LEAKY, ACT_NONE, ACT_RELU, CavpError

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
# weights of at least this many elements take their first gradient of a step by overwrite (beta = 0) and are left out of the
# arena memset (0: every weight gradient accumulates onto a zeroed arena).  Module constants, not environment variables: the
# product path reads no environment; tests / A-B scripts set them on the module.
_GRAD_OVERWRITE_MIN = 1 << 19
_FUSED_HEAD = True   # False: upsample, CE, and their backward as separate ops


class _Container(nn.Module):
    """Parameter container: compute happens in CAVP._forward_hip, never here."""

    def forward(self, *a, **k):  # pragma: no cover
        raise CavpError(f"{type(self).__name__} is a parameter container of the HIP path; call CAVP.forward")


def _bn(c):
    return nn.BatchNorm2d(c, eps=BN_EPS, momentum=BN_MOMENTUM)


def _conv(ci, co, k, stride=1, pad=0, dil=1, bias=False):
    return nn.Conv2d(ci, co, k, stride=stride, padding=pad, dilation=dil, bias=bias)


# ---------------------------------------------------------------------------------------------------------------
# module tree (names = reference state_dict keys; resnet.py:53-201, encoder_decoder.py:14-156, attn.py:17-244,
# audio_network.py:9-34, vgg.py:5-36)
# ---------------------------------------------------------------------------------------------------------------
class Bottleneck(_Container):
    expansion = 4

    def __init__(self, inplanes, planes, stride, dilation, downsample):
        super().__init__()
        self.conv1, self.bn1 = _conv(inplanes, planes, 1), _bn(planes)
        self.conv2, self.bn2 = _conv(planes, planes, 3, stride, dilation, dilation), _bn(planes)
        self.conv3, self.bn3 = _conv(planes, planes * 4, 1), _bn(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.relu_inplace = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride


def resnet50_blocks(last_three_dilation_stride: Sequence[bool]):
    """[(planes, stride, dilation, has_downsample)] per stage: torchvision-style replace_stride_with_dilation
    (resnet.py:159-184) followed by the DeepLab rewrite of layer4 to stride 1 / dilation 2,4,8
    (encoder_decoder.py:36-55)."""
    depth, width = (3, 4, 6, 3), (64, 128, 256, 512)
    flags = (False,) + tuple(bool(f) for f in last_three_dilation_stride)
    inplanes, dil, stages = 128, 1, []
    for si in range(4):
        stride = 1 if si == 0 else 2
        first_dil = dil
        if flags[si]:
            dil, stride = dil * stride, 1
        stage = []
        for bi in range(depth[si]):
            first = bi == 0
            stage.append((width[si], stride if first else 1, first_dil if first else dil,
                          first and (stride != 1 or inplanes != width[si] * 4)))
        inplanes = width[si] * 4
        stages.append(stage)
    stages[3] = [(pl, 1, 2 << i, ds) for i, (pl, _, _, ds) in enumerate(stages[3])]
    return stages


class ResNet(_Container):
    def __init__(self, last_three_dilation_stride):
        super().__init__()
        self.conv1 = nn.Sequential(_conv(3, 64, 3, 2, 1), _bn(64), nn.ReLU(inplace=True),
                                   _conv(64, 64, 3, 1, 1), _bn(64), nn.ReLU(inplace=True),
                                   _conv(64, 128, 3, 1, 1))
        self.bn1 = _bn(128)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.block_table = resnet50_blocks(last_three_dilation_stride)
        inplanes = 128
        for si, stage in enumerate(self.block_table):
            blocks = []
            for (planes, stride, dil, has_ds) in stage:
                ds = nn.Sequential(_conv(inplanes, planes * 4, 1, stride), _bn(planes * 4)) if has_ds else None
                blocks.append(Bottleneck(inplanes, planes, stride, dil, ds))
                inplanes = planes * 4
            setattr(self, f"layer{si + 1}", nn.Sequential(*blocks))


class Backbone(_Container):
    def __init__(self, back_bone, last_three_dilation_stride):
        super().__init__()
        if back_bone != 50:
            raise ValueError(f"HIP path implements the ResNet-50 visual backbone (north-star); got {back_bone}")
        self.backbone = ResNet(last_three_dilation_stride)


class ASPP(_Container):
    def __init__(self, cin, cout, rates=(6, 12, 18), hidden=256):
        super().__init__()
        self.rates = tuple(rates)
        self.map_convs = nn.ModuleList([_conv(cin, hidden, 1)] + [_conv(cin, hidden, 3, 1, r, r) for r in rates])
        self.map_bn = _bn(hidden * 4)
        self.global_pooling_conv = _conv(cin, hidden, 1)
        self.global_pooling_bn = _bn(hidden)
        self.red_conv = _conv(hidden * 4, cout, 1)
        self.pool_red_conv = _conv(hidden, cout, 1)
        self.red_bn = _bn(cout)
        self.leak_relu = nn.LeakyReLU()


class Upsampling(_Container):
    def __init__(self, classifier_in, num_classes, conv_in):
        super().__init__()
        self.classifier = _conv(classifier_in, num_classes, 1, bias=True)
        self.last_conv = nn.Sequential(_conv(conv_in, 256, 3, 1, 1), _bn(256), nn.ReLU(),
                                       _conv(256, 256, 3, 1, 1), _bn(256), nn.ReLU())


class DeepLabV3Plus(_Container):
    def __init__(self, num_classes, aspp_in_plane=2048, aspp_out_plane=256, classifier_in=256):
        super().__init__()
        conv_in = 112 if aspp_out_plane == 64 else 304
        self.aspp = ASPP(aspp_in_plane, aspp_out_plane, (6, 12, 18))
        self.reduce = nn.Sequential(_conv(aspp_out_plane, 48, 1), _bn(48), nn.ReLU())
        self.upsample = Upsampling(classifier_in, num_classes, conv_in)
        self.business_layer = [self.aspp, self.reduce, self.upsample.last_conv, self.upsample.classifier]


class Mlp(_Container):
    """timm 0.4.9 Mlp parameter layout (fc1, act, fc2, drop)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)


class PatchEmbed(_Container):
    def __init__(self, num_patches, dim_in, embed_dim):
        super().__init__()
        self.proj = nn.Linear(dim_in, embed_dim)
        self.num_patches = num_patches
        self.norm = nn.Identity()


class Attention(_Container):
    def __init__(self, dim, num_heads):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.q = nn.Linear(dim, dim, bias=False)
        self.k = nn.Linear(dim, dim, bias=False)
        self.v = nn.Linear(dim, dim, bias=False)
        self.attn_drop = nn.Dropout(0.0)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(0.0)


class Block(_Container):
    def __init__(self, dim, num_heads, mlp_ratio=4.0):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn = Attention(dim, num_heads)
        self.drop_path = nn.Identity()
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))


class CROSS_ATTENTION(_Container):
    def __init__(self, embed_dim, depth=1, num_heads=4, dim_in=None):
        super().__init__()
        if depth != 1:
            raise ValueError("CAVP uses depth=1 (cavp_model.py:119-121)")
        self.patch_embed_v = PatchEmbed(128 * 128, dim_in, embed_dim)
        self.patch_embed_a = PatchEmbed(1, dim_in, embed_dim)
        # present in checkpoints, never added on the path (attn.py:235-238)
        self.pos_embed_v = nn.Parameter(torch.zeros(1, 128 * 128, embed_dim))
        self.pos_embed_a = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_drop = nn.Dropout(0.0)
        self.blocks = nn.Sequential(Block(embed_dim, num_heads))
        self.norm = nn.LayerNorm(embed_dim)


class VGG(_Container):
    CFG = (64, "M", 128, "M", 256, 256, "M", 512, 512, "M")

    def __init__(self, out_plane):
        super().__init__()
        layers, cin = [], 1
        for v in self.CFG:
            if v == "M":
                layers.append(nn.MaxPool2d(2, 2))
            else:
                layers += [_conv(cin, v, 3, 1, 1, bias=True), nn.ReLU(inplace=True)]
                cin = v
        self.features = nn.Sequential(*layers)
        self.embeddings = nn.Sequential(nn.Linear(512 * 4 * 6, 4096), nn.ReLU(True), nn.Linear(4096, 4096),
                                        nn.ReLU(True), nn.Linear(4096, out_plane), nn.ReLU(True))


class AudioModel(_Container):
    def __init__(self, backbone, pretrain_path, out_plane, num_classes=2, in_plane=1):
        super().__init__()
        if backbone != "vgg":
            raise ValueError("HIP path implements the VGGish audio encoder (audio_backbone='vgg'); the torchvision "
                             "ResNet-18 branch is out of scope (SURVEY.md §2.1 row 5)")
        self.backbone = VGG(out_plane)
        if pretrain_path is not None:
            self.load_audio_model(pretrain_path)
        self.cls_head = nn.Linear(out_plane, num_classes)  # unused on the path, present in checkpoints

    def load_audio_model(self, path_):
        """audio_network.py:36-45: load VGGish weights, re-initialise the last FC."""
        param_dict = torch.load(path_, map_location="cpu")
        w = self.backbone.state_dict()["embeddings.4.weight"]
        param_dict["embeddings.4.weight"] = nn.init.kaiming_normal_(torch.zeros_like(w, device="cpu"))
        param_dict["embeddings.4.bias"] = torch.zeros(w.shape[0])
        self.backbone.load_state_dict(param_dict, strict=True)


class SoundBank:
    """Per-class FIFO of audio features (cavp_model.py:21-52).  Host-side bookkeeping used only through
    `forward_audio(audio_func=True)`, which no reference trainer enables; kept for API parity."""

    def __init__(self, out_dim=304, args=None, device=0):
        self.bank_vault = torch.zeros((args.num_classes, args.batch_size, out_dim), requires_grad=False, device=device)

    def update_bank(self, waveform, img_label):
        img_label[:, 0] = 0
        for i, row in enumerate(img_label):
            idx = row.nonzero().flatten().tolist()
            if len(idx) == 1:
                self.queue(idx[0], waveform[i, None] if waveform.dim() == 2 else waveform[i])

    def queue(self, class_idx, fea_a):
        self.bank_vault[class_idx] = torch.cat((self.bank_vault[class_idx][1:], fea_a.detach()), dim=0)

    def overwrite_audio_feature(self, shuffle_fea_a, org_fea_a, mod_idx_map):
        for idx, target_label in mod_idx_map.items():
            shuffle_fea_a[idx] = self.bank_vault[None, target_label][:, 0]
        return shuffle_fea_a


# ---------------------------------------------------------------------------------------------------------------
# packed (kernel-ready) parameters
# ---------------------------------------------------------------------------------------------------------------
class _ConvP:
    __slots__ = ("w", "scale", "shift", "kh", "kw", "stride", "pad", "dil", "cout", "cin")

    def __init__(self, w, scale, shift, kh, kw, stride, pad, dil, cout, cin):
        self.w, self.scale, self.shift = w, scale, shift
        self.kh, self.kw, self.stride, self.pad, self.dil, self.cout, self.cin = kh, kw, stride, pad, dil, cout, cin


class CAVP(nn.Module):
    def __init__(self, backbone, pretrain_path, num_classes=2, ignore_index=255, audio_backbone_pretrain_path=None,
                 visual_backbone=50, args=None, in_plane=1):
        super().__init__()
        seg_model = args.seg_model
        self.seg_model = seg_model
        self.num_classes = num_classes
        if seg_model == "DeepLabV3Plus":
            self.latent_dim = 304
            self.backbone = Backbone(backbone, args.last_three_dilation_stride)
            self.segment = DeepLabV3Plus(num_classes=num_classes, aspp_in_plane=2048, aspp_out_plane=256)
        elif seg_model == "PVT":   # cavp_model.py:106-115
            from .pvt import pvt_v2_b5
            self.latent_dim = 112
            self.backbone = pvt_v2_b5()
            ckpt_path = "../ckpts/pretrained/pvt_v2_b5.pth"   # the reference torch.load()s this path unconditionally
            if os.path.exists(ckpt_path):
                ckpt = torch.load(ckpt_path, map_location="cpu")
                ckpt.pop("head.weight", None)
                ckpt.pop("head.bias", None)
                self.backbone.load_state_dict(ckpt)
            elif not getattr(args, "allow_random_pvt", False):
                # the reference fails here (cavp_model.py:109 torch.load of a missing file); a silently random backbone is
                # worse than either, so say it loudly unless the caller opted in (args.allow_random_pvt = True)
                import warnings
                warnings.warn(f"PVTv2-B5 checkpoint {ckpt_path!r} not found: the visual backbone keeps its RANDOM "
                              f"initialisation (set args.allow_random_pvt = True to silence this)", RuntimeWarning, stacklevel=2)
            self.segment = DeepLabV3Plus(num_classes=num_classes, aspp_in_plane=512, aspp_out_plane=64)
        elif seg_model in ("HRNet", "OCR"):
            raise NotImplementedError(f"seg_model={seg_model!r}: alternate backbones outside the north-star scope "
                                      f"(SURVEY.md §2.1 row 17)")
        else:
            raise ValueError("UNKNOW BACKBONE")  # cavp_model.py:117
        self.cross_att = CROSS_ATTENTION(embed_dim=self.latent_dim, depth=1, dim_in=self.latent_dim)
        self.visual_projector = Mlp(self.latent_dim, 256, self.latent_dim, drop=0.0)
        self.audio_backbone = AudioModel(args.audio_backbone, audio_backbone_pretrain_path, self.latent_dim,
                                         in_plane=in_plane)
        self.memory = SoundBank(out_dim=self.latent_dim, args=args, device=args.local_rank)
        self.local_rank = args.local_rank
        if pretrain_path is not None and seg_model == "DeepLabV3Plus":
            self._load_backbone(pretrain_path)
        # compute configuration of the HIP path
        self.compute_dtype = torch.float32   # torch.bfloat16 = bf16 storage / f32 accumulate
        self._packed: Optional[Dict[str, _ConvP]] = None
        self._packed_sig = None

    # -- host logic -------------------------------------------------------------------------------------------
    def _load_backbone(self, path):
        """utils/pyt_utils.py:42-60 load_model(strict=False): accepts a raw state_dict or {'model': ...}."""
        sd = torch.load(path, map_location="cpu")
        sd = sd.get("model", sd)
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
        self.backbone.backbone.load_state_dict(sd, strict=False)

    def set_compute_dtype(self, dtype: torch.dtype) -> "CAVP":
        if dtype not in (torch.float32, torch.bfloat16):
            raise CavpError("compute dtype must be float32 (parity path) or bfloat16")
        self.compute_dtype = dtype
        self._packed = None
        return self

    def _signature(self):
        ps = list(self.parameters()) + list(self.buffers())
        return (self.compute_dtype, ps[0].device, ps[0].data_ptr(), sum(p._version for p in ps),
                getattr(self, "_param_epoch", 0),
                any(m.training for m in self.modules() if isinstance(m, nn.modules.batchnorm._BatchNorm)))

    def params_changed(self) -> None:
        """Invalidate the eval-mode parameter pack (packed weights + folded BatchNorm).  Tensor `_version`s catch torch-side
        updates; writes made by our own kernels through raw pointers - FusedSGDAdam.step, the running-statistics updates of
        a training step, hipGraph replays of either - bump no version, so every such writer calls this."""
        self._param_epoch = getattr(self, "_param_epoch", 0) + 1
        self._packed = None

    # -- parameter packing --------------------------------------------------------------------------------------
    def _fold(self, bn, lo=None, hi=None):
        c = bn.num_features
        dev = bn.weight.device
        scale = torch.empty(c, dtype=torch.float32, device=dev)
        shift = torch.empty(c, dtype=torch.float32, device=dev)
        ops.bn_fold(bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps, scale, shift)
        return scale, shift

    def _pack_conv(self, conv, bn=None, scale=None, shift=None, raw=False):
        if bn is not None:
            scale, shift = self._fold(bn)
        elif conv.bias is not None:
            shift = conv.bias.detach()
        if isinstance(conv, nn.Linear):
            w = ops.pack_weight(conv.weight, self.compute_dtype)
            return _ConvP(w, scale, shift, 1, 1, 1, 0, 1, conv.out_features, conv.in_features)
        w = conv.weight.detach() if raw else ops.pack_weight(conv.weight, self.compute_dtype)
        return _ConvP(w, scale, shift, conv.kernel_size[0], conv.kernel_size[1], conv.stride[0], conv.padding[0],
                      conv.dilation[0], conv.out_channels, conv.in_channels)

    def _pack(self) -> Dict[str, _ConvP]:
        P: Dict[str, _ConvP] = {}
        if self.seg_model == "PVT":
            from .pvt import pack_pvt
            P["pvt"] = pack_pvt(self.backbone, self.compute_dtype)
            return self._pack_rest(P)
        rn = self.backbone.backbone
        P["stem0"] = self._pack_conv(rn.conv1[0], rn.conv1[1], raw=True)
        P["stem1"] = self._pack_conv(rn.conv1[3], rn.conv1[4])
        P["stem2"] = self._pack_conv(rn.conv1[6], rn.bn1)
        for si in range(4):
            for bi, blk in enumerate(getattr(rn, f"layer{si + 1}")):
                key = f"l{si + 1}.{bi}"
                P[key + ".c1"] = self._pack_conv(blk.conv1, blk.bn1)
                P[key + ".c2"] = self._pack_conv(blk.conv2, blk.bn2)
                P[key + ".c3"] = self._pack_conv(blk.conv3, blk.bn3)
                if blk.downsample is not None:
                    P[key + ".ds"] = self._pack_conv(blk.downsample[0], blk.downsample[1])
        return self._pack_rest(P)

    def _pack_rest(self, P):
        aspp = self.segment.aspp
        ms, mh = self._fold(aspp.map_bn)
        hid = aspp.map_convs[0].out_channels
        for i, cv in enumerate(aspp.map_convs):
            P[f"aspp.map{i}"] = self._pack_conv(cv, scale=ms[i * hid:(i + 1) * hid], shift=mh[i * hid:(i + 1) * hid])
        P["aspp.gp"] = self._pack_conv(aspp.global_pooling_conv, aspp.global_pooling_bn)
        P["aspp.pool_red"] = self._pack_conv(aspp.pool_red_conv)
        P["aspp.red"] = self._pack_conv(aspp.red_conv, aspp.red_bn)
        P["reduce"] = self._pack_conv(self.segment.reduce[0], self.segment.reduce[1])
        up = self.segment.upsample
        P["head0"] = self._pack_conv(up.last_conv[0], up.last_conv[1])
        P["head1"] = self._pack_conv(up.last_conv[3], up.last_conv[4])
        P["cls"] = self._pack_conv(up.classifier)
        vgg = self.audio_backbone.backbone
        convs = [m for m in vgg.features if isinstance(m, nn.Conv2d)]
        P["a.conv0"] = self._pack_conv(convs[0], raw=True)
        for i, cv in enumerate(convs[1:], 1):
            P[f"a.conv{i}"] = self._pack_conv(cv)
        for i, j in enumerate((0, 2, 4)):
            P[f"a.fc{i}"] = self._pack_conv(vgg.embeddings[j])
        P["proj.fc1"] = self._pack_conv(self.visual_projector.fc1)
        P["proj.fc2"] = self._pack_conv(self.visual_projector.fc2)
        ca, blk = self.cross_att, self.cross_att.blocks[0]
        P["ca.pe_v"] = self._pack_conv(ca.patch_embed_v.proj)
        P["ca.pe_a"] = self._pack_conv(ca.patch_embed_a.proj)
        for n in ("q", "k", "v", "proj"):
            P["ca." + n] = self._pack_conv(getattr(blk.attn, n))
        P["ca.fc1"] = self._pack_conv(blk.mlp.fc1)
        P["ca.fc2"] = self._pack_conv(blk.mlp.fc2)
        return P

    def packed(self) -> Dict[str, _ConvP]:
        sig = self._signature()
        if self._packed is None or sig != self._packed_sig:
            if sig[-1]:
                raise CavpError("the eval-mode (folded BatchNorm) parameter pack was requested while a BatchNorm module is in "
                                "training mode; batch-statistics BatchNorm runs through forward_train / train_step")
            self._packed, self._packed_sig = self._pack(), sig
        return self._packed

    # -- kernels ------------------------------------------------------------------------------------------------
    @staticmethod
    def _conv(x, p: _ConvP, out=None, act=ACT_NONE, residual=None, nbias=None):
        n, h, w, _ = x.shape
        ho = (h + 2 * p.pad - p.dil * (p.kh - 1) - 1) // p.stride + 1
        wo = (w + 2 * p.pad - p.dil * (p.kw - 1) - 1) // p.stride + 1
        if out is None:
            out = torch.empty((n, ho, wo, p.cout), dtype=x.dtype, device=x.device)
        return ops.conv2d(x, p.w, out, kh=p.kh, kw=p.kw, stride=p.stride, pad=p.pad, dil=p.dil, scale=p.scale,
                          shift=p.shift, nbias=nbias, residual=residual, act=act)

    @staticmethod
    def _lin(x, p: _ConvP, act=ACT_NONE, residual=None, out=None, res_rows=0):
        if out is None:
            out = torch.empty(x.shape[:-1] + (p.cout,), dtype=x.dtype, device=x.device)
        return ops.linear(x, p.w, out, bias=p.shift, scale=p.scale, residual=residual, act=act, res_rows=res_rows)

    def _backbone_hip(self, image, P):
        dt, dev = self.compute_dtype, image.device
        B, _, H, W = image.shape
        h2, w2 = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        s0 = torch.empty((B, h2, w2, 64), dtype=dt, device=dev)
        p0 = P["stem0"]
        ops.conv3x3_smallcin_nchw(image, p0.w, s0, stride=2, scale=p0.scale, shift=p0.shift, act=ACT_RELU)
        x = self._conv(s0, P["stem1"], act=ACT_RELU)
        x = self._conv(x, P["stem2"], act=ACT_RELU)
        h4, w4 = (h2 - 1) // 2 + 1, (w2 - 1) // 2 + 1
        pooled = torch.empty((B, h4, w4, 128), dtype=dt, device=dev)
        x = ops.maxpool(x, pooled, 3, 2, 1)
        feats = []
        for si, stage in enumerate(self.backbone.backbone.block_table):
            for bi, (_, _, _, has_ds) in enumerate(stage):
                key = f"l{si + 1}.{bi}"
                o = self._conv(x, P[key + ".c1"], act=ACT_RELU)
                o = self._conv(o, P[key + ".c2"], act=ACT_RELU)
                res = self._conv(x, P[key + ".ds"]) if has_ds else x
                x = self._conv(o, P[key + ".c3"], act=ACT_RELU, residual=res)
            feats.append(x)
        return feats

    def _forward_feature_hip(self, feats, P):
        """segment.forward_feature (encoder_decoder.py:97-105) + ASPP (:137-156) -> fea_v NHWC [B, h, w, 304]."""
        f4, f1 = feats[-1], feats[0]
        dt, dev = f4.dtype, f4.device
        B, h, w, _ = f4.shape
        hid = P["aspp.map0"].cout
        cat = torch.empty((B, h, w, 4 * hid), dtype=dt, device=dev)
        for i in range(4):
            self._conv(f4, P[f"aspp.map{i}"], out=cat[..., i * hid:(i + 1) * hid], act=ACT_LEAKY)
        pool = torch.empty((B, f4.shape[-1]), dtype=torch.float32, device=dev)
        ops.global_avgpool(f4, pool)
        if dt != torch.float32:
            pool = ops.cast(pool, torch.empty_like(pool, dtype=dt))
        g = self._lin(pool, P["aspp.gp"], act=ACT_LEAKY)
        g = self._lin(g, P["aspp.pool_red"])
        if dt != torch.float32:
            g = ops.cast(g, torch.empty_like(g, dtype=torch.float32))
        aspp = self._conv(cat, P["aspp.red"], act=ACT_LEAKY, nbias=g)
        _, lh, lw, _ = f1.shape
        co = P["aspp.red"].cout
        fea_v = torch.empty((B, lh, lw, co + P["reduce"].cout), dtype=dt, device=dev)
        ops.bilinear(aspp, fea_v[..., :co], align_corners=True)
        self._conv(f1, P["reduce"], out=fea_v[..., co:], act=ACT_RELU)
        return fea_v, aspp

    def _audio_hip(self, audio, P):
        """AudioModel.forward -> VGG.forward (vgg.py:17-23): fea_a [B, 304] (>= 0)."""
        dt, dev = self.compute_dtype, audio.device
        B, cin, H, W = audio.shape
        if cin != 1:
            raise CavpError("VGGish audio encoder expects [B, 1, 96, 64] log-mel input")
        p0 = P["a.conv0"]
        x = torch.empty((B, H, W, p0.cout), dtype=dt, device=dev)
        ops.conv3x3_smallcin_nchw(audio, p0.w, x, stride=1, scale=None, shift=p0.shift, act=ACT_RELU)
        ci = 1
        for v in VGG.CFG[1:]:
            if v == "M":
                n, h, w, c = x.shape
                x = ops.maxpool(x, torch.empty((n, h // 2, w // 2, c), dtype=dt, device=dev), 2, 2, 0)
            else:
                x = self._conv(x, P[f"a.conv{ci}"], act=ACT_RELU)
                ci += 1
        x = x.reshape(B, -1)  # NHWC flatten == transpose(1,3).transpose(1,2).view(B,-1) of the NCHW tensor
        x = self._lin(x, P["a.fc0"], act=ACT_RELU)
        x = self._lin(x, P["a.fc1"], act=ACT_RELU)
        return self._lin(x, P["a.fc2"], act=ACT_RELU)

    def _fusion_hip(self, fea_v, fea_a, P):
        """forward_fusion (cavp_model.py:143-154) + CROSS_ATTENTION.forward (attn.py:232-244), dead audio-query
        branch (attn.py:161, dropped at cavp_model.py:151) elided."""
        Bv, h, w, Cc = fea_v.shape
        B = fea_a.shape[0]
        if B % Bv:
            raise CavpError(f"audio batch {B} is not a multiple of the visual batch {Bv}")
        # B > Bv (forward_train: audio of 2B, cavp_model.py:181): everything that depends on the images only runs once on Bv
        # rows; the gate reads q[b % Bv] and ca.proj adds residual row p % (Bv * T) instead of 2B copies of both
        T = h * w
        dt, dev = fea_v.dtype, fea_v.device
        blk = self.cross_att.blocks[0]
        tok = fea_v.view(Bv, T, Cc)
        hid = self._lin(tok, P["proj.fc1"], act=ACT_GELU)
        fea_v_proj = self._lin(hid, P["proj.fc2"])
        v0 = self._lin(fea_v_proj, P["ca.pe_v"])
        a0 = self._lin(fea_a, P["ca.pe_a"])
        n1w, n1b = blk.norm1.weight.detach(), blk.norm1.bias.detach()
        vn = ops.layernorm(v0, n1w, n1b, torch.empty_like(v0), blk.norm1.eps)
        an = ops.layernorm(a0, n1w, n1b, torch.empty_like(a0), blk.norm1.eps)
        k = self._lin(an, P["ca.k"])
        vv = self._lin(an, P["ca.v"])
        heads = blk.attn.num_heads
        attn = torch.empty((B, heads, T), dtype=torch.float32, device=dev)
        from . import train as _tr
        if _tr._RANK1_ATTN and ops.attn1_usable(blk.attn):
            # one key per batch item: q GEMM + gate + proj GEMM + residual in one pass over the tokens (csrc/attn_rank1.hip)
            u, pm = ops.attn1_prepare(blk.attn.q.weight.detach(), blk.attn.proj.weight.detach(), k, vv, heads, blk.attn.scale)
            r1 = ops.attn1_fwd(vn, u, pm, blk.attn.proj.bias.detach() if blk.attn.proj.bias is not None else None,
                               torch.empty((B, T, Cc), dtype=dt, device=dev), attn)
            q = o = None
        else:
            q = self._lin(vn, P["ca.q"])
            o = ops.attn_gate(q, k, vv, torch.empty((B, T, Cc), dtype=dt, device=dev), attn, heads, blk.attn.scale)
        if q is None:
            pass
        elif B == Bv:
            r1 = self._lin(o, P["ca.proj"], residual=vn)
        elif (Bv * T) % 256 == 0:
            r1 = self._lin(o, P["ca.proj"], residual=vn, res_rows=Bv * T)
        else:   # the periodic-residual epilogue wants a multiple of 256 rows: tiny inputs take the copy
            r1 = self._lin(o, P["ca.proj"], residual=vn.repeat(B // Bv, 1, 1))
        l2 = ops.layernorm(r1, blk.norm2.weight.detach(), blk.norm2.bias.detach(), torch.empty_like(r1), blk.norm2.eps)
        hh = self._lin(l2, P["ca.fc1"], act=ACT_GELU)
        r2 = self._lin(hh, P["ca.fc2"], residual=r1)
        fn = self.cross_att.norm
        fus = ops.layernorm(r2, fn.weight.detach(), fn.bias.detach(), torch.empty_like(r2), fn.eps)
        if B != Bv:   # pack["visual"]: the reference returns the duplicated projection
            fea_v_proj = fea_v_proj.repeat(B // Bv, 1, 1)
        return fus.view(B, h, w, Cc), fea_v_proj.view(B, h, w, Cc), attn

    def _cls_hip(self, fusion, P, input_shape):
        x = self._conv(fusion, P["head0"], act=ACT_RELU)
        x = self._conv(x, P["head1"], act=ACT_RELU)
        lo = self._conv(x, P["cls"])
        out = torch.empty((fusion.shape[0], P["cls"].cout) + tuple(input_shape), dtype=torch.float32, device=fusion.device)
        return ops.bilinear_to_nchw(lo, out, align_corners=False), lo

    @staticmethod
    def _as_f32(t):
        if t.dtype == torch.float32:
            return t
        return ops.cast(t.contiguous(), torch.empty(t.shape, dtype=torch.float32, device=t.device))

    def _forward_hip(self, image, audio, duplicate_visual: bool, taps: Optional[dict] = None, shuffle=None):
        if not image.is_cuda:
            raise CavpError("CAVP (MI355X path) needs inputs on a HIP device: there is no CPU fallback")
        if image.dtype != torch.float32 or audio.dtype != torch.float32:
            raise CavpError("image / audio must be float32 (they are converted on the fly by the stem kernels)")
        image, audio = image.contiguous(), audio.contiguous()
        P = self.packed()
        input_shape = tuple(image.shape[-2:])
        B = image.shape[0]
        # the audio encoder depends on nothing but its input: second stream, concurrent with the visual backbone (own scratch slot)
        from . import train as _tr
        side = None
        if _tr._SIDE_STREAM and self.seg_model != "PVT":   # (PVTv2: measured slower with the second branch, 9.89 -> 10.17 ms)
            side = getattr(self, "_side_stream", None)
            if side is None or side.device != image.device:
                side = self.__dict__["_side_stream"] = torch.cuda.Stream(device=image.device)
            main = torch.cuda.current_stream()
            ev0 = torch.cuda.Event()
            ev0.record(main)
            side.wait_event(ev0)
            with torch.cuda.stream(side), ops.workspace_slot(1):
                fea_a = self._audio_hip(audio, P)
                ev1 = torch.cuda.Event()
                ev1.record(side)
        if self.seg_model == "PVT":
            from .pvt import pvt_forward_hip
            feats = pvt_forward_hip(self.backbone, image, P["pvt"], self.compute_dtype)
        else:
            feats = self._backbone_hip(image, P)
        fea_v, aspp = self._forward_feature_hip(feats, P)
        if side is None:
            fea_a = self._audio_hip(audio, P)
        else:
            torch.cuda.current_stream().wait_event(ev1)
        if shuffle is not None:   # forward_audio (cavp_model.py:156-173): B clips -> features | shuffled features
            idx = self._bank_and_shuffle(fea_a, shuffle[0], shuffle[1])
            fea_a = torch.cat((fea_a, fea_a.index_select(0, idx)), dim=0)
        # forward_train duplicates the visual features to 2B (`torch.cat((fea_v, fea_v.clone()))`, cavp_model.py:181); here the
        # fusion stage reads the B rows periodically instead (see _fusion_hip)
        if fea_a.shape[0] != (2 if duplicate_visual else 1) * fea_v.shape[0]:
            raise CavpError(f"audio batch {fea_a.shape[0]} vs visual batch {fea_v.shape[0]}: train mode expects audio of 2B "
                            f"(cavp_model.py:181), inference one clip per image")
        fusion, fea_v_proj, attn = self._fusion_hip(fea_v, fea_a, P)
        out_pred, lo = self._cls_hip(fusion, P, input_shape)
        if taps is not None:
            for i, f in enumerate(feats):
                taps[f"layer{i + 1}"] = f.permute(0, 3, 1, 2)
            taps.update(aspp=aspp.permute(0, 3, 1, 2), fea_v=fea_v.permute(0, 3, 1, 2), fea_a=fea_a,
                        logits_lowres=lo.permute(0, 3, 1, 2))
        out_fusion = self._as_f32(fusion).permute(0, 3, 1, 2)
        pack = {"audio": self._as_f32(fea_a)[:, :, None, None],
                "visual": self._as_f32(fea_v_proj).permute(0, 3, 1, 2),
                "attn_v": attn.unsqueeze(-1)}
        return out_pred, out_fusion, pack

    # -- reference API ------------------------------------------------------------------------------------------
    def _stage_on_tape(self, mods, *inputs) -> bool:
        """True when a stage entry point has to run on a training tape (CAVPStageFunction): one of its BatchNorm layers is in
        training mode (batch statistics, running-statistics update) or autograd wants gradients through it.  Otherwise the
        forward-only eval kernels serve it.  Note for validation loops: a freshly built model has requires_grad parameters, so
        `model.eval()` alone still takes the tape (weight re-packs, saved activations, an autograd node - same results, more
        time and memory); wrap inference in `torch.no_grad()`, as the reference's evaluation scripts do."""
        bn_train = any(m.training for top in mods for m in top.modules() if isinstance(m, nn.modules.batchnorm._BatchNorm))
        want = torch.is_grad_enabled() and (any(isinstance(t, torch.Tensor) and t.requires_grad for t in inputs)
                                            or any(p.requires_grad for top in mods for p in top.parameters()))
        return bn_train or want

    def _stage_apply(self, kind, meta, mods, *inputs):
        from .train import CAVPStageFunction
        if not inputs[0].is_cuda:
            raise CavpError("CAVP (MI355X path) needs inputs on a HIP device: there is no CPU fallback")
        skip = {id(p) for p in self.params_without_grad()}
        params = [p for top in mods for p in top.parameters() if p.requires_grad and id(p) not in skip] if torch.is_grad_enabled() else []
        return CAVPStageFunction.apply(self, kind, meta, len(inputs), *inputs, *params)

    @staticmethod
    def _nchw_to_nhwc(t, dtype):
        """A caller's NCHW tensor as a dense NHWC tensor of the compute dtype (a channels_last tensor - what this model
        returns - is re-viewed, anything else goes through one layout pass of the boundary)."""
        v = t.permute(0, 2, 3, 1)
        if not v.is_contiguous():
            v = v.contiguous()
        if v.dtype != dtype:
            v = ops.cast(v, torch.empty(v.shape, dtype=dtype, device=v.device))
        return v

    def forward_cls(self, out, input_shape):
        """cavp_model.py:138-141: decoder head (two 3x3 conv + BN + ReLU, 1x1 classifier) + bilinear x4 (align_corners=False).
        `out`: fused features [B, 304, h, w] (NCHW-shaped, any memory layout) -> logits [B, C, *input_shape] f32."""
        if self._stage_on_tape((self.segment.upsample,), out):
            return self._stage_apply("cls", tuple(input_shape), (self.segment.upsample,), out)
        with torch.no_grad():
            x = self._nchw_to_nhwc(out, self.compute_dtype)
            pred, _ = self._cls_hip(x, self.packed(), tuple(input_shape))
        return pred

    def forward_fusion(self, visual, fea_a):
        """cavp_model.py:143-154: projector MLP + sigmoid cross-modal attention.  visual [B, 304, h, w], fea_a [B, 304] ->
        (fea_v [B, 304, h, w], {"audio": [B, 304, 1, 1], "visual": projected features, "attn_v": [B, heads, h*w, 1]})."""
        mods = (self.visual_projector, self.cross_att)
        if self._stage_on_tape(mods, visual, fea_a):
            a_in = fea_a.reshape(fea_a.shape[0], -1)
            fus, vis, attn_v = self._stage_apply("fusion", None, mods, visual, a_in)
            return fus, {"audio": a_in[:, :, None, None], "visual": vis, "attn_v": attn_v}
        with torch.no_grad():
            v = self._nchw_to_nhwc(visual, self.compute_dtype)
            a = fea_a.reshape(fea_a.shape[0], -1).contiguous()
            if a.dtype != self.compute_dtype:
                a = ops.cast(a, torch.empty(a.shape, dtype=self.compute_dtype, device=a.device))
            fusion, fea_v_proj, attn = self._fusion_hip(v, a, self.packed())
        return (self._as_f32(fusion).permute(0, 3, 1, 2),
                {"audio": self._as_f32(a)[:, :, None, None], "visual": self._as_f32(fea_v_proj).permute(0, 3, 1, 2),
                 "attn_v": attn.unsqueeze(-1)})

    def _bank_and_shuffle(self, fea_a, shuffle_info, ow_flag):
        """Host half of forward_audio (cavp_model.py:160-173): SoundBank bookkeeping on the detached features and the shuffle
        index the second half of the batch is gathered with."""
        shuffle_idx = shuffle_info["shuffle_idx"]
        if ow_flag:
            f32 = self._as_f32(fea_a.detach())
            # (the overwritten copy is discarded by the reference too: `shuffle_fea_a` is re-assigned from fea_a right after)
            self.memory.overwrite_audio_feature(f32.clone()[shuffle_idx], f32, shuffle_info["mod_idx_map"])
            self.memory.update_bank(f32, shuffle_info["image_label"])
        return torch.as_tensor(shuffle_idx, device=fea_a.device).long()

    def forward_audio(self, audio, shuffle_info=None, ow_flag=False):
        """cavp_model.py:156-173: audio features of the B clips followed by the same features in shuffled order ([2B, 304]);
        with ow_flag the SoundBank is updated from the image labels.  With gradients enabled the encoder and the gather run on a
        training tape of their own (CAVPStageFunction); inside forward_train(audio_func=True) they are part of the model's pass."""
        if self._stage_on_tape((self.audio_backbone,), audio):
            return self._stage_apply("audio", (shuffle_info, ow_flag), (self.audio_backbone,), audio)
        with torch.no_grad():
            a = audio.contiguous()
            fea_a = self._audio_hip(a, self.packed())
            idx = self._bank_and_shuffle(fea_a, shuffle_info, ow_flag)
            return self._as_f32(torch.cat((fea_a, fea_a.index_select(0, idx)), dim=0))

    def forward_inference(self, image, audio=None):
        return self._forward_hip(image, audio, duplicate_visual=False)

    def forward_train(self, image, audio=None, shuffle_info=None, ow_flag=False, audio_func=False):
        """cavp_model.py:175-188.  audio_func=False (every reference trainer): `audio` holds 2B clips (matched | shuffled).
        audio_func=True: `audio` holds B clips and `shuffle_info` = {"shuffle_idx", "mod_idx_map", "image_label"}; the
        second half of the audio features is the first half gathered by shuffle_idx (forward_audio)."""
        if audio_func and shuffle_info is None:
            raise CavpError("audio_func=True needs shuffle_info (cavp_model.py:160-162 indexes it)")
        shuffle = (shuffle_info, ow_flag) if audio_func else None
        bn_train = any(m.training for m in self.modules() if isinstance(m, nn.modules.batchnorm._BatchNorm))
        want_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if not bn_train and not want_grad:
            return self._forward_hip(image, audio, duplicate_visual=True, shuffle=shuffle)   # frozen-BN, forward only
        if not image.is_cuda:
            raise CavpError("CAVP (MI355X path) needs inputs on a HIP device: there is no CPU fallback")
        from .train import CAVPTrainFunction
        # Parameters the forward never reads (position embeddings, VGGish's cls_head: checkpoint-only tensors) stay OUT of the
        # autograd node, as they are out of the reference's graph: DistributedDataParallel(find_unused_parameters=True)
        # (main_vpo_mono.py:131-135) marks them unused up front; as inputs of the node that get a None gradient they would look
        # "used", their reducer hooks would never fire and the SECOND iteration would fail with "expected to have finished
        # reduction in the prior iteration" (tests/test_gpu_wrappers.py).
        skip = {id(p) for p in self.params_without_grad()}
        params = [p for p in self.parameters() if p.requires_grad and id(p) not in skip] if want_grad else []
        graphed = getattr(self, "_graphed_autograd", None)
        if graphed is not None and want_grad and bn_train and not audio_func and audio is not None:
            # opt-in (enable_graphed_autograd): forward and backward as two hipGraph replays behind one autograd node
            from .train import GraphedTrainFunction, GraphedTrainStep
            step = next((g for g in graphed if g.matches(image, audio)), None)
            if step is None and len(graphed) < 4:
                try:
                    step = GraphedTrainStep(self, image, audio)
                    graphed.append(step)
                except Exception as ex:  # noqa: BLE001 - a failed capture must not cost the training run: eager node from here on
                    import warnings
                    warnings.warn(f"enable_graphed_autograd: hipGraph capture failed ({type(ex).__name__}: {ex}); using the eager "
                                  f"autograd node", RuntimeWarning)
                    torch.cuda.synchronize()
                    self.__dict__["_graphed_autograd"] = None
                    step = None
            elif step is None and not self.__dict__.get("_graphed_cap_warned"):
                import warnings
                self.__dict__["_graphed_cap_warned"] = True
                warnings.warn(f"enable_graphed_autograd: 4 input shapes are captured already; image {tuple(image.shape)} / audio "
                              f"{tuple(audio.shape)} (and any further new shape) runs on the eager autograd node", RuntimeWarning)
            if step is not None:
                out_pred, out_fusion, visual, audio_f, attn_v = GraphedTrainFunction.apply(step, image, audio, *params)
                return out_pred, out_fusion, {"audio": audio_f, "visual": visual, "attn_v": attn_v}
        self._train_shuffle = shuffle
        out_pred, out_fusion, visual, audio_f, attn_v = CAVPTrainFunction.apply(self, image, audio, *params)
        return out_pred, out_fusion, {"audio": audio_f, "visual": visual, "attn_v": attn_v}

    def enable_graphed_autograd(self, on: bool = True) -> None:
        """Opt-in: `model(image, audio)` in training mode captures forward and backward as two hipGraphs per input shape (up to 4
        shapes) and replays them behind one autograd node (cavp_amd/train.py::GraphedTrainStep) - the reference's trainer loop
        (`out = model(...)`, torch loss, `loss.backward()`, torch optimisers) at graph speed.  The returned tensors are static
        buffers, overwritten by the next forward.  Changing the compute dtype, the set of trainable parameters or train / eval
        mode of the BatchNorm layers needs a fresh `enable_graphed_autograd()`."""
        self.__dict__["_graphed_autograd"] = [] if on else None

    def params_without_grad(self):
        """Parameters the forward never touches (present in checkpoints only): torch leaves their .grad None."""
        ps = [self.cross_att.pos_embed_v, self.cross_att.pos_embed_a]
        return ps + list(self.audio_backbone.cls_head.parameters())

    def _late_grad_ids(self):
        """Parameters whose gradients the backward finishes last (see GradArena): backbone, ASPP, low-level reduce."""
        late = list(self.backbone.parameters())
        if hasattr(self.segment, "aspp"):
            late += list(self.segment.aspp.parameters()) + list(self.segment.reduce.parameters())
        return {id(p) for p in late}

    def train_step(self, image, audio, label, ignore_index: int = 255, loss_scale: float = 1.0, all_reduce: bool = True,
                   _split_hook=None, want_pred: bool = False):
        """MI355X-native fused training step (no torch.autograd): forward_train (batch-stat BN, audio 2B) -> HIP
        cross-entropy on `out[:B] + out[B:]*0` (trainer_cavp_vpo_mono.py:171,187) -> hand-written backward.  Every
        gradient lands in one flat f32 arena (`p.grad` are views of it).  With a torch.distributed process group the
        arena is all-reduced over RCCL and averaged (DDP semantics, main_vpo_mono.py:131-135) in two pieces: the range
        the backward completes early (head, attention, audio encoder: ~75 % of the bytes) is reduced asynchronously
        while the backbone backward still runs, the remainder at the end.
        The segmentation head is one fused op (upsample + CE + backward); `want_pred=True`
        additionally materialises the full-resolution prediction into `self._last_outputs[0]` (otherwise None).
        Returns the (local) loss as a 1-element device tensor."""
        from . import train_ops as T
        from .train import (GradArena, TrainPass, allreduce_arena_early, allreduce_arena_late, collectives_on, dist_world,
                            run_train_forward)
        if not image.is_cuda:
            raise CavpError("CAVP (MI355X path) needs inputs on a HIP device: there is no CPU fallback")
        arena = getattr(self, "_grad_arena", None)
        if arena is None or arena.flat.device != image.device:
            big = {id(m.weight) for m in self.modules() if isinstance(m, (nn.Linear, nn.Conv2d))
                   and m.weight.numel() >= _GRAD_OVERWRITE_MIN} if _GRAD_OVERWRITE_MIN > 0 else set()
            arena = self._grad_arena = GradArena(list(self.parameters()), image.device, late_ids=self._late_grad_ids(),
                                                 no_zero_ids=big)
        arena.zero()
        tp = TrainPass(self, self.compute_dtype, arena=arena)
        B, C = image.shape[0], self.num_classes
        with torch.no_grad():
            lo, fusion, fea_v_proj, fea_a, attn = run_train_forward(self, image.contiguous(), audio.contiguous(), tp)
            world = dist_world() if all_reduce else 1
            out_pred = None
            if tuple(label.shape[-2:]) != tuple(image.shape[-2:]):
                raise CavpError("train_step: label and image resolution differ (the reference interpolates to the image size)")
            if want_pred or not _FUSED_HEAD:
                out_pred = torch.empty((lo.t.shape[0], C) + tuple(image.shape[-2:]), dtype=torch.float32, device=image.device)
                ops.bilinear_to_nchw(lo.t[..., :C], out_pred, align_corners=False)
            if _FUSED_HEAD:
                # upsample + CE + their backward in one op: the full-resolution prediction never reaches HBM
                loss, g = T.upsample_ce_head(lo.t, label, B, C, ignore_index, grad_scale=loss_scale / world)
            else:
                loss, dl = T.ce_loss(out_pred, label, B, ignore_index, grad_scale=loss_scale / world)   # SUM over ranks == mean
                g = torch.zeros(lo.t.shape, dtype=lo.t.dtype, device=lo.t.device)
                T.bilinear_bwd_from_nchw(dl, g[..., :C], n_valid=B, align_corners=False)
            lo.set_g(g)
            early = []
            if _split_hook is not None:
                tp.on_early_final = _split_hook
            elif all_reduce and collectives_on():
                tp.on_early_final = lambda: early.append(allreduce_arena_early(arena))
            tp.backward()
            tp.finish_padded()
            if all_reduce and collectives_on():
                allreduce_arena_late(arena, early[0] if early else None)   # + joins the early collective
            for p in arena.params:
                p.grad = arena.views[id(p)] if id(p) in tp.touched else None   # untouched = None, as torch would leave it
        self._last_outputs = (out_pred, fusion, attn)
        self.params_changed()   # running_mean / running_var were updated in place by the BatchNorm kernels
        return loss

    def capture_train_step(self, image, audio, label, ignore_index: int = 255, loss_scale: float = 1.0,
                           split: Optional[bool] = None):
        """Capture forward_train + CE + backward (about 1000 kernel launches) into hipGraphs and return
        `replay() -> loss`.  `image`, `audio`, `label` are the static input buffers: copy new batches into them before
        each replay.  Weight packing is part of the graph, so replays always see the current parameters.
        Single process: ONE graph.  Data parallel (or `split=True`): TWO graphs cut where the early gradient range is
        final; replay() = graph 1 -> asynchronous RCCL all-reduce of that range -> graph 2 (rest of the backward, runs
        concurrently with the collective) -> all-reduce of the late range -> join."""
        from .train import _no_gc_during_capture, allreduce_arena_early, allreduce_arena_late, collectives_on, dist_world
        world = dist_world()
        if split is None:
            split = collectives_on()
        with torch.no_grad():
            self.train_step(image, audio, label, ignore_index, loss_scale, all_reduce=False)   # warm-up: workspace, arena
            torch.cuda.synchronize()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self.train_step(image, audio, label, ignore_index, loss_scale, all_reduce=False)
            torch.cuda.current_stream().wait_stream(side)
            if not split:
                graph = torch.cuda.CUDAGraph()
                # thread_local: other threads (RCCL's watchdog polls its events) must not invalidate the capture
                with _no_gc_during_capture(), torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    loss = self.train_step(image, audio, label, ignore_index, loss_scale / world, all_reduce=False)
                graphs = (graph,)
            else:
                g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                torch.cuda.synchronize()
                cap = torch.cuda.Stream()
                cap.wait_stream(torch.cuda.current_stream())
                with _no_gc_during_capture(), torch.cuda.stream(cap):
                    g1.capture_begin(capture_error_mode="thread_local")

                    def cut():
                        g1.capture_end()
                        g2.capture_begin(pool=g1.pool(), capture_error_mode="thread_local")   # shares (and keeps alive) graph 1's allocations

                    loss = self.train_step(image, audio, label, ignore_index, loss_scale / world, all_reduce=False,
                                           _split_hook=cut)
                    g2.capture_end()
                torch.cuda.current_stream().wait_stream(cap)
                graphs = (g1, g2)
        arena = self._grad_arena

        def replay():
            self.params_changed()   # the graph updates the running statistics (and is usually followed by an optimiser step)
            if self.seg_model == "PVT" and getattr(self, "_pvt_drop_scales", None) is None:
                from .pvt_train import refresh_drop_path
                refresh_drop_path(self.backbone, image.shape[0], image.device)   # the graph reads the persistent mask buffer
            graphs[0].replay()
            if len(graphs) == 2:
                work = allreduce_arena_early(arena)
                graphs[1].replay()
                allreduce_arena_late(arena, work)
            elif collectives_on():
                allreduce_arena_late(arena, None)
            return loss
        # keep alive: the graphs, and every scratch buffer whose address they baked in (ops.workspace never frees a buffer it
        # has handed out, see there)
        self._train_graph = graphs
        return replay

    def forward(self, image, audio=None, shuffle_info=None, ow_flag=False, eval_mode=False, audio_func=False):
        if eval_mode:
            return self.forward_inference(image, audio)
        return self.forward_train(image, audio, shuffle_info, ow_flag, audio_func=audio_func)


def load_reference_checkpoint(model: "CAVP", ckpt, strict: bool = False):
    """Load a checkpoint written by the reference's engine (engine/engine.py:91-99: {"model": model_v.state_dict(), ...},
    saved from the DDP / DataParallel-wrapped model, hence `module.`-prefixed keys) the way its evaluation scripts do
    (test_avs_semantic.py:204-205: `model_v.load_state_dict(torch.load(path)["model"], strict=False)` on the wrapped
    model).  `ckpt`: a path, the full checkpoint dict, or a bare state_dict; prefixes `module.` are stripped so the
    unwrapped MI355X model can take it.  Returns torch's (missing_keys, unexpected_keys) record."""
    if isinstance(ckpt, (str, bytes)) or hasattr(ckpt, "__fspath__"):
        ckpt = torch.load(ckpt, map_location="cpu")
    sd = ckpt["model"] if isinstance(ckpt, dict) and "model" in ckpt and not torch.is_tensor(ckpt["model"]) else ckpt
    clean = {}
    for k, v in sd.items():
        while k.startswith("module."):
            k = k[len("module."):]
        clean[k] = v
    res = model.load_state_dict(clean, strict=strict)
    model._packed = None   # kernel-ready (packed / BN-folded) parameters are rebuilt on the next forward
    return res
