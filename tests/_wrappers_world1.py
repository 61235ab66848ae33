"""Child process of tests/test_gpu_wrappers.py: the model inside the two wrappers the reference always puts around it
(main_vpo_mono.py:127-144) - `DDP(SyncBatchNorm.convert_sync_batchnorm(model), device_ids=[rank], find_unused_parameters=True)`
on a one-rank `nccl` process group, and `nn.DataParallel(model, device_ids=["cuda:0"])` - driven with the trainer's call
sequence (trainer_cavp_vpo_mono.py:166-193): `model_v_(image, audio, None, ow_flag)`, torch CrossEntropy on
`out[:B] + out[B:] * 0` + ContrastLoss on the fusion halves, `backward()`, SGD + Adam steps, zero_grad; TWO iterations (DDP's
unused-parameter bookkeeping fails in the second one when a "used" parameter got no gradient), through the eager autograd node
and through `enable_graphed_autograd()`.  Deterministic mode: every gradient and every updated weight equals the unwrapped run to f32 rounding."""
import os
import sys
import types

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.parallel import DistributedDataParallel as DDP

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from cavp_amd import _lib as CL
    from cavp_amd.contrast import ContrastLoss
    from cavp_amd.synth import synth_inputs, synth_state_dict
    from models.cavp_model import CAVP   # the reference's import path
    CL.set_deterministic(True, dev)
    C, B, hw = 3, 4, (64, 64)
    args = types.SimpleNamespace(seg_model="DeepLabV3Plus", last_three_dilation_stride=[False, False, False], audio_backbone="vgg",
                                 num_classes=C, batch_size=B, local_rank="cpu")
    sd = None
    batches = []
    for seed in (51, 52):
        image, audio, label = [t.to(dev) for t in synth_inputs(B, hw, audio_batch=2 * B, num_classes=C, seed=seed)]
        # blocky labels so that ContrastLoss finds classes with enough pixels
        label = torch.zeros_like(label)
        label[:, 8:40, 10:50] = 1
        label[:, 44:60, 20:60] = 2
        label[:, :2] = 255
        shuf = label.clone()
        shuf[1:] = 0
        batches.append((image, audio, label, shuf))

    def run(wrap, graphed):
        nonlocal sd
        torch.manual_seed(7)    # ContrastLoss draws its anchor permutations from the default generator
        m = CAVP(50, None, num_classes=C, audio_backbone_pretrain_path=None, visual_backbone=50, args=args)
        if sd is None:
            sd = synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=1)
        m.load_state_dict(sd, strict=True)
        m.to(dev).train()
        if graphed:
            m.enable_graphed_autograd()
        audio_ids = {id(p) for p in m.audio_backbone.parameters()}
        opt_v = torch.optim.SGD([p for p in m.parameters() if id(p) not in audio_ids], lr=1e-3, momentum=0.9, weight_decay=1e-4)
        opt_a = torch.optim.Adam(m.audio_backbone.parameters(), lr=1e-4)
        if wrap == "ddp":
            w = DDP(nn.SyncBatchNorm.convert_sync_batchnorm(m), device_ids=[0], find_unused_parameters=True)
        elif wrap == "dp":
            w = nn.DataParallel(m, device_ids=["cuda:0"])
        else:
            w = m
        crit = ContrastLoss(temperature=0.1, ignore_idx=255, max_views=64)
        inner = w.module if wrap != "none" else w
        grads, losses = [], []
        for image, audio, label, shuf in batches:
            out, fus, pack = w(image, audio, None, False)
            assert set(pack) == {"audio", "visual", "attn_v"}
            loss = F.cross_entropy(out[:B] + out[B:] * 0.0, label, ignore_index=255) + crit(fus[:B], label, fus[B:], shuf)
            loss.backward()
            grads.append({k: p.grad.detach().clone() for k, p in inner.named_parameters() if p.grad is not None})
            losses.append(float(loss.detach()))
            opt_v.step()
            opt_a.step()
            opt_v.zero_grad()
            opt_a.zero_grad()
            inner.params_changed()     # (a no-op for callers that never touch it: the version counters of the weights moved)
        never = {k for k, _ in inner.named_parameters() if k not in grads[0]}
        weights = {k: p.detach().clone() for k, p in inner.named_parameters()}
        return losses, grads, weights, never

    def check(tag, ref, got):
        (l0, g0, w0, n0), (l1, g1, w1, n1) = ref, got
        # (the model's kernels are bit-reproducible in this mode; torch's own loss reductions are not, so the bars sit at f32
        # rounding level instead of exact equality)
        assert all(abs(a - b) <= 2e-6 * max(1.0, abs(a)) for a, b in zip(l0, l1)), (tag, l0, l1)
        for it, (a, b) in enumerate(zip(g0, g1)):
            assert a.keys() == b.keys(), (tag, it, sorted(set(a) ^ set(b))[:5])
            for k in a:
                n = float(a[k].double().norm())
                assert float((a[k].double() - b[k].double()).norm()) <= 1e-4 * n + 1e-12, (tag, it, k, float((a[k] - b[k]).abs().max()), n)
        for k in w0:
            assert float((w0[k].double() - w1[k].double()).norm()) <= 1e-6 * float(w0[k].double().norm()) + 1e-12, (tag, "weights", k)

    for graphed in (False, True):
        ref = run("none", graphed)
        assert len(ref[1][1]) >= 150 and all(bool(torch.isfinite(v).all()) for v in ref[1][1].values()), len(ref[1][1])
        # the checkpoint-only tensors (position embeddings, VGGish cls_head) get no gradient, wrapped or not (torch semantics)
        assert any("pos_embed" in k for k in ref[3]) and any("cls_head" in k for k in ref[3])
        check(f"ddp graphed={graphed}", ref, run("ddp", graphed))
        check(f"dp graphed={graphed}", ref, run("dp", graphed))
        print(f"WRAPPERS graphed={graphed}: DDP(SyncBN, find_unused_parameters) and DataParallel == unwrapped over 2 iterations "
              f"(losses {ref[0][0]:.6f} {ref[0][1]:.6f}, {len(ref[1][1])} gradients)")
    print("WRAPPERS_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
