#!/usr/bin/env python3
"""Measures, under cavp_set_deterministic, the comparisons whose test bars were set above the f32-atomics noise:
graph replay vs eager at full size, linearity in the loss scale, RCCL-world-1 style repeat, audio_func=True vs explicit concat.
GPU box only; prints the distances so the bars in tests/ can be tightened to what the deterministic mode actually gives."""
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cavp_amd import _lib  # noqa: E402
from cavp_amd.synth import synth_inputs, synth_state_dict  # noqa: E402

DEV = torch.device("cuda", 0)


def build(C, B, lds=(False, False, False), dtype=torch.float32):
    from cavp_amd.cavp_model import CAVP
    args = types.SimpleNamespace(seg_model="DeepLabV3Plus", last_three_dilation_stride=list(lds), audio_backbone="vgg",
                                 num_classes=C, batch_size=B, local_rank="cpu")
    m = CAVP(50, None, num_classes=C, args=args)
    sd = synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=1)
    m.load_state_dict(sd, strict=True)
    m.train().to(DEV).set_compute_dtype(dtype)
    return m, sd


def grads(m):
    return {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}


def dist(g1, g2, s=1.0):
    tot = torch.sqrt(sum((v.double() ** 2).sum() for v in g2.values()))
    err = torch.sqrt(sum(((s * g1[k].double() - g2[k].double()) ** 2).sum() for k in g1))
    return float(err / tot)


def main():
    _lib.set_deterministic(True, DEV)
    C, B, HW = 2, 32, (224, 224)
    image, audio, label = [t.to(DEV) for t in synth_inputs(B, HW, audio_batch=2 * B, num_classes=C, seed=5)]
    m1, _ = build(C, B)
    m2, _ = build(C, B)
    l1 = float(m1.train_step(image, audio, label, loss_scale=1.0).item())
    g1 = grads(m1)
    l2 = float(m2.train_step(image, audio, label, loss_scale=4.0).item())
    g2 = grads(m2)
    print(f"loss-scale linearity (det): loss {l1:.8f} {l2:.8f}  |4 g1 - g2| / |g2| = {dist(g1, g2, 4.0):.3e}")
    m3, _ = build(C, B)
    l3 = float(m3.train_step(image, audio, label).item())
    g3 = grads(m3)
    print(f"repeat eager (det): loss diff {abs(l3 - l1):.3e}  |g3 - g1| / |g1| = {dist(g3, g1):.3e}")
    m4, _ = build(C, B)
    rep = m4.capture_train_step(image, audio, label)
    l4 = float(rep().item())
    torch.cuda.synchronize()
    print(f"graph replay vs eager (det): loss diff {abs(l4 - l1):.3e}  |g4 - g1| / |g1| = {dist(grads(m4), g1):.3e}")
    # small model: audio_func vs explicit concat (tests/test_gpu_boundary.py)
    C, B, hw = 3, 2, (64, 64)
    m, sd = build(C, B)
    image, audio, label = [t.to(DEV) for t in synth_inputs(B, hw, num_classes=C, seed=3)]
    idx = torch.tensor([1, 0], device=DEV)
    info = {"shuffle_idx": idx, "mod_idx_map": {}, "image_label": torch.tensor([[1, 1, 0], [1, 0, 1]], device=DEV)}
    crit = torch.nn.CrossEntropyLoss(ignore_index=255)

    def step(**kw):
        m.zero_grad(set_to_none=True)
        m.load_state_dict({k: v for k, v in sd.items() if "running_" in k or "num_batches" in k}, strict=False)
        out, fus, pack = m(image, **kw)
        loss = crit(out[:B] + out[B:] * 0.0, label)
        loss.backward()
        return out.detach().clone(), grads(m)
    o1, ga = step(audio=audio, shuffle_info=info, ow_flag=False, audio_func=True)
    o2, gb = step(audio=torch.cat((audio, audio[idx])), shuffle_info=None, ow_flag=False)
    worst_b = worst_o = 0.0
    cmin = 1.0
    for k in ga:
        a, b = ga[k].double().flatten(), gb[k].double().flatten()
        if float(b.norm()) == 0:
            continue
        rel, cos = float((a - b).norm() / b.norm()), float((a @ b) / (a.norm() * b.norm()))
        cmin = min(cmin, cos)
        if k.startswith("backbone.") or k.startswith("segment.aspp") or k.startswith("segment.reduce"):
            worst_b = max(worst_b, rel)
        else:
            worst_o = max(worst_o, rel)
    print(f"audio_func vs concat (det): out diff {float((o1 - o2).abs().max() / o2.abs().max()):.3e}  worst rel backbone {worst_b:.3e} other {worst_o:.3e}  min cos {cmin:.6f}")
    o3, gc = step(audio=torch.cat((audio, audio[idx])), shuffle_info=None, ow_flag=False)
    print(f"same call twice (det): out diff {float((o3 - o2).abs().max()):.3e}  grads {max(float((gc[k] - gb[k]).abs().max()) for k in gb):.3e}")


if __name__ == "__main__":
    main()
