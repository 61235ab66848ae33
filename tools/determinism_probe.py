"""Run-to-run reproducibility probe of the train-mode forward / backward (GPU).  Prints the largest difference between two
runs of the same step on the same inputs: the only sanctioned source is the order of f32 atomics in column reductions."""
import sys
import types

import torch

sys.path.insert(0, ".")
from cavp_amd.synth import synth_inputs, synth_state_dict  # noqa: E402


def build(B, C, dtype):
    from cavp_amd.cavp_model import CAVP
    args = types.SimpleNamespace(seg_model="DeepLabV3Plus", last_three_dilation_stride=[False, True, True],
                                 audio_backbone="vgg", num_classes=C, batch_size=B, local_rank="cpu")
    m = CAVP(50, None, num_classes=C, args=args)
    m.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=1), strict=True)
    m.train().to("cuda:0").set_compute_dtype(dtype)
    return m


def main():
    for B, hw in ((8, (64, 64)), (32, (224, 224))):
        for dtype in (torch.float32, torch.bfloat16):
            image, audio, label = [t.to("cuda:0") for t in synth_inputs(B, hw, audio_batch=2 * B, num_classes=2, seed=5)]
            outs, losses, grads = [], [], []
            for rep in range(3):
                m = build(B, 2, dtype)
                with torch.no_grad():
                    o = m(image, audio, None, False)
                outs.append([o[0].float().clone(), o[1].float().clone(), o[2]["visual"].float().clone(), o[2]["audio"].float().clone()])
                losses.append(float(m.train_step(image, audio, label).item()))
                grads.append(torch.cat([p.grad.flatten().double() for p in m.parameters() if p.grad is not None]))
            names = ["out_pred", "fusion", "visual_proj", "audio"]
            d = [max(float((outs[0][i] - outs[r][i]).abs().max()) for r in (1, 2)) for i in range(4)]
            s = [float(outs[0][i].abs().max()) for i in range(4)]
            gd = max(float((grads[0] - grads[r]).norm() / grads[0].norm()) for r in (1, 2))
            print(f"B={B} {hw} {dtype}: fwd max |diff| " + ", ".join(f"{n} {a:.2e} (max {b:.2g})" for n, a, b in zip(names, d, s))
                  + f"; losses {losses}; grad rel diff {gd:.2e}", flush=True)


if __name__ == "__main__":
    main()
