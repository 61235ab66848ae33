#!/usr/bin/env python3
"""Synthetic-data training driver on the MI355X path: the counterpart of the reference's main_vpo_mono.py +
trainer_cavp_vpo_mono.py::train loop (SURVEY.md §8f row f3 "synthetic-data train driver") with every stage on HIP:

    waveform --MelFrontEnd--> log-mel  \
    image ----------------------------> CAVP.train_step (forward_train + CE + backward, one flat gradient arena,
                                         one RCCL all-reduce) --> FusedSGDAdam.step(poly lr)

One process per GPU (python -m torch.distributed.run --nproc-per-node N tools/train_synth.py ...), 127.0.0.1 rendezvous.
There is no dataset here (no network): images / waveforms / labels are seeded random tensors of the config_avss_binary
shapes; the point is the plumbing and its throughput, not accuracy.

usage: python tools/train_synth.py [--steps 20] [--batch 32] [--dtype bf16|f32] [--num-classes 2] [--from-waveform]
"""
import argparse
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from cavp_amd.hostinfo import cap_torch_threads
    cap_torch_threads()   # (container CPU quota: cavp_amd/hostinfo.py)
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32, help="frames per GPU per step")
    ap.add_argument("--dtype", choices=["bf16", "f32"], default="bf16")
    ap.add_argument("--num-classes", type=int, default=2)
    ap.add_argument("--hw", type=int, default=224)
    ap.add_argument("--lr", type=float, default=1e-3)            # config_avss_binary.py:52-57
    ap.add_argument("--lr-power", type=float, default=0.9)
    ap.add_argument("--momentum", type=float, default=0.9)
    ap.add_argument("--weight-decay", type=float, default=1e-4)
    ap.add_argument("--total-iters", type=int, default=1000)
    ap.add_argument("--seg-model", choices=["DeepLabV3Plus", "PVT"], default="DeepLabV3Plus", help="PVT = config #4's PVTv2-B5 backbone")
    ap.add_argument("--fixed-batch", action="store_true", help="train on ONE batch (over-fit sanity: the loss must fall)")
    ap.add_argument("--from-waveform", action="store_true", help="start from 16 kHz waveforms (HIP log-mel front-end)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from cavp_amd.audio_frontend import MelFrontEnd
    from cavp_amd.cavp_model import CAVP
    from cavp_amd.optim import FusedSGDAdam, warmup_poly_lr
    from cavp_amd.synth import synth_state_dict

    hyp = types.SimpleNamespace(seg_model=a.seg_model, last_three_dilation_stride=[False, False, False],
                                audio_backbone="vgg", num_classes=a.num_classes, batch_size=a.batch, local_rank=local,
                                audio_len=1.0, spec_min=-100, spec_max=100, allow_random_pvt=True)
    model = CAVP(50, None, num_classes=a.num_classes, args=hyp)
    model.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=1))
    model.train().to(dev).set_compute_dtype(torch.bfloat16 if a.dtype == "bf16" else torch.float32)
    sched = warmup_poly_lr(a.lr, a.lr_power, a.total_iters, 0)
    front = MelFrontEnd(hyp, device=dev) if a.from_waveform else None

    g = torch.Generator().manual_seed(1234 + rank)
    B = a.batch
    opt = None
    t0 = None
    first = None
    for it in range(a.steps):
        if a.fixed_batch:
            g = torch.Generator().manual_seed(1234 + rank)      # the same batch every step
        image = torch.randn(B, 3, a.hw, a.hw, generator=g).to(dev)
        label = torch.randint(0, a.num_classes, (B, a.hw, a.hw), generator=g).to(dev)
        if front is not None:                                   # matched clips ‖ shuffled clips = 2B (cavp_model.py:181)
            wave = (torch.randn(2 * B, 1, 16000, generator=g) * 0.1).to(dev)
            audio = front(wave)
        else:
            audio = (torch.rand(2 * B, 1, 96, 64, generator=g) * 2 - 1).to(dev)
        loss = model.train_step(image, audio, label)
        if opt is None:
            opt = FusedSGDAdam(model, model._grad_arena, a.lr, momentum=a.momentum, weight_decay=a.weight_decay)
        # the reference sets the learning rate AFTER the optimiser step (trainer_cavp_vpo_mono.py:193-203): step `it` runs with
        # the rate computed at the end of step it - 1 (the configured start rate for the first one)
        opt.step(sched(it - 1) if it > 0 else a.lr)
        if it == 1:                                             # skip the first two (allocation / warm-up) steps
            torch.cuda.synchronize()
            t0 = time.time()
        if first is None:
            first = float(loss.item())
        if rank == 0 and (it % 5 == 0 or it == a.steps - 1):
            print(f"iter {it:4d}  lr {sched(it):.3e}  loss {float(loss.item()):.4f}", flush=True)
    torch.cuda.synchronize()
    if rank == 0 and t0 is not None and a.steps > 2:
        dt = (time.time() - t0) / (a.steps - 2)
        print(f"{B * world / dt:.1f} frames/s over {world} GPU(s) ({dt * 1e3:.1f} ms/step incl. host-side input generation, "
              f"eager launches, optimiser step)")
    if rank == 0 and a.fixed_batch:
        print(f"fixed batch: loss {first:.4f} -> {float(loss.item()):.4f} after {a.steps} steps")
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
