#!/usr/bin/env python3
"""bench.py — frames/s of the CAVP hot path on MI355X (driver contract in the task statement).

    python bench.py --gpus N --steps K --warmup W                  (N > 1 without a launcher: bench.py starts its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (`CAVP.forward`) over one batch of B=32 synthetic frames
(224x224 RGB + 96x64 log-mel) per GPU, inputs resident in HBM before the timed region.  Frames shard over the batch
(SURVEY.md §8e): every rank runs the same per-GPU batch (weak scaling) and the inference path needs no collective;
`value` = N * B * K / max-over-ranks(elapsed).

Extra objects on the JSON line:
  roofline     — the dominant kernel (MFMA implicit-GEMM conv/linear): algorithmic FLOPs / bytes of all its launches in
                 one step ÷ their summed duration, measured live with HIP events on the launching stream.
  cpu_baseline — the CPU oracle (pure-PyTorch restatement pinned to the reference by tests/golden) timed on this box's
                 host cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import types

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
MFMA_PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}   # dense matrix peaks, same guide


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="frames per GPU per step")
    ap.add_argument("--dtype", choices=["bf16", "f32"], default="bf16")
    ap.add_argument("--mode", choices=["train", "eval"], default="train",
                    help="train = forward_train + CE + backward (+ gradient all-reduce when N > 1): the BASELINE.json "
                         "metric; eval = inference forward only")
    ap.add_argument("--config", choices=["c1p", "c1", "c4", "c5"], default="c1p",
                    help="c1p = ResNet-50 224x224 OS16, 2 classes (config_avss_binary shape: the BASELINE metric's default); c1 = the same "
                         "model in BASELINE config #1's VPO-SS plumbing (OS8: layer3 / layer4 / ASPP at 28x28, 22 classes); "
                         "c4 = PVTv2-B5 512x512 (config #4; not the BASELINE metric); c5 = clip-shaped step of config #5: the c1p model on "
                         "--batch = 5 x clips frames (use --batch 30), loss = CE + ContrastLoss on the fusion halves, through the "
                         "autograd boundary (eager: the class-balanced sampling runs on the host)")
    ap.add_argument("--pmc", action="store_true",
                    help="measure roofline.traffic live: two extra rocprofv3 --pmc passes of this command (FETCH_SIZE, WRITE_SIZE) "
                         "through tools/pmc_traffic.py; default: the committed profiles/ figure, labelled as such")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--split-graph", action="store_true",
                    help="force the two-graph replay of the data-parallel path (cut where the early gradients are final) on one GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--lib", default="", help="A/B and anatomy runs: load this build of libcavp_hip.so (e.g. cavp_amd/libcavp_hip_profile.so from "
                                              "`python -m cavp_amd.build --profile`) instead of the product library")
    ap.add_argument("--grad-allreduce", choices=["f32", "bf16"], default="f32",
                    help="wire format of the data-parallel gradient all-reduce (--gpus > 1): f32 = the reference's DDP (479 MB per step), "
                         "bf16 = cast / sum / cast back, 240 MB per step on the xGMI links (cavp_amd.train.set_grad_allreduce_dtype)")
    ap.add_argument("--deterministic", action="store_true", help="opt-in bit-reproducible reductions (cavp_set_deterministic)")
    ap.add_argument("--no-token-fusion", action="store_true", help="A/B: GELU as a separate pass, duplicated token tensors copied")
    ap.add_argument("--no-side-stream", action="store_true", help="A/B: the audio encoder on the main stream instead of a second one")
    ap.add_argument("--no-group-wgrad", action="store_true", help="A/B: one weight-gradient launch (+ slab reduce) per layer instead of grouped launches")
    ap.add_argument("--no-rank1-attn", action="store_true", help="A/B: the cross-modal attention as q GEMM + gate + proj GEMM (round 3) instead of the one-key collapse")
    ap.add_argument("--wgrad-big", default="", help="A/B: cavp_set_wgrad_big MODE:SCHEDULE (mode 0 = the 256x256 weight-gradient tile where it qualifies, "
                                                     "1 = never; schedule 2 = 16 waves, 1 / 0 = 8 waves)")
    ap.add_argument("--no-branch-stream", action="store_true", help="A/B: the bottlenecks' down-sample branch on the main stream (rounds 1-4)")
    ap.add_argument("--no-side-packs", action="store_true", help="A/B: the audio encoder's weight re-packs on the main stream (rounds 1-4)")
    ap.add_argument("--no-bn-bwd-fusion", action="store_true", help="A/B: BatchNorm backward always as reduce launch + apply launch (rounds 1-4)")
    ap.add_argument("--no-tail-split", action="store_true", help="A/B: never split a 256x256-tile launch with a nearly empty last round")
    ap.add_argument("--trainer-loop", action="store_true",
                    help="the reference trainer's call sequence instead of the fused step: out = model(image, audio) -> torch "
                         "cross-entropy on out[:B] + out[B:]*0 -> loss.backward() (trainer_cavp_vpo_mono.py:166-193); with the graph "
                         "on (default) through CAVP.enable_graphed_autograd(), with --no-graph through the eager autograd node")
    ap.add_argument("--no-f32", action="store_true", help="skip the secondary f32 (parity path) training-step measurement")
    ap.add_argument("--no-eval-leg", action="store_true", help="skip the secondary eval-forward measurement (`eval_forward` object of the default line)")
    ap.add_argument("--cpu-sample-batch", type=int, default=8)
    ap.add_argument("--cpu-protocol-full", action="store_true",
                    help="run every leg of SURVEY.md 8(d)'s CPU-baseline protocol to the letter (eval B=2 and B=32, train B=2; 5 warm-ups + 20 timed each: minutes)")
    return ap.parse_args()


def model_cfg(name="c1p"):
    if name == "c4":   # config_avss.py shape: PVTv2-B5, 512x512, 71 classes
        return dict(C=71, lds=[False, False, False], hw=(512, 512), seg_model="PVT")
    if name == "c1":   # C1: config_vpo_ss.py plumbing at 224x224 - OS8 ([False, True, True]), 22 classes
        return dict(C=22, lds=[False, True, True], hw=(224, 224), seg_model="DeepLabV3Plus")
    # C1' (SURVEY.md §8d): config_avss_binary.py shape — 224x224, OS16, VGGish audio, num_classes=2
    return dict(C=2, lds=[False, False, False], hw=(224, 224), seg_model="DeepLabV3Plus")


def build_model(cfg, B, dtype, device):
    from cavp_amd.cavp_model import CAVP
    from cavp_amd.synth import synth_state_dict
    args = types.SimpleNamespace(seg_model=cfg.get("seg_model", "DeepLabV3Plus"), last_three_dilation_stride=cfg["lds"], audio_backbone="vgg", allow_random_pvt=True,
                                 num_classes=cfg["C"], batch_size=B, local_rank="cpu")
    m = CAVP(50, None, num_classes=cfg["C"], audio_backbone_pretrain_path=None, visual_backbone=50, args=args)
    sd = synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=1)
    m.load_state_dict(sd, strict=True)
    m.eval().to(device).set_compute_dtype(dtype)
    return m, sd


def live_taps(h_in, w_in, h_out, w_out, kh, kw, stride, pad, dil):
    """Taps of a conv that touch the image for at least one output position: the kernels skip the others (ASPP d = 12 / 18 on
    14x14), so they are not algorithmic work either."""
    def axis(n_in, n_out, k):
        return sum(1 for t in range(k) if any(0 <= o * stride - pad + t * dil < n_in for o in range(n_out)))
    return axis(h_in, h_out, kh) * axis(w_in, w_out, kw)


# Whole-step byte model: SURVEY.md section 8d's "perfectly fused" forward minimum (C1', bf16, B = 32:
# 147.4 MB per frame) x the train-mode forward factor (57.8 / 39.41 GFLOP: the fusion block and the decoder head run on 2B)
# x 3 (forward: every activation written once and read once; backward: every saved activation read once, every gradient
# written once and read once; weights and their f32 gradients are < 3 % of that and are left out).
FUSED_MIN_MB_PER_FRAME_BF16 = {"train": 147.4 * (57.8 / 39.41) * 3.0, "eval": 147.4}
# C1 (OS8, 22 classes): 221.9 MB per frame; its train-mode forward adds the same 18.39 GFLOP of 2B work
# (audio, projector, fusion, head) to 82.75 GFLOP
FUSED_MIN_MB_PER_FRAME_BF16_C1 = {"train": 221.9 * ((82.75 + 57.8 - 39.41) / 82.75) * 3.0, "eval": 221.9}
# C4 (PVTv2-B5, 512 x 512, 71 classes): derived in round 4 by tools/byte_model.py - forward hooks on the REFERENCE model with the
# survey's rule (every conv / linear / pool reads its input once and writes its output once, one extra read per residual join,
# both interpolates in + out, weights once per batch); the same hooks give 71.98 M elements per frame for C1' against the
# survey's 70.07 M (+2.7 %).  Eval forward 612.50 M elements per frame, train-mode forward (fusion block and head on 2B) 706.64 M,
# weights 160.34 M elements per batch.
C4_ELEMS_PER_FRAME = {"eval": 612.50e6, "train_forward": 706.64e6, "weights": 160.34e6}


def fused_min_mb_per_frame_c4(mode, batch):
    if mode == "eval":
        return (C4_ELEMS_PER_FRAME["eval"] + C4_ELEMS_PER_FRAME["weights"] / batch) * 2 / 1e6
    return (C4_ELEMS_PER_FRAME["train_forward"] + C4_ELEMS_PER_FRAME["weights"] / batch) * 2 * 3.0 / 1e6


class KernelTimer:
    """HIP-event timing of every libcavp_hip launch on the launching (current) stream."""

    # wrappers whose Python side marshals a job table before the launch (16 cavp_wgrad_job structs: ~0.3 ms of host time)
    HEAVY_HOST = ("conv2d_wgrad_group",)

    def __init__(self):
        self.records = []
        # Round 3's timer recorded e0, ran the Python wrapper, recorded e1: on a drained stream e0 completes at once and the
        # wrapper's HOST time (argument marshalling, the launch itself) was counted as kernel time - the grouped weight gradients
        # read 8.3 ms per step against 3.5 ms in the rocprofv3 trace.  Now a filler kernel (torch.cuda._sleep) is queued in front
        # of e0: the device is still busy with it while the host marshals and launches, so e0 .. e1 brackets device time only.
        # The filler is sized from a calibration of _sleep on this box; an event pair behind a filler with nothing between its
        # records still reads a few microseconds apart: that is calibrated too and subtracted.
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(1000)
        torch.cuda.synchronize()
        e0.record()
        torch.cuda._sleep(2_000_000)
        e1.record()
        torch.cuda.synchronize()
        cyc_per_us = 2_000_000 / max(e0.elapsed_time(e1) * 1e3, 1e-3)
        self.filler_cycles = {False: int(60 * cyc_per_us), True: int(1500 * cyc_per_us)}   # ~60 us / ~1.5 ms of device time
        pairs = []
        for _ in range(64):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda._sleep(self.filler_cycles[False])
            e0.record()
            e1.record()
            pairs.append((e0, e1))
        torch.cuda.synchronize()
        t = sorted(a.elapsed_time(b) for a, b in pairs)
        self.overhead_ms = t[len(t) // 2]

    def _ms(self, e0, e1):
        return max(0.0, e0.elapsed_time(e1) - self.overhead_ms)

    def wrap(self, ops_mod, train_mod=None):
        self._orig = {}
        for name in ("conv2d", "conv3x3_smallcin_nchw", "maxpool", "global_avgpool", "bilinear", "bilinear_to_nchw",
                     "layernorm", "attn_gate", "cast"):
            fn = getattr(ops_mod, name)
            self._orig[(ops_mod, name)] = fn
            setattr(ops_mod, name, self._timed(name, fn))
        if train_mod is not None:
            for name in ("conv2d_dgrad", "conv2d_wgrad", "conv2d_wgrad_group", "colstats", "colsum", "scale_shift_act", "bn_act_bwd_reduce",
                         "bn_act_bwd_apply", "act_bwd", "add", "layernorm_bwd", "attn_gate_bwd", "maxpool_bwd",
                         "bilinear_bwd", "bilinear_bwd_from_nchw", "bcast_add", "ce_loss", "upsample_ce_head", "smallcin_wgrad",
                         "unpack_weight_grad", "pack_weight_dgrad", "bn_finalize_tiles", "bn_bwd_sum_tiles", "col_tile_stats", "colsum_groups",
                         "layernorm_bwd_add"):
                if not hasattr(train_mod, name):
                    continue
                fn = getattr(train_mod, name)
                self._orig[(train_mod, name)] = fn
                setattr(train_mod, name, self._timed(name, fn))
        return self

    def unwrap(self, *_):
        for (mod, name), fn in self._orig.items():
            setattr(mod, name, fn)

    def _timed(self, name, fn):
        def run(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda._sleep(self.filler_cycles[name in self.HEAVY_HOST])   # keeps the device busy while the host marshals + launches
            e0.record()
            r = fn(*a, **k)
            e1.record()
            flops = nbytes = 0
            if name in ("conv2d_dgrad", "conv2d_wgrad"):
                # dgrad(dy, w_t, dx): same MACs as the forward conv = M_dy * Cout_f * Cin_f * k * k (stride-1 form)
                t0, t1, t2 = a[0], a[1], a[2]
                es = t0.element_size()
                kw_ = dict(kh=k.get("kh", 1), kw=k.get("kw", 1), stride=k.get("stride", 1), pad=k.get("pad", 0), dil=k.get("dil", 1))
                if name == "conv2d_dgrad":   # (dy, w_t, dx): the MACs of the forward conv dx -> dy, live taps only
                    kk = live_taps(t2.shape[1], t2.shape[2], t0.shape[1], t0.shape[2], **kw_)
                    m_dy = t0.shape[0] * t0.shape[1] * t0.shape[2]
                    flops = 2 * m_dy * t0.shape[3] * t2.shape[3] * kk
                    nbytes = (t0.numel() + t1.numel() + t2.numel()) * es
                else:                        # (x, dy, dw)
                    kk = live_taps(t0.shape[1], t0.shape[2], t1.shape[1], t1.shape[2], **kw_)
                    m_dy = t1.shape[0] * t1.shape[1] * t1.shape[2]
                    flops = 2 * m_dy * t1.shape[3] * t0.shape[3] * kk
                    nbytes = (t0.numel() + t1.numel()) * es + t2.numel() * 4
            if name == "conv2d_wgrad_group":   # one launch for up to 16 layers: sum over its jobs
                for j in a[0]:
                    t0, t1 = j["x"], j["dy"]
                    kk = live_taps(t0.shape[1], t0.shape[2], t1.shape[1], t1.shape[2], j["kh"], j["kw"], j["stride"], j["pad"], j["dil"])
                    flops += 2 * t1.shape[0] * t1.shape[1] * t1.shape[2] * t1.shape[3] * t0.shape[3] * kk
                    nbytes += (t0.numel() + t1.numel()) * t0.element_size() + j["dw"].numel() * 4
            if name == "conv2d":
                x, w, out = a[0], a[1], a[2]
                es = x.element_size()
                m = out.shape[0] * out.shape[1] * out.shape[2]
                kk = live_taps(x.shape[1], x.shape[2], out.shape[1], out.shape[2], k.get("kh", 1), k.get("kw", 1),
                               k.get("stride", 1), k.get("pad", 0), k.get("dil", 1))
                flops = 2 * m * out.shape[3] * x.shape[3] * kk
                nbytes = (x.shape[0] * x.shape[1] * x.shape[2] * x.shape[3] + w.numel() + m * out.shape[3]) * es
                if k.get("residual") is not None:
                    nbytes += m * out.shape[3] * es
            desc = ""
            if name not in ("conv2d", "conv2d_dgrad", "conv2d_wgrad", "conv2d_wgrad_group"):
                # pointwise / reduction / pooling / resampling kernels (BatchNorm apply and backward, LayerNorm, max-pool, bilinear,
                # CE head, casts ...): the byte floor of such a kernel is every tensor it is handed once - inputs read once, outputs
                # written once (per-channel vectors included; an in-place accumulation counts its tensor once although it is read
                # and written: the figure is a FLOOR).  Views count their own elements only (channel slices of a concat buffer).
                def tb(v):
                    if isinstance(v, torch.Tensor):
                        return v.numel() * v.element_size()
                    if isinstance(v, (tuple, list)):
                        return sum(tb(q) for q in v)
                    return 0
                nbytes = sum(tb(v) for v in a) + sum(tb(v) for v in k.values())
                big = max((v for v in list(a) + list(k.values()) if isinstance(v, torch.Tensor)), key=lambda t: t.numel(), default=None)
                if big is not None:
                    desc = f"{tuple(big.shape)} {str(big.dtype).replace('torch.', '')}"
            if name in ("conv2d_dgrad", "conv2d_wgrad"):
                desc = f"{tuple(a[0].shape)} {tuple(a[1].shape)} -> {tuple(a[2].shape)} k{k.get('kh', 1)} s{k.get('stride', 1)} d{k.get('dil', 1)}"
            if name == "conv2d_wgrad_group":
                desc = f"{len(a[0])} weight gradients: " + " ".join(f"{tuple(j['dw'].shape)}" for j in a[0])
            if name == "conv2d":
                desc = (f"x{tuple(a[0].shape)} -> y{tuple(a[2].shape)} k{k.get('kh', 1)} s{k.get('stride', 1)} "
                        f"d{k.get('dil', 1)}")
            self.records.append((name, e0, e1, flops, nbytes, desc))
            return r
        return run

    def per_layer(self, reps):
        torch.cuda.synchronize()
        n = len(self.records) // reps
        rows = []
        for i in range(n):
            name, _, _, fl, nb, desc = self.records[i]
            ms = sum(self._ms(self.records[i + r * n][1], self.records[i + r * n][2]) for r in range(reps)) / reps
            rows.append(f"{i:3d} {name:22s} {ms * 1e3:9.1f} us {fl / 1e9:9.2f} GF {fl / (ms * 1e-3) / 1e12 if ms > 0 else 0:8.2f} TF/s "
                        f"{nb / 1e6:8.1f} MB {nb / (ms * 1e-3) / 1e9 if ms > 0 else 0:8.1f} GB/s  {desc}")
        return "\n".join(rows)

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for name, e0, e1, fl, nb, _ in self.records:
            a = agg.setdefault(name, [0, 0.0, 0, 0])
            a[0] += 1
            a[1] += self._ms(e0, e1)
            a[2] += fl
            a[3] += nb
        return agg

    def launch_roof_ms(self, names, peak_flops, peak_bytes):
        """Sum over the launches of `names` of max(flops / peak_flops, bytes / peak_bytes): the time the launches would take if
        every one of them ran AT its own roof (a step mixes HBM-bound 1x1 layers with MFMA-bound 3x3 ones)."""
        return sum(max(fl / peak_flops, nb / peak_bytes) for name, _, _, fl, nb, _ in self.records if name in names) * 1e3


def measure_roofline(model, run_step, image, dtype_name, reps=3, config="c1p", live_pmc=False, trainer_loop=False):
    from cavp_amd import ops, train_ops
    import cavp_amd.train as _tr
    side_was = _tr._SIDE_STREAM
    _tr._SIDE_STREAM = False   # per-launch durations are taken with one stream: a co-running branch would inflate them
    # the same eager step without the timer: the sum of the per-kernel times must fit into it (see `host_bound` below)
    with torch.set_grad_enabled(config == "c5" or trainer_loop):
        run_step()
        torch.cuda.synchronize()
        t_e = time.perf_counter()
        for _ in range(reps):
            run_step()
        torch.cuda.synchronize()
        eager_ms = (time.perf_counter() - t_e) / reps * 1e3
    kt = KernelTimer().wrap(ops, train_ops)
    try:
        with torch.set_grad_enabled(config == "c5" or trainer_loop):   # (these back-propagate through torch.autograd; the fused steps need none)
            for _ in range(reps):
                run_step()
        agg = kt.summary()
        roof_ms = kt.launch_roof_ms(("conv2d", "conv2d_dgrad"), MFMA_PEAK_TFLOPS[dtype_name] * 1e12, HBM_PEAK_GBS * 1e9) / reps
        if os.environ.get("CAVP_BENCH_PER_LAYER"):
            with open(os.environ["CAVP_BENCH_PER_LAYER"], "w") as f:
                f.write(kt.per_layer(reps) + "\n")
    finally:
        kt.unwrap(ops)
        _tr._SIDE_STREAM = side_was
    # dominant kernel = igemm_kernel: forward convs / linears (conv2d) + data gradients (conv2d_dgrad, same kernel)
    launches, ms, flops, nbytes = agg["conv2d"]
    if "conv2d_dgrad" in agg:
        l2, m2, f2, b2 = agg["conv2d_dgrad"]
        launches, ms, flops, nbytes = launches + l2, ms + m2, flops + f2, nbytes + b2
    launches //= reps
    ms /= reps
    flops //= reps
    nbytes //= reps
    total_ms = sum(v[1] for v in agg.values()) / reps
    tflops = flops / (ms * 1e-3) / 1e12
    gbs = nbytes / (ms * 1e-3) / 1e9
    frac_mfma = tflops / MFMA_PEAK_TFLOPS[dtype_name]
    frac_hbm = gbs / HBM_PEAK_GBS
    # which roof bounds the kernel follows from the roofline MODEL (arithmetic intensity of its launches vs the ridge point
    # peak_flops / peak_bytes), not from whichever fraction happens to be larger
    ridge = MFMA_PEAK_TFLOPS[dtype_name] * 1e12 / (HBM_PEAK_GBS * 1e9)
    if flops / max(nbytes, 1) >= ridge:
        roof = {"bound": "mfma", "achieved": round(tflops, 2), "peak": MFMA_PEAK_TFLOPS[dtype_name], "unit": "TFLOP/s",
                "frac": round(frac_mfma, 4)}
    else:
        roof = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(frac_hbm, 4)}
    roof["intensity_flop_per_byte"] = round(flops / max(nbytes, 1), 1)
    roof["ridge_flop_per_byte"] = round(ridge, 1)
    traffic = None
    mode_name = "train" if getattr(model, "training", False) else "eval"
    tj = None
    if live_pmc:   # two rocprofv3 --pmc passes of this very command (tools/pmc_traffic.py); PMC counters cannot be read in-process
        sys.path.insert(0, os.path.join(REPO, "tools"))
        import pmc_traffic
        tj = pmc_traffic.measure(mode_name, config=config, batch=image.shape[0], dtype=dtype_name)
        tsrc = "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (tools/pmc_traffic.py)"
    else:
        cname = "" if config == "c1p" else config + "_"
        tname = f"traffic_{cname}{dtype_name}.json" if mode_name == "eval" else f"traffic_{cname}train_{dtype_name}.json"
        tpath = next((q for q in (os.path.join(REPO, "profiles", r + tname) for r in ("r06_", "r05_", "r04_", "r03_", "r02_", "r01_")) if os.path.exists(q)), "")
        if tpath:
            with open(tpath) as f:
                tj = json.load(f)
            tsrc = "profiles/" + os.path.basename(tpath) + " (committed measurement, not taken in this run; --pmc measures it live)"
    if tj is not None and tj.get("batch") == image.shape[0] and tj.get("dtype") == dtype_name:
        traffic = {"hbm_bytes_per_step": tj["hbm_bytes_per_step"], "hbm_bytes_per_launch": tj["hbm_bytes_per_launch"],
                   "all_kernels_hbm_bytes_per_step": tj.get("all_kernels_hbm_bytes_per_step"),
                   "vs_algorithmic": round(tj["hbm_bytes_per_step"] / nbytes, 3), "source": tsrc}
    roof.update({
        "traffic": traffic,
        "kernel": "igemm_kernel (cavp_conv2d_nhwc: all conv / linear launches of one step)",
        "launches_per_step": launches,
        "avg_launch_us": round(ms * 1e3 / launches, 2),
        "event_pair_overhead_us": round(kt.overhead_ms * 1e3, 2),
        "algorithmic_gflop_per_step": round(flops / 1e9, 2),
        "algorithmic_mb_per_step": round(nbytes / 1e6, 1),
        "achieved_tflops": round(tflops, 2), "achieved_gbs": round(gbs, 1),
        "frac_of_mfma_peak": round(frac_mfma, 4), "frac_of_hbm_peak": round(frac_hbm, 4),
        "share_of_step_kernel_time": round(ms / total_ms, 3),
        # launch-level view: every launch against ITS roof (max of its MFMA and HBM times), summed - the aggregate `frac` above
        # prices HBM-bound 1x1 layers and MFMA-bound 3x3 layers against one roof
        "launch_level": {"sum_of_launch_roofs_ms": round(roof_ms, 3), "measured_ms": round(ms, 3), "frac": round(roof_ms / ms, 4)},
        "other_kernels_ms": {k: round(v[1] / reps, 3) for k, v in agg.items() if k not in ("conv2d", "conv2d_dgrad")},
        # the pointwise / reduction families against the HBM roof: byte floor = every tensor handed to the launch once (KernelTimer)
        "other_kernels": {k: {"launches_per_step": v[0] // reps, "ms_per_step": round(v[1] / reps, 3),
                              "floor_mb_per_step": round(v[3] / reps / 1e6, 1),
                              "achieved_gbs": round(v[3] / max(v[1], 1e-9) / 1e6, 1),
                              "frac_of_hbm_peak": round(v[3] / max(v[1], 1e-9) / 1e6 / HBM_PEAK_GBS, 4)}
                          for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])
                          if k not in ("conv2d", "conv2d_dgrad", "conv2d_wgrad", "conv2d_wgrad_group") and v[3] > 0},
    })
    step_flops = flops
    if "conv2d_wgrad" in agg or "conv2d_wgrad_group" in agg:
        wl, wms, wf, wb = (sum(agg[k][i] for k in ("conv2d_wgrad", "conv2d_wgrad_group") if k in agg) for i in range(4))
        wms /= reps
        step_flops += wf // reps
        roof["wgrad_kernel"] = {"launches_per_step": wl // reps, "ms_per_step": round(wms, 3),
                                "achieved_tflops": round(wf / reps / (wms * 1e-3) / 1e12, 2),
                                "frac_of_mfma_peak": round(wf / reps / (wms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS[dtype_name], 4)}
    # consistency of the event timing: everything the timer saw, summed, against the untimed eager step.  A sum above the
    # step means host time leaked into the "kernel" intervals (round 3's driver line: 20.9 ms of kernels in a 15.06 ms step).
    roof["kernel_sum_ms"] = round(total_ms, 3)
    roof["eager_step_ms"] = round(eager_ms, 3)
    roof["host_bound"] = bool(total_ms > 1.1 * eager_ms)
    rp = rocprof_igemm_ms(config, mode_name, dtype_name)
    if rp is not None:
        roof["igemm_ms_rocprof"] = rp   # the committed rocprofv3 kernel trace of the same command, beside the event figure
        # Two clocks on the same launches: HIP events around each launch (live, above) and the committed rocprofv3 trace of the
        # graph replays (igemm_kernel* + igemm_big_kernel + the split-K epilogues).  `frac` is the SMALLER of the two fractions.
        rp_ms = rp.get("igemm_ms_per_step")
        if rp_ms and image.shape[0] == rp.get("batch", 32) and config == "c1p":
            roof["frac_event_timing"] = roof["frac"]
            t_r, g_r = flops / (rp_ms * 1e-3) / 1e12, nbytes / (rp_ms * 1e-3) / 1e9
            frac_r = t_r / MFMA_PEAK_TFLOPS[dtype_name] if roof["bound"] == "mfma" else g_r / HBM_PEAK_GBS
            roof["frac_rocprof_timing"] = round(frac_r, 4)
            roof["timing_sources_agree_within"] = round(abs(rp_ms - ms) / ms, 3)
            if frac_r < roof["frac"]:
                roof["frac"] = round(frac_r, 4)
                roof["achieved"] = round(t_r, 2) if roof["bound"] == "mfma" else round(g_r, 1)
                roof["frac_source"] = "committed rocprofv3 trace (graph replays only; smaller than the live HIP-event figure)"
            else:
                roof["frac_source"] = "live HIP events (smaller than the committed rocprofv3 figure)"
    roof["_step_gflop"] = step_flops / 1e9   # consumed by main(): the whole-step object needs ms_per_step of the timed region
    roof["_measured_step_bytes"] = traffic["all_kernels_hbm_bytes_per_step"] if traffic and "all_kernels_hbm_bytes_per_step" in traffic else None
    return roof


def rocprof_igemm_ms(config, mode_name, dtype_name):
    """igemm kernel time per step from the committed rocprofv3 --kernel-trace --stats summary of this command
    (profiles/rNN_rocprof_igemm_<cfg><mode>_<dtype>.json, written by tools/summarize_rocprof.py --igemm-json), newest round first."""
    cname = "" if config == "c1p" else config + "_"
    for r in ("r06_", "r05_", "r04_", "r03_"):
        q = os.path.join(REPO, "profiles", f"{r}rocprof_igemm_{cname}{mode_name}_{dtype_name}.json")
        if os.path.exists(q):
            with open(q) as f:
                j = json.load(f)
            j["source"] = "profiles/" + os.path.basename(q)
            return j
    return None


def eval_forward_leg(model, image, audio, B, dtype_name, steps=30, warmup=5):
    """Inference forward of the same model (eval-mode BN folded into the conv epilogues) on one hipGraph: frames/s, ms and the
    fraction of the HBM roof north_star's target is stated on (fused byte minimum of SURVEY.md section 8d / 8 TB/s)."""
    was_training = model.training
    model.eval()
    try:
        with torch.no_grad():
            model(image, audio, eval_mode=True)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                model(image, audio, eval_mode=True)
            torch.cuda.current_stream().wait_stream(side)
            from cavp_amd.train import _no_gc_during_capture
            with _no_gc_during_capture(), torch.cuda.graph(graph, capture_error_mode="thread_local"):
                model(image, audio, eval_mode=True)
            for _ in range(warmup):
                graph.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                graph.replay()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
    finally:
        model.train(was_training)
    min_gb = FUSED_MIN_MB_PER_FRAME_BF16["eval"] * B / 1e3
    out = {"value": round(B / ms * 1e3, 1), "unit": "frames/s", "ms_per_step": round(ms, 3), "steps": steps, "warmup": warmup, "dtype": dtype_name,
           "launch": "hipGraph replay", "bound": "hbm", "fused_min_gb": round(min_gb, 3), "achieved_gbs": round(min_gb / ms * 1e3, 1),
           "frac_of_hbm_peak": round(min_gb / ms * 1e3 / HBM_PEAK_GBS, 4), "target": {"frames_per_s": 400, "frac_of_hbm_peak": 0.60}}
    tpath = next((q for q in (os.path.join(REPO, "profiles", f"{r}traffic_{dtype_name}.json") for r in ("r06_", "r05_", "r04_", "r03_")) if os.path.exists(q)), "")
    if tpath:
        with open(tpath) as f:
            tj = json.load(f)
        if tj.get("batch") == B and tj.get("all_kernels_hbm_bytes_per_step"):
            out["measured_hbm_gb"] = round(tj["all_kernels_hbm_bytes_per_step"] / 1e9, 2)
            out["measured_vs_fused_min"] = round(tj["all_kernels_hbm_bytes_per_step"] / 1e9 / min_gb, 2)
            out["traffic_source"] = "profiles/" + os.path.basename(tpath) + " (committed PMC measurement of `bench.py --mode eval`, not taken in this run)"
    return out


def graphed_step_roofline(model, image, audio, label, ms_step, config, B, dtype_name):
    """Whole-step roofline object for the lines whose launches sit inside graph replays (--trainer-loop, c5): algorithmic FLOPs =
    the conv / linear / weight-gradient launches of ONE eager fused step of the same model on the same batch, counted by the
    launch wrappers (durations unused); bytes = the perfectly-fused minimum of the C1' model (the ContrastLoss of c5 adds < 1 %)."""
    from cavp_amd import ops, train_ops
    kt = KernelTimer().wrap(ops, train_ops)
    try:
        with torch.no_grad():
            model.train_step(image, audio, label, all_reduce=False)
        agg = kt.summary()
    finally:
        kt.unwrap(ops)
    gflop = sum(v[2] for k, v in agg.items() if k in ("conv2d", "conv2d_dgrad", "conv2d_wgrad", "conv2d_wgrad_group")) / 1e9
    min_gb = FUSED_MIN_MB_PER_FRAME_BF16["train"] * B / 1e3
    return {"bound": "hbm", "ms_per_step": round(ms_step, 3), "algorithmic_gflop": round(gflop, 1), "fused_min_gb": round(min_gb, 2),
            "achieved_tflops": round(gflop / ms_step, 2), "frac_of_mfma_peak": round(gflop / ms_step / MFMA_PEAK_TFLOPS[dtype_name], 4),
            "achieved_gbs": round(min_gb / ms_step * 1e3, 1), "frac_of_hbm_peak": round(min_gb / ms_step * 1e3 / HBM_PEAK_GBS, 4),
            "note": "FLOPs counted on one eager fused step of the same model and batch; the timed region is the trainer loop"}


def _host_cpu():
    """(physical cores, model name, sockets) of the box."""
    cores = os.cpu_count() or 1
    try:
        import psutil
        cores = psutil.cpu_count(logical=False) or cores
    except Exception:
        pass
    model, sockets = "unknown", set()
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name") and model == "unknown":
                    model = ln.split(":", 1)[1].strip()
                if ln.startswith("physical id"):
                    sockets.add(ln.split(":", 1)[1].strip())
    except OSError:
        pass
    return cores, model, max(1, len(sockets))


def _cpu_quota():
    from cavp_amd.hostinfo import cpu_quota
    return cpu_quota()


def cap_host_threads():
    """torch's intra-op pool no larger than the CPUs the container may use (cavp_amd/hostinfo.py has the story)."""
    from cavp_amd.hostinfo import cap_torch_threads
    return cap_torch_threads()


def _timed_iters(fn, budget_s, min_iters, max_iters, warm=True):
    """One untimed warm-up call, then >= min_iters timed calls (more while the budget lasts); (iterations, seconds, median s)."""
    if warm:
        fn()
    ts = []
    t_all = time.perf_counter()
    while len(ts) < max_iters and (len(ts) < min_iters or time.perf_counter() - t_all < budget_s):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return len(ts), sum(ts), ts[len(ts) // 2]


def cpu_baseline_train(sd, cfg, sample_batch):
    """Oracle forward_train + CE + autograd backward on the host cores, bounded sample.  torch's CPU convolutions do not scale
    to a whole two-socket box on this model (round 2 measured 128 threads at 1.1 frames/s against 0.9 on ONE), so the thread
    count is swept - one timed step each at {4, 8, 16, 32, 64, all cores} - and the best count is then timed properly (median of
    >= 3 further steps).  `cores` = the threads of the reported figure; the sweep and a single-thread figure travel with it."""
    from cavp_amd.synth import synth_inputs
    from oracle import cavp_oracle as O
    cores, cpu_model, sockets = _host_cpu()

    def make(batch):
        image, audio, label = synth_inputs(batch, cfg["hw"], audio_batch=2 * batch, num_classes=cfg["C"], seed=0)
        params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running_" not in k}
        sd2 = dict(sd)
        sd2.update(params)

        def step():
            for q in params.values():
                q.grad = None
            out, _, _ = O.cavp_forward(sd2, image, audio, cfg["lds"], eval_mode=False)
            O.ce_loss_train(out, label, batch).backward()
        return step

    step = make(sample_batch)
    quota = _cpu_quota()
    usable = cores if quota is None else min(cores, quota)   # more threads than the container's CPU quota only thrash
    torch.set_num_threads(min(16, usable))
    step()   # warm-up (allocator, oneDNN primitive caches)
    sweep = {}
    for nt in sorted({t for t in (4, 8, 16, 32, 64, usable, cores) if t <= cores}):
        torch.set_num_threads(nt)
        t0 = time.perf_counter()
        step()
        sweep[nt] = time.perf_counter() - t0
    best_nt = min(sweep, key=sweep.get)
    torch.set_num_threads(best_nt)
    n, secs, med = _timed_iters(step, budget_s=12.0, min_iters=3, max_iters=8, warm=False)
    torch.set_num_threads(1)
    # (B = 2 is the smallest batch the reference's training step accepts: the ASPP image-pooling BatchNorm sees B values per channel)
    n1, secs1, med1 = _timed_iters(make(2), budget_s=6.0, min_iters=1, max_iters=2, warm=False)
    torch.set_num_threads(usable)
    return {"value": round(sample_batch / med, 2), "unit": "frames/s", "cores": best_nt, "kind": "port",
            "cpu_model": cpu_model, "sockets": sockets, "physical_cores": cores, "container_cpu_quota": quota,
            "iterations": n, "warmup_iterations": 1,
            "thread_sweep_frames_per_s": {str(k): round(sample_batch / v, 2) for k, v in sorted(sweep.items())},
            "single_thread": {"value": round(2.0 / med1, 3), "unit": "frames/s", "cores": 1, "iterations": n1,
                              "sample": "B=2 frames (audio 4), same step, no warm-up"},
            "sample": f"median of {n} x (forward_train + CE + backward) of B={sample_batch} frames (audio {2 * sample_batch}) on {best_nt} "
                      f"threads (the best of a one-step sweep over {sorted(sweep)} threads, after one warm-up step), fp32, torch CPU "
                      f"autograd over the oracle, same model and synthetic inputs"}


def cpu_baseline(sd, cfg, sample_batch):
    """The oracle (kind 'port': our restatement of the reference, pinned by golden vectors) on the host cores."""
    from cavp_amd.synth import synth_inputs
    from oracle import cavp_oracle as O
    cores, cpu_model, sockets = _host_cpu()
    quota = _cpu_quota()
    cores = cores if quota is None else min(cores, quota)   # the threads the container can actually run
    image, audio, _ = synth_inputs(sample_batch, cfg["hw"], num_classes=cfg["C"], seed=0)
    with torch.no_grad():
        torch.set_num_threads(cores)
        n, secs, med = _timed_iters(lambda: O.cavp_forward(sd, image, audio, cfg["lds"], eval_mode=True), 12.0, 5, 20)
        torch.set_num_threads(1)
        n1, secs1, med1 = _timed_iters(lambda: O.cavp_forward(sd, image[:1], audio[:1], cfg["lds"], eval_mode=True), 6.0, 1, 5)
        torch.set_num_threads(cores)
    return {"value": round(sample_batch / med, 2), "unit": "frames/s", "cores": cores, "kind": "port",
            "cpu_model": cpu_model, "sockets": sockets, "iterations": n, "warmup_iterations": 1,
            "single_thread": {"value": round(1.0 / med1, 3), "unit": "frames/s", "cores": 1, "iterations": n1, "sample": "B=1 frame"},
            "sample": f"median of {n} x eval forward of B={sample_batch} frames after one warm-up, fp32, torch CPU ({cores} threads), "
                      f"same C1' model and synthetic inputs"}


def cpu_baseline_survey(sd, cfg, full: bool):
    """SURVEY.md section 8(d)'s CPU-baseline protocol beside the bounded sample above: median of >= 20 timed iterations after >= 5
    warm-ups, threads = the CPUs this container may use (stated), fp32 - eval forward at B = 2 (and B = 32 with `full`), train
    forward + backward at B = 2 - plus the iteration counts actually run.  The default bench line carries the legs that fit its time
    budget (eval B = 2 in full; train B = 2 with as many of the 20 iterations as 20 s allow, count stated); `--cpu-protocol-full`
    runs every leg to the letter (minutes) - its output is committed as profiles/r06_cpu_baseline_survey_protocol.json."""
    from cavp_amd.synth import synth_inputs
    from oracle import cavp_oracle as O
    cores, cpu_model, sockets = _host_cpu()
    quota = _cpu_quota()
    usable = cores if quota is None else min(cores, quota)
    torch.set_num_threads(usable)
    out = {"threads": usable, "physical_cores": cores, "container_cpu_quota": quota, "cpu_model": cpu_model, "sockets": sockets,
           "protocol": ">= 5 warm-ups, median of >= 20 timed iterations (SURVEY.md 8d); legs that stop early say so"}

    def leg(fn, frames, budget_s, warmups=5, iters=20):
        for _ in range(warmups):
            fn()
        ts, t_all = [], time.perf_counter()
        while len(ts) < iters and (full or len(ts) < 3 or time.perf_counter() - t_all < budget_s):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        return {"value": round(frames / ts[len(ts) // 2], 3), "unit": "frames/s", "warmup_iterations": warmups, "iterations": len(ts),
                "complete": len(ts) >= iters}

    for b in ((2, 32) if full else (2,)):
        image, audio, _ = synth_inputs(b, cfg["hw"], num_classes=cfg["C"], seed=0)
        with torch.no_grad():
            out[f"eval_forward_B{b}"] = leg(lambda: O.cavp_forward(sd, image, audio, cfg["lds"], eval_mode=True), b, 20.0)
    image, audio, label = synth_inputs(2, cfg["hw"], audio_batch=4, num_classes=cfg["C"], seed=0)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running_" not in k}
    sd2 = dict(sd)
    sd2.update(params)

    def step():
        for q in params.values():
            q.grad = None
        o, _, _ = O.cavp_forward(sd2, image, audio, cfg["lds"], eval_mode=False)
        O.ce_loss_train(o, label, 2).backward()
    out["train_fwd_bwd_B2"] = leg(step, 2, 20.0, warmups=5 if full else 2)
    torch.set_num_threads(1)
    image1, audio1, _ = synth_inputs(1, cfg["hw"], num_classes=cfg["C"], seed=0)
    with torch.no_grad():
        out["eval_forward_B1_single_thread"] = leg(lambda: O.cavp_forward(sd, image1, audio1, cfg["lds"], eval_mode=True), 1, 6.0,
                                                   warmups=5 if full else 1, iters=20 if full else 3)
    torch.set_num_threads(usable)
    return out


def _free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(a) -> int:
    """`python bench.py --gpus N` without a launcher (the reference spawns one process per GPU itself, main_vpo_mono.py:274-275
    `mp.spawn`; engine/engine.py:50-54): re-run this command line under torch.distributed.run with N ranks on 127.0.0.1 and
    pass the ranks' output through.  Returns the launcher's exit code."""
    import subprocess
    if os.environ.get("CAVP_BENCH_SHARE_GPU") != "1" and torch.cuda.is_available() and torch.cuda.device_count() < a.gpus:
        print(f"[bench] --gpus {a.gpus} but this node has {torch.cuda.device_count()} GPU(s)", file=sys.stderr)
        return 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(a))
    cap_host_threads()
    if a.lib:
        from cavp_amd import _lib as _cl9
        _cl9.LIB_PATH = os.path.abspath(a.lib)
    if a.no_token_fusion:
        import cavp_amd.train as _tr
        _tr._FUSE_TOKEN_PATH = False
    if a.no_side_stream:
        import cavp_amd.train as _tr
        _tr._SIDE_STREAM = False
    if a.no_group_wgrad:
        import cavp_amd.train as _tr
        _tr._GROUP_WGRAD = False
    if a.no_branch_stream:
        import cavp_amd.train as _tr
        _tr._BRANCH_STREAM = False
    if a.no_side_packs:
        import cavp_amd.train as _tr
        _tr._SIDE_PACKS = False
    if a.no_bn_bwd_fusion:
        import cavp_amd.train as _tr
        _tr._FUSE_BN_BWD = False
    if a.no_tail_split:
        from cavp_amd import _lib as _cl0
        _cl0.load().cavp_set_tail_split(0)
    if a.no_rank1_attn:
        import cavp_amd.train as _tr
        _tr._RANK1_ATTN = False
    if a.wgrad_big:
        from cavp_amd import _lib as _cl1b
        assert _cl1b.load().cavp_set_wgrad_big(*(int(v) for v in a.wgrad_big.split(":"))) == 0
    if a.grad_allreduce == "bf16":
        import cavp_amd.train as _tr
        _tr.set_grad_allreduce_dtype(torch.bfloat16)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} rank(s); they must agree")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    # CAVP_BENCH_SHARE_GPU=1 (plumbing check on a 1-GPU box only): every rank uses cuda:0 and the collectives run on gloo
    share = os.environ.get("CAVP_BENCH_SHARE_GPU") == "1"
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import torch.distributed as dist
    if world > 1:
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        if dist.get_world_size() != a.gpus:
            raise SystemExit(f"bench.py: process group of {dist.get_world_size()} rank(s) for --gpus {a.gpus}")
    if a.deterministic:
        from cavp_amd import _lib as _cl
        _cl.set_deterministic(True, dev)
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    cfg = model_cfg(a.config)
    B = a.batch
    from cavp_amd.synth import synth_inputs
    model, sd = build_model(cfg, B, dtype, dev)
    train = a.mode == "train"
    image, audio, label = synth_inputs(B, cfg["hw"], audio_batch=2 * B if train else B, num_classes=cfg["C"], seed=100 + rank)
    image, audio, label = image.to(dev), audio.to(dev), label.to(dev)
    if a.config == "c5":
        # config #5 (AVSBench-MS): 5-frame clips batched as B = 5 x clips (the reference loops the frames at B = 1,
        # trainer_cavp_avs_obj.py:317-330); loss = CE on out[:B] + out[B:]*0 + ContrastLoss(temperature 0.1, max_views 512) on the
        # fusion halves (trainer_cavp_vpo_mono.py:171-189), back-propagated through the model's autograd node
        if not train or B % 5:
            raise SystemExit("bench.py --config c5: training step only, --batch must be a multiple of 5 (frames per clip)")
        import torch.nn.functional as F
        from cavp_amd.contrast import ContrastLoss
        crit = ContrastLoss(temperature=0.1, ignore_idx=255, max_views=512)
        label_shuf = synth_inputs(B, cfg["hw"], audio_batch=2 * B, num_classes=cfg["C"], seed=900 + rank)[2].to(dev)
        a.trainer_graphed = not a.no_graph    # the model's part of the step on two hipGraph replays; the losses stay eager torch / host code
        a.no_graph = True
        model.train()
        if a.trainer_graphed:
            model.enable_graphed_autograd()

        def run_step():
            model.zero_grad(set_to_none=True)
            out, fus, _ = model(image, audio, None, False)
            loss = F.cross_entropy(out[:B] + out[B:] * 0.0, label, ignore_index=255) + crit(fus[:B], label, fus[B:], label_shuf)
            loss.backward()
            return loss
        run_step_local = run_step
    elif train and a.trainer_loop:
        import torch.nn.functional as F
        model.train()
        a.trainer_graphed = not a.no_graph
        if a.trainer_graphed:
            model.enable_graphed_autograd()
        a.no_graph = True          # (no capture_train_step below: the model replays its own graphs inside the autograd node)

        def run_step():
            model.zero_grad(set_to_none=True)
            out, _, _ = model(image, audio, None, False)
            loss = F.cross_entropy(out[:B] + out[B:] * 0.0, label, ignore_index=255)
            loss.backward()
            return loss
        run_step_local = run_step
    elif train:
        model.train()

        def run_step():
            return model.train_step(image, audio, label)

        def run_step_local():   # rank-local (no collectives): what rank 0 alone re-runs for the per-kernel timing
            return model.train_step(image, audio, label, all_reduce=False)
    else:
        def run_step():
            return model(image, audio, eval_mode=True)
        run_step_local = run_step

    with torch.set_grad_enabled(a.config == "c5" or a.trainer_loop):
        run_step()      # eager warm-up: packs weights, sizes the workspace
        torch.cuda.synchronize()
        if train and not a.no_graph:
            try:
                step = model.capture_train_step(image, audio, label, split=True if a.split_graph else None)   # ~1000 launches as hipGraph(s)
            except Exception as ex:  # noqa: BLE001 - a failed capture must not cost the measurement: fall back to eager launches
                print(f"[bench] hipGraph capture failed ({type(ex).__name__}: {ex}); timing eager launches instead", file=sys.stderr)
                torch.cuda.synchronize()
                a.no_graph = True
                step = run_step
        elif train:
            step = run_step
        elif a.no_graph:
            def step():
                return model(image, audio, eval_mode=True)
        else:
            graph = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                model(image, audio, eval_mode=True)
            torch.cuda.current_stream().wait_stream(side)
            from cavp_amd.train import _no_gc_during_capture
            with _no_gc_during_capture(), torch.cuda.graph(graph, capture_error_mode="thread_local"):
                outs = model(image, audio, eval_mode=True)

            def step():
                graph.replay()
                return outs

        for _ in range(a.warmup):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    collective = None
    if world > 1 and train and getattr(model, "_grad_arena", None) is not None:
        # the two gradient collectives of a step on their own (nothing else on the device), max over ranks: what the step would
        # pay if nothing overlapped - the timed region above hides the early piece behind the backbone's backward
        from cavp_amd.train import _allreduce_range, grad_allreduce_dtype
        arena = model._grad_arena
        n, split = arena.flat.numel(), arena.split
        keep = arena.flat.clone()
        ms = {}
        for name_, lo, hi in (("early", split, n), ("late", 0, split)):
            for _ in range(2):
                _allreduce_range(arena, lo, hi, False)
            torch.cuda.synchronize()
            dist.barrier()
            t_c = time.perf_counter()
            for _ in range(5):
                _allreduce_range(arena, lo, hi, False)
            torch.cuda.synchronize()
            tt = torch.tensor([(time.perf_counter() - t_c) / 5 * 1e3], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            ms[name_] = float(tt.item())
        arena.flat.copy_(keep)
        wire_b = 2 if grad_allreduce_dtype() == torch.bfloat16 else 4
        collective = {"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(), "wire_dtype": a.grad_allreduce,
                      "early_piece_mb": round((n - split) * wire_b / 1e6, 1), "late_piece_mb": round(split * wire_b / 1e6, 1),
                      "early_piece_ms": round(ms["early"], 3), "late_piece_ms": round(ms["late"], 3),
                      "note": "each piece timed alone after the run (max over ranks, incl. the bf16 casts when wire_dtype is bf16); "
                              "inside the step the early piece overlaps the backbone's backward"}

    if rank == 0:
        value = world * B * a.steps / elapsed
        c1name = ("C1 (config_vpo_ss plumbing at 224x224): CAVP ResNet-50 OS8 + VGGish" if a.config == "c1" else
                  f"C5 (config #5, AVSBench-MS clip shape: {B // 5} clips x 5 frames; loss = CE + ContrastLoss(T 0.1, 512 views) on the "
                  f"fusion halves through the model's autograd node): CAVP ResNet-50 OS16 + VGGish" if a.config == "c5" else
                  "C1' (config_avss_binary shape): CAVP ResNet-50 OS16 + VGGish")
        line = {
            "metric": (f"frames/sec end-to-end CAVP fwd+bwd, B={B} {cfg['hw'][0]}x{cfg['hw'][1]}" if train else
                       f"frames/sec end-to-end CAVP forward (inference), B={B} {cfg['hw'][0]}x{cfg['hw'][1]}"),
            "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": (f"C4 (config_avss shape): CAVP PVTv2-B5 (DropPath 0.1) + VGGish, training step = forward_train "
                                    f"({B} images + {2 * B} audio clips per GPU) + cross-entropy + full backward, 512x512 RGB + "
                                    f"96x64 mel, num_classes=71, random-init (synthetic) weights" if (train and a.config == "c4") else
                                    f"{c1name}, training step = "
                                    f"forward_train (batch-stat BN, {B} images + {2 * B} audio clips per GPU) + cross-entropy + "
                                    f"full backward" + (" + gradient all-reduce over RCCL in two pieces, the first overlapped with the backbone backward" if world > 1 else "") +
                                    f", 224x224 RGB + 96x64 mel, num_classes={cfg['C']}, random-init (synthetic) weights"
                                    if train else
                                    (f"C4 (config_avss shape): CAVP PVTv2-B5 + VGGish, eval forward, B={B}/GPU, 512x512 RGB + 96x64 "
                                     f"mel, num_classes=71, random-init (synthetic) weights" if a.config == "c4" else
                                     f"{c1name}, eval forward, "
                                     f"B={B}/GPU, 224x224 RGB + 96x64 mel, num_classes={cfg['C']}, random-init (synthetic) weights")),
                       "global_batch": world * B, "per_gpu_batch": B, "parallelism": f"dp{world}",
                       "ranks": dist.get_world_size() if world > 1 else 1,
                       "collective_backend": (dist.get_backend() if world > 1 else None),
                       "launch": ("trainer loop: autograd node over two hipGraph replays (enable_graphed_autograd) + torch loss" if getattr(a, "trainer_graphed", False)
                                  else "trainer loop: eager autograd node + torch loss" if (a.trainer_loop or a.config == "c5")
                                  else "eager" if a.no_graph else "hipGraph replay"), "deterministic": bool(a.deterministic)},
        }
        if collective is not None:
            line["collective"] = collective
        if not a.no_roofline and getattr(a, "trainer_graphed", False):
            print("[bench] --trainer-loop on graphs: no per-launch roofline (the launches are inside the replays); use --no-graph for it",
                  file=sys.stderr)
            a.no_roofline = True
        if not a.no_roofline:
            roof = measure_roofline(model, run_step_local, image, a.dtype, config=a.config, live_pmc=a.pmc,
                                    trainer_loop=a.trainer_loop)
            step_gflop, meas_bytes = roof.pop("_step_gflop"), roof.pop("_measured_step_bytes")
            ms_step = elapsed / a.steps * 1e3
            if a.config in ("c1p", "c1", "c4", "c5") and a.dtype == "bf16":
                # the WHOLE step against both roofs (north_star: "fraction of the conv-bound HBM roofline"): algorithmic FLOPs of
                # every conv / linear launch incl. weight gradients, and the perfectly-fused byte minimum
                if a.config == "c4":
                    min_gb = fused_min_mb_per_frame_c4("train" if train else "eval", B) * B / 1e3
                else:
                    min_gb = (FUSED_MIN_MB_PER_FRAME_BF16 if a.config in ("c1p", "c5") else FUSED_MIN_MB_PER_FRAME_BF16_C1)["train" if train else "eval"] * B / 1e3
                roof["step"] = {
                    "bound": "hbm", "ms_per_step": round(ms_step, 3),
                    "algorithmic_gflop": round(step_gflop, 1), "fused_min_gb": round(min_gb, 2),
                    "achieved_tflops": round(step_gflop / ms_step, 2), "frac_of_mfma_peak": round(step_gflop / ms_step / MFMA_PEAK_TFLOPS[a.dtype], 4),
                    "achieved_gbs": round(min_gb / ms_step * 1e3, 1), "frac_of_hbm_peak": round(min_gb / ms_step * 1e3 / HBM_PEAK_GBS, 4),
                    "measured_hbm_gb": round(meas_bytes / 1e9, 2) if meas_bytes else None,
                    "measured_vs_fused_min": round(meas_bytes / 1e9 / min_gb, 2) if meas_bytes else None,
                }
            line["roofline"] = roof
        if getattr(a, "trainer_graphed", False) and a.dtype == "bf16":
            # (--trainer-loop / c5 on graph replays: no per-launch view, but the whole step against both roofs: the algorithmic
            # FLOPs of the same model's fused eager step, counted once through the launch wrappers, and the fused byte minimum)
            try:
                line["roofline"] = {"step": graphed_step_roofline(model, image, audio, label, elapsed / a.steps * 1e3, a.config, B, a.dtype)}
            except Exception as ex:  # noqa: BLE001
                print(f"[bench] step roofline of the graphed trainer loop failed: {type(ex).__name__}: {ex}", file=sys.stderr)
        if train and a.dtype == "bf16" and a.config == "c1p" and world == 1 and not a.no_eval_leg:
            # north_star states its target on the inference forward (">= 400 frames/s at >= 60 % of the conv-bound HBM roofline"):
            # the same model, eval forward on one hipGraph, beside the training headline
            try:
                line["eval_forward"] = eval_forward_leg(model, image, audio[:B], B, a.dtype)
            except Exception as ex:  # noqa: BLE001
                print(f"[bench] eval_forward leg failed: {type(ex).__name__}: {ex}", file=sys.stderr)
        if train and a.dtype == "bf16" and a.config in ("c1p", "c1") and world == 1 and not a.no_f32:
            # the same step on the f32 parity path (every kernel within 1e-3 of the reference, tests/): a throughput at the
            # north-star tolerance next to the bf16 headline
            del step
            model32, _ = build_model(cfg, B, torch.float32, dev)
            model32.train()
            with torch.no_grad():
                model32.train_step(image, audio, label)
                st32 = model32.capture_train_step(image, audio, label)
                for _ in range(2):
                    st32()
                torch.cuda.synchronize()
                t32 = time.perf_counter()
                for _ in range(5):
                    st32()
                torch.cuda.synchronize()
                ms32 = (time.perf_counter() - t32) / 5 * 1e3
            line["f32_parity_path"] = {"value": round(B / ms32 * 1e3, 2), "unit": "frames/s", "ms_per_step": round(ms32, 3), "steps": 5,
                                       "dtype": "f32", "note": "same training step, exact-f32 MFMA kernels (logits within 1e-3 of the reference)"}
        if world == 1 and not a.no_cpu_baseline:
            if a.config in ("c1p", "c1"):
                line["cpu_baseline"] = (cpu_baseline_train(sd, cfg, max(2, a.cpu_sample_batch)) if train
                                        else cpu_baseline(sd, cfg, a.cpu_sample_batch))
                line["cpu_baseline"]["survey_8d_protocol"] = cpu_baseline_survey(sd, cfg, a.cpu_protocol_full)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
