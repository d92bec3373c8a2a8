#!/usr/bin/env python3
"""Headline benchmark: images/sec of the multi-label forward (image -> [B,80] logits), ViT-B/16 @ 224, B=256 per GPU.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one synthetic batch already resident in HBM: patch-embed, 12 residual
attention blocks, ln_post + projection, L2-normalise + x4.0 cosine logits against the 80 cached class-prompt text
features (CustomCLIP.forward(if_test=True): the tail kernel does ln_post + projection + normalise + logits), then - for
N > 1 - the RCCL all-gather of the per-rank [256,80] logits.  Weak scaling: 256 images per GPU (global 2048 at N=8 =
BASELINE configs[3]).  Rank 0 prints ONE JSON line.  Default dtype fp16 (the reference's GPU precision and the north star's
target dtype: it meets the +-0.2 mAP clause); the bf16 rate and mAP of the same kernels are reported under "bf16".

roofline: the dominant kernel family is the MFMA GEMM (96 % of algorithmic FLOPs).  ``achieved`` = algorithmic
FLOPs of the GEMM launches of one step / their summed duration, from HIP events recorded around every GEMM launch
on the launch stream inside the timed region - on its sampled steps (--profile-every), which run the batch as ONE part: the
other steps run it as two halves on two HIP streams whose kernels overlap (hip/engine.py), so an event
pair around a launch would include the time it shares the chip; ``peak`` = 2516.6 TFLOP/s dense bf16/fp16 MFMA (256 CU x
2.4 GHz x 4096 FLOP/clk/CU).  cpu_baseline: the CPU oracle (oracle/clip_oracle.py, torch-CPU fp32 restatement pinned to the
reference) timed on the host cores on a bounded sample, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import leclip_amd  # noqa: E402
# The image engine runs the two halves of a batch on two HIP streams, which only overlap when they sit on different hardware queues.  The
# runtime has 4 by default and shares them among every stream in use (null stream, the two part streams, RCCL's): room for the rest.
# Must happen before the HIP runtime starts; measured equal to the default on one GPU (profiles/r02_ab_stream_parts.txt).
leclip_amd.configure()

PEAK_MFMA_TFLOPS = 2516.6
PEAK_HBM_GBS = 8000.0


def flops_per_image(arch):
    p, d, e, L = arch.grid ** 2, arch.vision_width, arch.embed_dim, arch.vision_layers
    t = p + 1
    return 2 * p * 3 * arch.vision_patch_size ** 2 * d + L * (24 * t * d * d + 4 * t * t * d) + 2 * d * e


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU")
    ap.add_argument("--dtype", default="fp16", choices=["bf16", "fp16", "fp32"],
                    help="compute dtype of the timed run.  Default fp16: the reference's own GPU precision (clip/model.py:470 "
                         "convert_weights) and the dtype the north-star target names; it meets the +-0.2 mAP clause.  bf16 (BASELINE "
                         "configs[1]'s wording) runs the same kernels at the same rate but misses that clause (8-bit mantissa): with the "
                         "default dtype the bf16 rate and mAP are measured too and reported under \"bf16\" in the same JSON line")
    ap.add_argument("--no-second-dtype", action="store_true", help="skip the bf16 companion measurement of the default run")
    ap.add_argument("--arch", default="ViT-B/16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured HIP graph on the steps that are not profiled")
    ap.add_argument("--profile-every", type=int, default=20, help="record per-kernel HIP events on every N-th timed step (0 = never)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--streams", type=int, default=None, help="parts the image engine runs a large batch as, each on a HIP stream of its own "
                    "(kernel tails of one part overlap the other's kernels; default: the engine's own, 2)")
    ap.add_argument("--split", default=None, help="experiments: explicit part sizes, e.g. 144,112")
    ap.add_argument("--last-block", default="full", choices=["full", "class-token"],
                    help="full (default): the last residual block is computed for every token, as the reference does and as the algorithmic "
                         "FLOP count of the metric assumes.  class-token: the engine's own default outside this benchmark - only what "
                         "ln_post(x[:, 0]) consumes (same logits bit for bit, 6 %% fewer FLOPs executed); its rate is reported beside the headline")
    ap.add_argument("--walk", default=None, choices=["default", "c_proj", "c_fc", "alternate"],
                    help="A/B: row walk-order policy of the block kernels (hip/engine.py _Walk; default: the engine's own = library order for every launch)")
    ap.add_argument("--no-companions", action="store_true", help="skip the cfg5 (ViT-L/14@336) and cfg3 (tuning step) companion child runs of the default run")
    ap.add_argument("--raw-batch", type=int, default=8, help="--mode multicrop: raw 640x480 images per step (305 windows each)")
    ap.add_argument("--no-producer-merge", action="store_true", help="A/B: LayerNorm partials merged by a launch behind out-proj / c_proj instead of "
                    "inside them (hip/engine.py producer_merge)")
    ap.add_argument("--gemm-family", type=int, default=None, choices=[128, 256, 384], help="A/B: force a GEMM kernel family (leclip_set_gemm_family)")
    ap.add_argument("--image-dtype", default="compute", choices=["compute", "fp32"],
                    help="dtype the synthetic images are resident in when the timed region starts: the tower's compute dtype (default; SURVEY 8d: "
                         "'cast to bf16/fp16 for cfgs 2-5') or fp32 (the engine's patch-extraction kernel then casts inside the step)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="--mode tune A/B: every step computes its own image features first (no lookahead: Caption_distill_double.pipeline_image_tower = False)")
    ap.add_argument("--no-text-beside", action="store_true",
                    help="--mode tune A/B: the text tower's forward behind the image tower instead of beside its stream parts (CustomCLIP.text_beside_image)")
    ap.add_argument("--mode", default="score", choices=["score", "tune", "multicrop"],
                    help="score: the headline inference step (default).  tune: BASELINE configs[2], one prompt-tuning step = frozen "
                         "image tower on the batch + text tower forward/backward w.r.t. the 16 context vectors + BCE + SGD")
    ap.add_argument("--tune-model", default="CustomCLIP", choices=["CustomCLIP", "DenseCLIP"],
                    help="--mode tune: CustomCLIP = image batches through the frozen image tower, BCE (BASELINE configs[2] as worded); DenseCLIP = the "
                         "step the reference's shipped configs run: tokenised captions as images, global + local head, three prompt sets, "
                         "double_ranking + EMA distillation loss (value = captions/s)")
    ap.add_argument("--dry-launch", action="store_true", help="with --gpus N > 1 and no torchrun environment: print the launcher "
                    "command this invocation would start (one JSON line) and exit without starting it")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        return self_launch(args)
    if args.mode == "tune":
        return tune(args)
    if args.mode == "multicrop":
        return multicrop(args)

    import torch
    import torch.distributed as dist
    from leclip_amd import parallel, synth
    from leclip_amd.clip import build_model, convert_weights
    from leclip_amd.config import get_cfg_default
    from leclip_amd.datasets import coco_object_categories
    from leclip_amd.hip import ops
    from leclip_amd.trainers import CustomCLIP

    overrides = env_overrides()
    if args.gemm_family is not None:
        ops.set_gemm_family(args.gemm_family)
        overrides = list(overrides) + [f"gemm_family={args.gemm_family}"]
    rank, world, local = parallel.init_from_env()
    if world != args.gpus:   # only reachable under a launcher whose rank count differs from --gpus (a bare invocation self-launches)
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher's WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    dev = torch.device("cuda", torch.cuda.current_device())
    arch = synth.ARCHS[args.arch]
    DT = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}
    sd = synth.make_state_dict(arch, seed=0, dist="cond")
    ctx = torch.from_numpy(synth.make_ctx(16, arch.transformer_width, seed=0))
    B = args.batch
    images = torch.from_numpy(synth.make_images(B, arch.image_resolution, seed=1234, start=rank * B)).to(dev)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def build(dtype_name):
        model = build_model(sd).float()
        if DT[dtype_name] != torch.float32:
            convert_weights(model, DT[dtype_name])
        cfg = get_cfg_default()
        cfg.INPUT.SIZE = (arch.image_resolution, arch.image_resolution)
        cc = CustomCLIP(cfg, coco_object_categories, model)
        with torch.no_grad():
            cc.prompt_learner.ctx.copy_(ctx)
        cc = cc.to(dev).eval()
        eng = cc.image_encoder.engine(dev)
        if args.streams is not None:
            eng.streams = args.streams
        if args.split:
            eng.split_sizes = [int(v) for v in args.split.split(",")]
        eng.cls_last_block = args.last_block == "class-token"
        if args.walk is not None:
            eng.walk = args.walk
        eng.producer_merge = not args.no_producer_merge
        return cc

    def measure(cc, steps, warmup, profile_every, image_dtype=None):
        """W untimed + exactly K timed steps of the hot path, bracketed by barrier + synchronize; max over ranks."""
        image_dtype = image_dtype or args.image_dtype
        eng = cc.image_encoder.engine(dev)
        run_streams, run_split = eng.streams, eng.split_sizes
        scorer = parallel.ShardedScorer(lambda x: cc(x, if_test=True)[0])
        # the synthetic images are "cast to bf16/fp16 for cfgs 2-5" - input preparation, done once here, outside the timed
        # region (inputs resident in HBM in the dtype the tower computes in; the reference's own encode_image casts with image.type(self.dtype),
        # clip/model.py:377: same rounding).  The engine takes fp32 images as well (a patch-extraction kernel then does the cast).
        images_dt = images.to(eng.dtype) if image_dtype == "compute" else images
        step = lambda: scorer.score_local(images_dt)
        with torch.no_grad():
            cc.class_text_features()          # text tower runs once; its features are cached for inference
            if profile_every > 0:   # the sampled steps run the batch as one part: allocate that shape's workspace before the timed region too
                eng.streams, eng.split_sizes = 1, None
                out = step()
                eng.streams, eng.split_sizes = run_streams, run_split
            for _ in range(warmup):
                out = step()
            fence()
            graph = None
            if args.graph and world == 1:
                # the launches of one step captured once (the C-ABI launches go to torch's current stream, which is the
                # capturing stream here); replayed with one hipGraphLaunch per step
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    gout = step()
                fence()
            prof = []
            sampled_ev = []
            sample_at = min(profile_every // 2, steps // 2) if profile_every > 0 else 0   # mid-run: clocks and caches in their steady state
            t0 = time.perf_counter()
            for i in range(steps):
                # per-kernel HIP events on every `profile_every`-th step of the timed region (a pair of event records around
                # each launch costs a few microseconds of GPU idle; sampling keeps the measured rate honest)
                sampled = profile_every > 0 and i % profile_every == sample_at
                ops.set_profile(prof if sampled else None)
                # a sampled step runs the batch as ONE part, so that an event pair brackets a kernel that has the chip to itself
                # (with two parts in flight a launch's elapsed time includes the CUs it shares with the other stream's kernel)
                eng.streams, eng.split_sizes = (1, None) if sampled else (run_streams, run_split)
                if graph is not None and not sampled:
                    graph.replay()
                    out = gout
                elif sampled:
                    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ev[0].record()
                    out = step()
                    ev[1].record()
                    sampled_ev.append(ev)
                else:
                    out = step()
            ops.set_profile(None)
            eng.streams, eng.split_sizes = run_streams, run_split
            fence()
            dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        assert out.shape == (world * B, 80) and bool(torch.isfinite(out).all())
        info = {"sampled_step_ms": sum(a.elapsed_time(b) for a, b in sampled_ev) / len(sampled_ev) if sampled_ev else None,
                "parts": len(eng._parts(images) or [None]),
                "logits": out.float().cpu().numpy() if rank == 0 else None}      # the last timed step's gathered [world * B, 80] logits
        return dt, prof, info

    cc = build(args.dtype)
    dt, prof, run_info = measure(cc, args.steps, args.warmup, args.profile_every)

    # per-kernel-family durations from the HIP events recorded inside the timed region
    fam = {}
    shapes = {}
    for name, fl, nb, e0, e1, shape in prof:
        sec = e0.elapsed_time(e1) * 1e-3
        for key, table in ((name, fam), (shape if name == "gemm" and shape else None, shapes)):
            if key is None:
                continue
            f = table.setdefault(key, [0.0, 0, 0, 0])
            f[0] += sec
            f[1] += fl
            f[2] += nb
            f[3] += 1
    g = fam.get("gemm", [1e-9, 0, 0, 1])
    gemm_tflops = g[1] / g[0] * 1e-12
    off = min(args.profile_every // 2, args.steps // 2) if args.profile_every > 0 else 0
    prof_steps = len([i for i in range(args.steps) if i % args.profile_every == off]) if args.profile_every > 0 else 1
    kernels = {k: {"launches_per_step": v[3] // prof_steps, "avg_us": v[0] / v[3] * 1e6,
                   "tflops": v[1] / v[0] * 1e-12, "alg_gbs": v[2] / v[0] * 1e-9} for k, v in fam.items()}

    ips = world * B * args.steps / dt
    fpi = flops_per_image(arch)
    traffic, traffic_src = _pmc_traffic()
    result = {
        "metric": f"images/sec (multi-label forward, B={B}, {arch.image_resolution}x{arch.image_resolution})", "value": ips, "unit": "img/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{args.arch} image tower + 80 learnable-prompt class features (cached) -> x4.0 cosine logits, "
                               f"B={B}/GPU {args.dtype}, inference-only (BASELINE configs[1]; configs[3] at 8 GPUs)",
                   "global_batch": world * B, "parallelism": f"dp{world}" + ("+allgather(logits)" if world > 1 else ""),
                   "world_size": dist.get_world_size() if dist.is_initialized() else 1,
                   "backend": (dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else "")) if dist.is_initialized() else None,
                   "flops_per_image": fpi, "stream_parts": run_info["parts"],
                   "image_dtype": args.dtype if args.image_dtype == "compute" else "fp32"},
        "end_to_end_tflops_per_gpu": ips / world * fpi * 1e-12,
        "end_to_end_mfma_frac": ips / world * fpi * 1e-12 / PEAK_MFMA_TFLOPS,
        "roofline": {"bound": "mfma", "kernel": ops._capi.load().leclip_gemm_kernel_name(B * arch.vision_tokens, arch.vision_width,
                                                                                         arch.vision_width, ops.dtype_code(DT[args.dtype])).decode(),
                     "achieved": gemm_tflops, "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": gemm_tflops / PEAK_MFMA_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                     "flops_per_launch": g[1] / max(g[3], 1), "avg_launch_us": g[0] / max(g[3], 1) * 1e6,
                     "measured_on": "the sampled steps of the timed region (1 in %d), which run the batch as ONE part so that each launch has the "
                                    "chip to itself; the other steps run it as %d parts on HIP streams of their own, whose kernels overlap - "
                                    "that is why ms_per_step is below the sum of the kernel times" % (max(args.profile_every, 1), run_info["parts"]),
                     "sampled_step_ms": run_info["sampled_step_ms"]},   # one part + an event pair around each of its launches
        "kernels": kernels,
        # the GEMM family per launch shape (same HIP events, sampled one-part steps): which of the four block GEMMs trails
        "gemm_shapes": {k: {"launches_per_step": v[3] // prof_steps, "avg_us": v[0] / v[3] * 1e6, "tflops": v[1] / v[0] * 1e-12,
                            "mfma_frac": v[1] / v[0] * 1e-12 / PEAK_MFMA_TFLOPS} for k, v in sorted(shapes.items(), key=lambda kv: -kv[1][0])},
        "env_overrides": overrides,
    }

    # Accuracy evidence at every N (BASELINE configs[3]: "eval mAP on synthetic 80-label set" on the all-gathered logits): rank 0 scores the
    # gathered [world * 256, 80] logits of the last timed step against the committed reference fixture's rows - the reference's own model.py
    # on these very images - with the evaluator's mAP; no CPU oracle in the loop, so the N = 8 line carries it too.
    fixture = reference_fixture(args.arch, B, world) if args.dtype in LABEL_BAND else None
    if rank == 0 and fixture is not None:
        result["mAP_gathered"] = map_on_gathered_logits(run_info["logits"], fixture, args.dtype)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"], result["mAP"], ref_pack = cpu_baseline(args, arch, sd, cc, ctx, dev)
        result["mAP"]["accuracy_gate"] = accuracy_gate(result["mAP"], args.dtype)
    else:
        ref_pack = None
    # Companion measurement: the engine's own default computes, in the LAST residual block, only what the class token needs (identical
    # logits, tests/test_gpu_parity.py::test_full_batch_properties; 6 % fewer FLOPs executed).  The headline above keeps the whole block,
    # like the reference and like the FLOP count its fractions use; this is the rate a user of the package gets.
    if world == 1 and args.last_block == "full" and not args.no_second_dtype:
        eng = cc.image_encoder.engine(dev)
        eng.cls_last_block = True
        dt3, _, _ = measure(cc, args.steps, max(2, args.warmup // 2), 0)
        eng.cls_last_block = False
        t, d = arch.vision_tokens, arch.vision_width
        skipped = (t - 1) * (20 * d * d + 4 * t * d)   # the other rows' q projection (2 d^2), out-proj (2 d^2), MLP (16 d^2) and attention (4 T d)
        result["class_token_last_block"] = {"value": B * args.steps / dt3, "unit": "img/s", "ms_per_step": dt3 / args.steps * 1e3, "steps": args.steps,
                                            "flops_per_image_executed": fpi - skipped,
                                            "note": "engine default outside this benchmark; logits bit-identical to the headline run's"}
    # Companion measurement (VERDICT r4 task 4a): the hand-over the reference actually has - fp32 images from parse_batch_test, cast inside
    # encode_image (clip/model.py:376-377).  Same engine; an fp32 batch goes through the patch-extraction kernel (which does the cast) instead of the
    # im2col-free gather of the 16-bit path.  One sampled step gives the patch path's own time.
    if world == 1 and args.image_dtype == "compute" and args.dtype != "fp32" and not args.no_second_dtype:
        dt4, prof4, _ = measure(cc, args.steps, max(2, args.warmup // 2), args.steps, image_dtype="fp32")
        pe = [e0.elapsed_time(e1) * 1e3 for name, _, _, e0, e1, _ in prof4 if name == "patch_embed"]
        result["fp32_images"] = {"value": B * args.steps / dt4, "unit": "img/s", "ms_per_step": dt4 / args.steps * 1e3, "steps": args.steps,
                                 "patch_embed_us": sum(pe) / max(len(pe), 1), "patch_embed_us_16bit_images": kernels.get("patch_embed", {}).get("avg_us"),
                                 "note": "fp32 NCHW images resident in HBM, cast inside the timed region by the patch-extraction kernel"}
    # Companion measurement: BASELINE configs[1] words the config as bf16.  Same kernels, same rate, but bf16 misses the
    # north star's +-0.2 mAP clause, so it is not the headline: its rate and mAP sit beside the fp16 line.
    if world == 1 and args.dtype == "fp16" and not args.no_second_dtype:
        del cc
        cc2 = build("bf16")
        dt2, _, info2 = measure(cc2, args.steps, max(2, args.warmup // 2), 0)
        comp = {"value": B * args.steps / dt2, "unit": "img/s", "ms_per_step": dt2 / args.steps * 1e3, "steps": args.steps}
        if fixture is not None:
            comp["mAP_gathered"] = map_on_gathered_logits(info2["logits"], fixture, "bf16")
        if ref_pack is not None:
            comp["mAP"] = score_against(ref_pack, cc2, arch, dev, "bf16")
            comp["mAP"]["accuracy_gate"] = accuracy_gate(comp["mAP"], "bf16")
        result["bf16"] = comp
        del cc2
    # Companion measurements (VERDICT r4 task 6): BASELINE configs[4] (ViT-L/14@336 fp16, one GPU's share B = 128) and configs[2] (prompt-tuning step,
    # B = 512) as CHILD processes of this run, after the headline's timed region - each prints its own line (with its own roofline block), of which
    # the key figures are kept here so that the driver's record carries them.
    if world == 1 and not args.no_second_dtype and not args.no_companions and args.arch == "ViT-B/16" and args.batch == 256:
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        result["cfg5_vitl"] = companion_run(["--arch", "ViT-L/14@336px", "--batch", "128", "--steps", "10", "--warmup", "3", "--no-cpu-baseline",
                                             "--no-second-dtype"], ("value", "unit", "ms_per_step", "steps", "dtype", "end_to_end_mfma_frac", "roofline", "gemm_shapes", "kernels"))
        result["cfg3_tune"] = companion_run(["--mode", "tune", "--dtype", "bf16", "--steps", "10", "--warmup", "3"],
                                            ("value", "unit", "ms_per_step", "steps", "dtype", "roofline", "config"))
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def companion_run(flags, keep):
    """One more configuration of this script as a child process (never an exec: the parent has initialised the GPU); its JSON line, cut to ``keep``."""
    import subprocess
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__)] + flags, capture_output=True, text=True, timeout=420)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
        return {k: d[k] for k in keep if k in d}
    except Exception as exc:      # the headline line must not die of a companion
        return {"error": repr(exc)[:300]}


def launcher_argv(n_gpus, bench_args, port):
    """The command a bare ``python bench.py --gpus N ...`` turns into: one rank per GPU under torch.distributed.run on
    127.0.0.1 (the container's hostname may not resolve), the same flags passed through."""
    passed = [a for a in bench_args if a != "--dry-launch"]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + passed


def self_launch(args):
    """``--gpus N`` (N > 1) without a torchrun environment: start the N ranks as a CHILD process and exit with its code.  Nothing
    here imports torch or touches a device - replacing a process that has initialised the GPU (exec) is forbidden on this pool, and a
    child keeps the parent's stdout, so rank 0's JSON line is this invocation's JSON line."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    argv = launcher_argv(args.gpus, sys.argv[1:], port)
    if args.dry_launch:
        print(json.dumps({"launch": argv}))
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL needs it across processes)
    env.setdefault("OMP_NUM_THREADS", "4")
    sys.exit(subprocess.run(argv, env=env).returncode)


def env_overrides():
    """LECLIP_* variables in the environment.  The product library reads none of them (the timing / ablation hooks of round 1
    now exist only in the diagnostic build, `make diag`); a run with LECLIP_GEMM_* / LECLIP_ATTN_* set is refused anyway so
    that a profiling script's exports can never sit under a headline number.  LECLIP_HIP_LIB (library path) is recorded."""
    found = sorted(k for k in os.environ if k.startswith("LECLIP_"))
    bad = [k for k in found if k.startswith(("LECLIP_GEMM_", "LECLIP_ATTN_", "LECLIP_LN_"))]
    if bad:
        print(f"bench.py: refusing to run with kernel diagnostic switches in the environment: {bad}", file=sys.stderr)
        sys.exit(2)
    return [f"{k}={os.environ[k]}" for k in found]


def multicrop(args):
    """Secondary measurement (VERDICT r4 task 5): the reference's REAL inference path - SURVEY row N2; trainers/Caption_distill_double.py:637-676 with
    dassl/data/data_manager.py:311-492.  A step = B raw uint8 640 x 480 images -> the full image + the 304 sliding windows of scales (2, 3, 4) (the
    shipped run_eval.sh) through the Pillow-exact device resampler (crop_resize_kernel: bicubic Resize, CenterCrop, ToTensor, Normalize) ->
    the hot path (image tower + cosine logits) in chunks of --batch crops -> window aggregation (max / min, threshold 0.3) -> 1.4 s_ag + global.
    Reports images/s and crops/s, and a roofline block for crop_resize_kernel: algorithmic bytes = the windows' source footprints + 301 KB of
    16-bit output per crop, against the HBM peak (the kernel computes its coefficient tables in fp64 per output tile: how far below the byte
    roof it runs says how much of it is arithmetic)."""
    import numpy as np
    import torch
    from leclip_amd import synth
    from leclip_amd.clip import build_model, convert_weights
    from leclip_amd.config import get_cfg_default
    from leclip_amd.datasets import coco_object_categories
    from leclip_amd.hip import ops
    from leclip_amd.multicrop import MultiCropper
    from leclip_amd.trainers import CustomCLIP

    overrides = env_overrides()
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    dev = torch.device("cuda", torch.cuda.current_device())
    arch = synth.ARCHS[args.arch]
    DT = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}
    sd = synth.make_state_dict(arch, seed=0, dist="cond")
    model = build_model(sd).float()
    if DT[args.dtype] != torch.float32:
        convert_weights(model, DT[args.dtype])
    cfg = get_cfg_default()
    cfg.INPUT.SIZE = (arch.image_resolution, arch.image_resolution)
    cc = CustomCLIP(cfg, coco_object_categories, model)
    with torch.no_grad():
        cc.prompt_learner.ctx.copy_(torch.from_numpy(synth.make_ctx(16, arch.transformer_width, seed=0)))
    cc = cc.to(dev).eval()
    eng = cc.image_encoder.engine(dev)
    H, W, scales = 480, 640, (2, 3, 4)
    Braw, chunk = args.raw_batch, args.batch
    rng = np.random.default_rng(1234)
    raw = torch.from_numpy(rng.integers(0, 256, size=(Braw, 3, H, W), dtype=np.uint8)).to(dev)
    cropper = MultiCropper(arch.image_resolution, scales, dtype=eng.dtype)
    win, counts = cropper.windows(H, W, dev)
    nw = int(win.shape[0])                                    # 1 + 304
    wnp = win.cpu().numpy()
    src_bytes = int((wnp[:, 2].astype(np.int64) * wnp[:, 3]).sum()) * 3
    out_bytes = nw * 3 * arch.image_resolution ** 2 * (2 if eng.dtype != torch.float32 else 4)
    ev = []

    def step(timed):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if timed else None
        if timed:
            e[0].record()
        crops = ops.crop_resize(raw, win, arch.image_resolution, cropper.mean, cropper.std, eng.dtype)     # [B, 305, 3, S, S]
        if timed:
            e[1].record()
        flat = crops.view(Braw * nw, 3, arch.image_resolution, arch.image_resolution)
        logits = torch.cat([cc(flat[i:i + chunk], if_test=True)[0].float() for i in range(0, Braw * nw, chunk)]).view(Braw, nw, -1)
        if timed:
            e[2].record()
        out = ops.window_aggregate(logits[:, 0].contiguous(), logits[:, 1:].contiguous(), threshold=0.3, weight=1.4)
        if timed:
            e[3].record()
            ev.append(e)
        return out

    with torch.no_grad():
        cc.class_text_features()
        for _ in range(args.warmup):
            out = step(False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            out = step(i % 5 == 2)           # events on every fifth step
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    assert out.shape == (Braw, 80) and bool(torch.isfinite(out).all())
    crop_us = sum(a[0].elapsed_time(a[1]) for a in ev) / len(ev) * 1e3
    tower_us = sum(a[1].elapsed_time(a[2]) for a in ev) / len(ev) * 1e3
    agg_us = sum(a[2].elapsed_time(a[3]) for a in ev) / len(ev) * 1e3
    alg = Braw * (src_bytes + out_bytes)
    crops_s = Braw * nw * args.steps / dt
    result = {
        "metric": f"images/sec (multi-crop inference: {nw} windows per {W}x{H} image, scales {scales})", "value": Braw * args.steps / dt, "unit": "img/s",
        "crops_per_s": crops_s, "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic (uniform random uint8 pixels)",
        "config": {"workload": f"{args.arch}: {Braw} raw {W}x{H} uint8 images per step -> {nw} crops each (crop_resize_kernel) -> image tower + cosine logits in "
                               f"chunks of {chunk} -> window_aggregate (SURVEY N2; reference CDD.py:637-676)", "crops_per_step": Braw * nw, "windows_per_scale": counts},
        "roofline": {"bound": "hbm", "kernel": "crop_resize_kernel", "achieved": alg / crop_us * 1e-3, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                     "frac": alg / crop_us * 1e-3 / PEAK_HBM_GBS, "traffic": None, "algorithmic_bytes_per_launch": alg, "avg_launch_us": crop_us,
                     "source_footprint_bytes_per_image": src_bytes, "output_bytes_per_image": out_bytes,
                     "note": "per output tile the kernel rebuilds Pillow's bicubic coefficient tables in IEEE double: below the byte roof = arithmetic"},
        "kernels": {"crop_resize": {"avg_us": crop_us, "crops_per_s_alone": Braw * nw / crop_us * 1e6}, "tower_and_logits": {"avg_us": tower_us, "crops_per_s_alone": Braw * nw / tower_us * 1e6},
                    "window_aggregate": {"avg_us": agg_us}},
        "resampler_keeps_up": Braw * nw / crop_us * 1e6 >= 0.9 * (Braw * nw / tower_us * 1e6),
        "env_overrides": overrides,
    }
    print(json.dumps(result))
    return 0


def tune(args):
    """Secondary measurement (not the headline metric): prompt-tuning steps on image batches, BASELINE configs[2].  Under
    WORLD_SIZE > 1 each rank tunes on its own 512 images and the context gradients are averaged with one flat RCCL
    all-reduce per step (trainer._allreduce_grads): weak scaling, value = images of all ranks per second."""
    import torch
    import torch.distributed as dist
    from leclip_amd import parallel, synth
    from leclip_amd.config import get_cfg_default
    from leclip_amd.registry import build_trainer

    overrides = env_overrides()
    if args.gemm_family is not None:
        from leclip_amd.hip import ops as _ops
        _ops.set_gemm_family(args.gemm_family)
        overrides = list(overrides) + [f"gemm_family={args.gemm_family}"]
    rank, world, _ = parallel.init_from_env()
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher's WORLD_SIZE={world}")
    B = args.batch if args.batch != 256 else 512
    cfg = get_cfg_default()
    dense = args.tune_model == "DenseCLIP"
    cfg.merge_from_list(["MODEL.BACKBONE.NAME", args.arch, "MODEL.BACKBONE.PATH", "synthetic:0:cond", "TRAINER.Caption.PREC",
                         args.dtype, "TRAIN.LOSSFUNC", "double_ranking" if dense else "bce", "OPTIM.WARMUP_EPOCH", "0"]
                        + (["TRAIN.MODEL", "DenseCLIP", "TRAINER.Caption.use_evidence", "True", "TRAIN.ema", "True"] if dense else []))
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):     # the trainer announces itself on stdout like the reference; stdout carries ONE JSON line
        tr = build_trainer(cfg)
    arch = synth.ARCHS[args.arch]
    labels = torch.from_numpy((synth.uniform(3 + rank, "tune.labels", (B, 80), 0, 1) < 0.04).astype("float32")).to(tr.device)
    if dense:      # captions: the cached class prompts, cycled (the offline image has no merge table for free text); labels = their classes
        from leclip_amd.clip import tokenize
        from leclip_amd.datasets import coco_object_categories
        base = tokenize([f"a photo of a {c.replace('_', ' ')}." for c in coco_object_categories])
        pick = (torch.arange(B) * 7 + 13 * rank) % 80
        images = base[pick].contiguous().to(tr.device)
        labels = torch.zeros(B, 80)
        labels[torch.arange(B), pick] = 1.0
        labels = labels.to(tr.device)
    else:
        images = torch.from_numpy(synth.make_images(B, arch.image_resolution, seed=1234, start=rank * B)).to(tr.device)
    batch = {"img": images, "label": labels}
    if args.no_pipeline or dense:
        tr.pipeline_image_tower = False
        if args.no_pipeline:
            overrides = list(overrides) + ["pipeline_image_tower=False"]
    if args.no_text_beside:
        for root in vars(tr).values():
            if hasattr(root, "text_beside_image"):
                root.text_beside_image = False
        overrides = list(overrides) + ["text_beside_image=False"]

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    from leclip_amd.hip import ops
    from leclip_amd.hip.engine import VisionEngine

    def vision_engines():      # the image engines this trainer's models have built so far (stream parts are switched off on the sampled step)
        engs = []
        for root in vars(tr).values():
            for mod in (root.modules() if isinstance(root, torch.nn.Module) else []):
                if isinstance(getattr(mod, "_engine", None), VisionEngine):
                    engs.append(mod._engine)
        return engs

    # a loader's lookahead: the trainer computes the NEXT batch's frozen-tower features beside this step's backward (forward_backward, next_batch=);
    # the synthetic batch is the same tensor every step, and every step still runs one image tower, one text forward / backward and one update
    nb = None if args.no_pipeline else batch
    for _ in range(args.warmup):
        out = tr.forward_backward(batch, next_batch=nb)
    prof = []
    if args.profile_every > 0:   # the sampled step's one-part workspace, allocated outside the timed region
        engs = vision_engines()
        keep = [(e.streams, e.split_sizes) for e in engs]
        for e in engs:
            e.streams, e.split_sizes = 1, None
        out = tr.forward_backward(batch, next_batch=nb)
        for e, (a_, b_) in zip(engs, keep):
            e.streams, e.split_sizes = a_, b_
    fence()
    sample_at = args.steps // 2 if args.profile_every > 0 else -1
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i == sample_at:   # ONE step of the timed region with a HIP-event pair around every launch, batch as one part (as in score mode)
            engs = vision_engines()
            keep = [(e.streams, e.split_sizes) for e in engs]
            for e in engs:
                e.streams, e.split_sizes = 1, None
            ops.set_profile(prof)
            out = tr.forward_backward(batch, next_batch=nb)
            ops.set_profile(None)
            for e, (a_, b_) in zip(engs, keep):
                e.streams, e.split_sizes = a_, b_
        else:
            out = tr.forward_backward(batch, next_batch=nb)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=tr.device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    if rank == 0:
        what = (f"captions/sec (DenseCLIP caption-as-image tuning step, {args.arch} text tower: {B} captions fwd + 240 prompts fwd/bwd w.r.t. 3 x 16 ctx, "
                f"global + local head, double_ranking + EMA distillation, B={B}/GPU)") if dense else \
               f"images/sec (prompt-tuning step, {args.arch} frozen image tower + text tower fwd/bwd w.r.t. 16 ctx, B={B}/GPU)"
        print(json.dumps({"metric": what,
                          "value": world * B * args.steps / dt, "unit": "captions/s" if dense else "img/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": args.dtype, "data": "synthetic",
                          "config": {"workload": (f"reference's shipped tuning step (TRAIN.MODEL=DenseCLIP): {args.arch}, 3 x 16 learnable context tokens, B={B} captions/GPU, "
                                                  f"double_ranking + EMA KL, SGD") if dense else
                                                 f"BASELINE configs[2]: {args.arch}, 16 learnable context tokens, B={B}/GPU, BCE, SGD",
                                     "global_batch": world * B, "parallelism": f"dp{world}" + ("+allreduce(ctx grads)" if world > 1 else ""),
                                     "independent_towers": "one after the other" if args.no_text_beside else "side by side on HIP streams (text_beside_image)",
                                     "image_tower_lookahead": bool(nb is not None and not dense and not args.no_text_beside)},
                          "roofline": _tune_roofline(prof), "last_loss": out["loss"], "env_overrides": overrides}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _tune_roofline(prof):
    """roofline block of a --mode tune line: the MFMA GEMM family of ONE sampled step of the timed region (forward and backward GEMMs of
    the text tower - M = 80 x 77 = 6 160 rows per prompt set - and, on image batches, the frozen image tower's), from HIP events around every
    launch; per launch shape underneath.  traffic: HBM bytes per GEMM launch from the committed counter passes of this command
    (profiles/r*_tune_pmc_summary.json, collected by profiles/collect.sh extra), or null when there is none."""
    fam, shapes = [0.0, 0, 0], {}
    for name, fl, nb, e0, e1, shape in prof:
        if name != "gemm":
            continue
        sec = e0.elapsed_time(e1) * 1e-3
        fam[0] += sec; fam[1] += fl; fam[2] += 1
        v = shapes.setdefault(shape, [0.0, 0, 0])
        v[0] += sec; v[1] += fl; v[2] += 1
    if not fam[2]:
        return None
    tf = fam[1] / fam[0] * 1e-12
    top = sorted(shapes.items(), key=lambda kv: -kv[1][0])[:8]
    traffic, traffic_src = _pmc_traffic("tune_pmc_summary.json")
    return {"bound": "mfma", "kernel": "gemm_tn_384x256x32_pp / gemm_tn_256x256x64_pp / gemm_tn_128x128x64 (by tile count)", "achieved": tf, "peak": PEAK_MFMA_TFLOPS,
            "unit": "TFLOP/s", "frac": tf / PEAK_MFMA_TFLOPS, "traffic": traffic, "traffic_source": traffic_src, "launches": fam[2], "avg_launch_us": fam[0] / fam[2] * 1e6,
            "gemm_ms_of_step": fam[0] * 1e3, "measured_on": "one sampled step inside the timed region, batch as one part",
            "gemm_shapes": {k: {"launches": v[2], "avg_us": v[0] / v[2] * 1e6, "tflops": v[1] / v[0] * 1e-12,
                                "mfma_frac": v[1] / v[0] * 1e-12 / PEAK_MFMA_TFLOPS} for k, v in top}}


def _pmc_traffic(suffix="pmc_summary.json"):
    """HBM bytes per GEMM launch.  NOT measured by this process: rocprofv3 --pmc cannot run inside the timed run, so the
    figure is read from the committed summary of the separate counter passes over this same command
    (profiles/r<NN>_pmc_summary.json, newest round; r<NN>_tune_pmc_summary.json for the tuning step), and the source file is named next to it."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)))
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            return json.load(f).get("gemm_hbm_bytes_per_launch"), "committed rocprofv3 --pmc passes: profiles/" + os.path.basename(files[-1])
    except Exception:
        return None, None


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_baseline(args, arch, sd, cc, ctx, dev):
    """Time the CPU oracle on the host cores on a bounded sample of the same workload and, on those same images,
    compare mAP of the HIP logits against mAP of the oracle logits (labels from the oracle logits)."""
    import numpy as np
    import torch
    from leclip_amd import synth
    from leclip_amd.evaluation import mAP
    from oracle import clip_oracle as co

    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    affinity = cores          # what this process may run on (printed beside the count used)
    cores = min(cores, 16)   # the GPU box gives one GPU's job a 16-core share; more threads only oversubscribe
    torch.set_num_threads(cores)
    cb = 32
    toks = cc.tokenized_prompts.cpu()
    prefix, suffix = co.prompt_buffers(toks, sd, 16)
    with torch.no_grad():
        txt = co.text_encoder(co.prompt_learner_forward(ctx, prefix, suffix), toks, sd)
        imgs0 = torch.from_numpy(synth.make_images(cb, arch.image_resolution, seed=4321))
        co.encode_image(imgs0[:2], sd)  # warm-up
        ref_logits, hip_logits, spent, n = [], [], 0.0, 0
        while spent < args.cpu_seconds and n < 16:
            imgs = torch.from_numpy(synth.make_images(cb, arch.image_resolution, seed=4321, start=n * cb))
            t0 = time.perf_counter()
            fi = co.encode_image(imgs, sd)
            lg = co.cosine_logits(fi, txt, 4.0)
            spent += time.perf_counter() - t0
            ref_logits.append(lg.numpy())
            hip_logits.append(cc(imgs.to(dev), if_test=True)[0].float().cpu().numpy())
            n += 1
    ref = np.concatenate(ref_logits)
    hip = np.concatenate(hip_logits)
    labels = synth.make_labels_from_logits(ref, seed=7, pos_frac=0.1, noise=0.5)
    # also the B=8 figure (BASELINE configs[0], the reference's own CPU-runnable case): 1 warm-up + best of 3
    with torch.no_grad():
        imgs8 = torch.from_numpy(synth.make_images(8, arch.image_resolution, seed=1234))
        co.cosine_logits(co.encode_image(imgs8, sd), txt, 4.0)
        best8 = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            co.cosine_logits(co.encode_image(imgs8, sd), txt, 4.0)
            best8 = min(best8, time.perf_counter() - t0)
    base = {"value": n * cb / spent, "unit": "img/s", "cores": cores, "affinity_cores": affinity, "host_cores": os.cpu_count(), "kind": "port", "cpu_model": _cpu_model(),
            "b8_value": 8 / best8, "b8_note": "BASELINE configs[0] (B=8): best of 3 after one warm-up, same oracle, same threads",
            "sample": f"{n} batches of {cb} images, fp32 torch-CPU oracle forward + logits, {spent:.1f} s of CPU work, "
                      f"torch {torch.__version__}, {cores} threads"}
    m = {"n_images": int(ref.shape[0]), "oracle_fp32": mAP(labels, ref), "hip": mAP(labels, hip)}
    m.update(label_index_evidence(ref, hip, args.dtype))
    return base, m, {"ref": ref, "labels": labels, "n": n, "cb": cb}


# A-priori logit tolerances per dtype at logit scale 4.0 - the ones the parity tests hold the kernels to (tests/test_gpu_parity.py: fp32 1e-3 = the north
# star's clause, fp16 4e-3, bf16 4e-2; measured over rounds 2 - 4: 5e-6, 2.5e-3 .. 3.2e-3, 2.8e-2 .. 3.3e-2).  The label-index band is min(2 x the run's own
# maximum error, this tolerance) and the maximum error itself must stay inside it: a band derived from the run's error alone can never fail on a uniformly
# worse kernel - these cannot widen.
LABEL_BAND = {"fp32": 1e-3, "fp16": 4e-3, "bf16": 4e-2}


def label_index_evidence(ref, hip, dtype):
    """Label-index agreement between reference fp32 logits and the HIP logits, with the evidence for every image whose top-1 label
    differs: the reference's margin between ITS top-1 and the label the HIP path picked, against the error band min(2 x max |logit
    difference|, LABEL_BAND[dtype]) (the dtype's a-priori logit tolerance).  Two logits closer than the band cannot be ordered by arithmetic with that error, so such a
    disagreement is a tie broken by rounding; a HIP pick that the reference ranks further below its top-1 than the band is a WRONG
    label, however close the reference's own top-2 may be (north star: label-index outputs bit-exact in the fp32 mode, where no
    disagreement occurs)."""
    import numpy as np
    err = float(np.abs(ref - hip).max())
    band = min(2.0 * err, LABEL_BAND[dtype])
    r1, h1 = ref.argmax(1), hip.argmax(1)
    items = []
    for i in np.nonzero(r1 != h1)[0]:
        top2 = np.sort(ref[i])[-2:]
        to_pick = float(ref[i, r1[i]] - ref[i, h1[i]])
        items.append({"image": int(i), "oracle_top1": int(r1[i]), "hip_top1": int(h1[i]),
                      "oracle_top1_top2_margin": float(top2[1] - top2[0]),
                      "oracle_margin_to_hip_pick": to_pick,
                      "inside_error_band": bool(to_pick <= band)})
    return {"max_abs_logit_diff": err, "top1_agree": float((r1 == h1).mean()), "error_band": band, "error_band_cap": LABEL_BAND[dtype],
            "max_error_within_cap": bool(err <= LABEL_BAND[dtype]),
            "logit_spread_std": float(ref.std(1).mean()), "top1_disagreements": items[:16], "n_top1_disagreements": len(items),
            "top1_disagreements_all_inside_band": all(d["inside_error_band"] for d in items)}


def accuracy_gate(m, dtype):
    """North star: mAP within +-0.2 of the reference path, label indices exact up to ties inside the arithmetic's error band.
    bf16 makes no claim on the mAP clause: at N = 2 048 against the reference's own logits it measures -0.25 (tests/test_gpu_parity.py::
    test_cfg4_logits_against_the_reference_at_size, round 4) - the 8-bit mantissa of the activations, which a bf16 run of the reference
    would carry too (profiles/r02_lowprec_error_budget.py) - so bf16 is the RATE-ONLY companion dtype and its gate
    line says so instead of leaving an undecided "MISSED"; the label-index and error-bound checks still apply to it."""
    if dtype == "bf16":
        d = m["hip"] - m.get("reference", m.get("oracle_fp32"))
        bad = []
        if not m["top1_disagreements_all_inside_band"]:
            bad.append("a top-1 disagreement lies outside the error band")
        if not m["max_error_within_cap"]:
            bad.append("max |logit error| exceeds the dtype's constant bound")
        tail = ("; MISSED: " + "; ".join(bad)) if bad else "; label indices exact up to ties inside the constant error band"
        return f"rate-only dtype, mAP clause not claimed (delta {d:+.3f}; decided at N = 2048: -0.25 against the reference)" + tail
    missed = []
    if not abs(m["hip"] - m.get("reference", m.get("oracle_fp32"))) <= 0.2:
        missed.append("|mAP - reference| > 0.2")
    if not m["top1_disagreements_all_inside_band"]:
        missed.append("a top-1 disagreement lies outside the error band")
    if not m["max_error_within_cap"]:
        missed.append("max |logit error| exceeds the dtype's constant bound")
    return f"MISSED in {dtype}: " + "; ".join(missed) if missed else "met (|mAP - reference| <= 0.2; every top-1 disagreement is a tie inside the constant error band)"


def reference_fixture(arch_name, batch, world):
    """tests/golden/vitb16_cfg4_logits.npz: logits of the REFERENCE's own model.py on the images this benchmark scores (rank r:
    synth.make_images(256, seed=1234, start=256 r), r < 8), the labels drawn from them and the reference mAP() (oracle/make_golden.py
    cfg4_goldens).  -> (reference logits, labels) for the first world * batch images, or None when the run is not that workload."""
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", "vitb16_cfg4_logits.npz")
    n = world * batch
    if arch_name != "ViT-B/16" or batch != 256 or n > 2048 or not os.path.exists(path):
        return None
    g = np.load(path)
    labels = np.unpackbits(g["labels"], axis=1)[:, :int(g["n_classes"])].astype(np.int64)
    return g["logits"][:n], labels[:n]


def map_on_gathered_logits(hip, fixture, dtype):
    """mAP of the (gathered) HIP logits of all ranks against the reference fixture's rows: BASELINE configs[3]'s "eval mAP" at N = 8, and
    the same evidence on rank 0's 256 images at N = 1.  No CPU oracle involved: the reference logits and labels are committed data."""
    from leclip_amd.evaluation import mAP
    ref, labels = fixture
    m = {"n_images": int(ref.shape[0]), "reference": mAP(labels, ref), "hip": mAP(labels, hip),
         "source": "tests/golden/vitb16_cfg4_logits.npz (reference model.py logits + labels; oracle/make_golden.py cfg4_goldens)"}
    m["delta"] = m["hip"] - m["reference"]
    m.update(label_index_evidence(ref, hip, dtype))
    m["accuracy_gate"] = accuracy_gate(m, dtype)
    return m


def score_against(pack, cc, arch, dev, dtype="bf16"):
    """mAP of another model (same weights, other dtype) on the images / labels / oracle logits of the cpu_baseline sample."""
    import numpy as np
    import torch
    from leclip_amd import synth
    from leclip_amd.evaluation import mAP
    with torch.no_grad():
        hip = np.concatenate([cc(torch.from_numpy(synth.make_images(pack["cb"], arch.image_resolution, seed=4321, start=i * pack["cb"])).to(dev),
                                 if_test=True)[0].float().cpu().numpy() for i in range(pack["n"])])
    ref = pack["ref"]
    m = {"n_images": int(ref.shape[0]), "oracle_fp32": mAP(pack["labels"], ref), "hip": mAP(pack["labels"], hip)}
    m.update(label_index_evidence(ref, hip, dtype))
    return m


if __name__ == "__main__":
    main()
