"""Headline benchmark: frames/s of TransVOD++ Late-Fusion inference at 800x1333 RGB-D on N MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[4], SURVEY.md 8d "config E"): one synthetic 32-frame RGB-D clip,
frames [32,4,800,1333] ~ N(0,1) (seed 42), no padding, every frame gets an output with the other
31 frames as its reference frames ("all-current" mode, models/clip_inference.py).  Weights are each
module's own initialisation under seed 42 (no checkpoints exist offline).  The clip is FIXED and
its frames are sharded in contiguous blocks over the N ranks (strong scaling): one rank per GPU,
one RCCL all-gather of the per-frame reference query sets per step.  A step = the whole clip.
Inputs are resident in HBM before the timed region.  Compute type fp32 throughout.

The JSON line also carries
  roofline     the MSDA forward kernel at the encoder geometry (csrc/msda_level.hip; N = frames per
               micro-batch = the rank's frames by default, Lq = S = 4200, L = 1): algorithmic bytes
               4*(N*S*256 + 3*N*Lq*8*4 + N*Lq*256) per launch over the mean kernel duration of those
               launches inside the timed steps, taken from HIP events that the launch itself stamps on
               its stream (hipExtLaunchKernelGGL); peak = 8 TB/s HBM3E (MI355X_MICROARCH.md).  With the
               two-stream schedule active (several micro-batches per rank, --overlap 1) the kernel shares
               the CUs with the other stream in the timed region, so the durations come from one extra
               single-stream step after it and both averages are reported.
  cpu_baseline (rank 0, N=1 only) the same path - this repository's host code on CPU tensors with
               the CPU oracle standing in for the two HIP operators - timed on a bounded sample
               (one 12-frame clip at full resolution, ~12 s) on the box's host cores; the same clip then
               goes through the HIP path and the differences are reported (`check_vs_hip_path`: the
               oracle as checker at 800x1333, where the unit tests use small images).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "depth-fusion-in-transformer-based-video-object-detection_amd")
for _p in (PKG, ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# fp32 means fp32: no reduced-precision matmul / convolution modes
torch.backends.cuda.matmul.allow_tf32 = False
torch.backends.cudnn.allow_tf32 = False

# HBM-side traffic of one MSDA encoder-geometry launch, per frame, from the PMC passes committed in
# profiles/r01_pmc_msda_level_N8.md: (2 x FETCH_SIZE + WRITE_SIZE) KiB at N = 8 frames, the
# factor 2 being the gfx950 FETCH_SIZE correction calibrated in the same run
MSDA_TRAFFIC_PER_FRAME = (2 * 24211.8 + 33600.0) * 1024 / 8

# per-frame algorithmic work of config E (all-current mode)
BYTES_PER_FRAME = 4.333e9
FLOPS_PER_FRAME = 343.3e9
HBM_PEAK = 8.0e12
FP32_MFMA_PEAK = 157.3e12


def build(device, num_ref_frames):
    from models import build_model
    from models.config import transvodpp_args
    torch.manual_seed(42)
    model, _, _ = build_model(transvodpp_args("LateFusion", num_ref_frames=num_ref_frames, device=str(device)))
    return model.to(device).eval()


def cpu_baseline(height, width, threads, frames=12):
    """The same host code on CPU tensors, the oracle as the MSDA / RoIAlign operator."""
    from oracle import msda_oracle
    import models.ops.functions.ms_deform_attn_func as f
    from dfx import ops
    from models.clip_inference import ClipRunner
    torch.set_num_threads(threads)
    msda_oracle.set_threads(threads)
    saved = (f.MSDeformAttnFunction, ops.roi_align)

    def roi_align(inp, rois, output_size, spatial_scale, sampling_ratio, aligned=True, channels_last=False):
        size = output_size if isinstance(output_size, int) else output_size[0]
        if channels_last:
            out = msda_oracle.roi_align(inp.permute(0, 3, 1, 2).contiguous(), rois, size, spatial_scale,
                                        sampling_ratio, aligned)
            return out.flatten(2).transpose(1, 2).contiguous()
        return msda_oracle.roi_align(inp, rois, size, spatial_scale, sampling_ratio, aligned)

    f.MSDeformAttnFunction, ops.roi_align = msda_oracle.OracleMSDAFunction, roi_align
    try:
        model = build("cpu", frames - 1)
        clip = torch.randn(frames, 4, height, width, generator=torch.Generator().manual_seed(42))
        runner = ClipRunner(model, micro_batch=1)
        t0 = time.perf_counter()
        want = runner(clip)
        dt = time.perf_counter() - t0
    finally:
        f.MSDeformAttnFunction, ops.roi_align = saved
    line = {"value": round(frames / dt, 4), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"one {frames}-frame clip at {height}x{width}, all-current mode, {dt:.1f} s of CPU work, "
                      "torch CPU ops + oracle/msda_oracle.c for MSDA and RoIAlign"}
    # the same clip through the HIP path (same seed -> same weights): the checker role of the oracle, at full resolution
    got = ClipRunner(build(torch.device("cuda", torch.cuda.current_device()), frames - 1), micro_batch=frames)(clip.cuda())
    line["check_vs_hip_path"] = {
        "max_abs_diff_pred_logits": float((got["pred_logits"].cpu() - want["pred_logits"]).abs().max()),
        "max_abs_diff_pred_boxes": float((got["pred_boxes"].cpu() - want["pred_boxes"]).abs().max()),
        "temporal_topk_sets_equal": all(set(a.tolist()) == set(b.tolist())
                                        for pg, pc in zip(got["topk"], want["topk"]) for a, b in zip(pg.cpu(), pc))}
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=32, help="clip length T (fixed across N)")
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--width", type=int, default=1333)
    ap.add_argument("--micro-batch", type=int, default=32,
                    help="frames per pass through the spatial stage (capped at the frames of the rank); larger "
                         "is faster up to the whole block: 8 -> 143.3, 16 -> 131.6, 32 -> 127.3 ms per 32-frame clip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--overlap", type=int, default=1,
                    help="two-stream schedule inside a rank: backbones of micro-batch i+1 beside the transformer "
                         "tail of micro-batch i (ClipRunner; same results)")
    ap.add_argument("--pipeline", type=int, default=0,
                    help="clips are a stream: queue each step with ClipRunner.submit, so the transformer tail of clip k "
                         "runs on a second HIP stream beside the backbones of clip k+1 (same results; 0 = one clip at a "
                         "time on one stream - the default, so that the MSDA kernel's stamped times in the timed region are "
                         "its own; 1 = on for a single GPU: +3-4 %% frames/s; 2 = on for N > 1 as well)")
    ap.add_argument("--deterministic", type=int, default=0,
                    help="ask MIOpen for run-to-run deterministic convolution solvers (costs ~6 %% here)")
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the multi-rank "
                         "code path with several ranks on one GPU)")
    ap.add_argument("--conv-autotune", type=int, default=0,
                    help="let MIOpen time its fp32 solvers per convolution shape during warm-up (cudnn.benchmark)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N>1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the operators have no CPU path)")
    if a.backend != "nccl":                       # rehearsal: several ranks may share one GPU
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    torch.backends.cudnn.benchmark = bool(a.conv_autotune)
    torch.backends.cudnn.deterministic = bool(a.deterministic)
    if world > 1:
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)      # RCCL on ROCm
        else:
            dist.init_process_group(a.backend)
    assert a.frames % world == 0, "the clip must split evenly over the ranks"
    per_rank = a.frames // world

    from dfx import _lib, ops
    from models.clip_inference import ClipRunner
    _lib.load()                                                     # fail loudly without the HIP library
    model = build(device, a.frames - 1)
    clip = torch.randn(a.frames, 4, a.height, a.width, generator=torch.Generator().manual_seed(42))
    mine = clip[rank * per_rank:(rank + 1) * per_rank].to(device)   # resident in HBM before timing
    runner = ClipRunner(model, micro_batch=min(a.micro_batch, per_rank), overlap=bool(a.overlap))
    n_micro = -(-per_rank // min(a.micro_batch, per_rank))
    overlapped = bool(a.overlap) and n_micro >= ClipRunner.MIN_OVERLAP_BATCHES
    # N > 1: the exchange would sit on the side stream; that is how torch.distributed's NCCL backend is meant to be
    # used, but it could not be exercised on RCCL in this round (one GPU per session), so it needs --pipeline 2
    pipelined = n_micro == 1 and (a.pipeline >= 2 or (a.pipeline == 1 and world == 1))
    step = (lambda: runner.submit(mine)) if pipelined else (lambda: runner(mine))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    barrier()
    ops.profile_start()                                            # MSDA kernels stamp their own begin/end events
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    launches = ops.profile_stop()
    timed_region_launches = None
    if overlapped or pipelined:
        # In the timed region the MSDA kernel shares the CUs with the other stream's convolutions, so
        # its stamped duration there is not the kernel's own.  The roofline figures come from one extra
        # single-stream step (same inputs, same kernels) after the timed region; both are reported.
        timed_region_launches = launches
        saved_overlap, runner.overlap = runner.overlap, False
        runner(mine)
        barrier()
        ops.profile_start()
        runner(mine)
        barrier()
        launches = ops.profile_stop()
        runner.overlap = saved_overlap

    t = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    if rank == 0:
        fps = a.frames * a.steps / dt
        enc = [(sec, nbytes) for (sec, nbytes, lq, s) in launches if lq == s and sec > 0]
        roof = None
        if enc:
            mean_t = sum(x for x, _ in enc) / len(enc)
            nbytes = enc[0][1]
            roof = {"bound": "hbm", "kernel": "msda_fused_level<2> (encoder / late-fusion geometry, level in LDS)",
                    "achieved": round(nbytes / mean_t / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                    "frac": round(nbytes / mean_t / HBM_PEAK, 4),
                    "traffic": int(MSDA_TRAFFIC_PER_FRAME * min(a.micro_batch, per_rank)),
                    "traffic_source": "PMC FETCH_SIZE/WRITE_SIZE, profiles/r01_pmc_msda_level_N8.md",
                    "launches": len(enc), "bytes_per_launch": nbytes, "avg_launch_us": round(mean_t * 1e6, 2),
                    "measured": "HIP events stamped by the launch, timed region"}
            if timed_region_launches is not None:
                shared = [sec for (sec, _, lq, s) in timed_region_launches if lq == s and sec > 0]
                roof["measured"] = ("HIP events stamped by the launch, one extra single-stream step after the timed "
                                    "region (two-stream schedule / clip pipeline off)")
                roof["avg_launch_us_timed_region_shared_cus"] = round(sum(shared) / max(len(shared), 1) * 1e6, 2)
        line = {
            "metric": "frames/sec at 800x1333 RGB-D, TransVOD++ Late-Fusion", "value": round(fps, 3),
            "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"TransVOD++ LateFusion, {a.frames}-frame {a.height}x{a.width} RGB-D clip, "
                                   f"all-current mode (R={a.frames - 1}), L=1 DC5 (S=4200), 300 queries, 3 classes",
                       "frames_per_gpu": per_rank, "micro_batch": min(a.micro_batch, per_rank),
                       "parallelism": f"frame-shard x{world} + 1 all-gather/clip",
                       "two_stream_overlap": overlapped, "clip_pipeline": pipelined},
            "roofline": roof,
            "e2e": {"hbm_frac": round(fps / world * BYTES_PER_FRAME / HBM_PEAK, 4),
                    "fp32_mfma_frac": round(fps / world * FLOPS_PER_FRAME / FP32_MFMA_PEAK, 4),
                    "bytes_per_frame": BYTES_PER_FRAME, "flops_per_frame": FLOPS_PER_FRAME},
        }
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(a.height, a.width, min(a.cpu_threads, os.cpu_count() or 1))
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
