"""Headline benchmark: frames/s of TransVOD++ Late-Fusion inference at 800x1333 RGB-D on N MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: starts its N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[4], SURVEY.md 8d "config E"): synthetic 32-frame RGB-D clips,
frames [32,4,800,1333] ~ N(0,1) (clip c: seed 42 + c), no padding, every frame gets an output with the other
31 frames of its clip as its reference frames ("all-current" mode, models/clip_inference.py).  Weights are each
module's own initialisation under seed 42 (no checkpoints exist offline).  Every clip is sharded in contiguous
blocks of 32/N frames over the N ranks (4 frames per GPU at N = 8), one rank per GPU, and the per-frame reference
query sets are exchanged with one RCCL all-gather per step.  A step serves ONE clip (BASELINE.json configs[4]: one
32-frame clip sharded 32/N frames per GPU - strong scaling: `value` = 32 * K / time whatever N is).  For N > 1 the
line also carries `clip_stream_throughput`, measured after the timed region: N clips per step, every rank runs its
32/N frames of each of them, i.e. 32 frames per rank and step whatever N is (the throughput mode of a stream of clips,
weak scaling; --clips-per-step 0 makes it the timed mode).
Inputs are resident in HBM before the timed region.  Compute type fp32 throughout.
Tensors derived from the padding mask alone (resized masks, sine positional embeddings, valid ratios, reference grids) are
memoised on the mask tensor (util/memo.py: keyed on tensor identity + version); the runner hands the same all-valid mask
object to every step, so the timed steps do not recompute them - a server with a fixed frame size would not either.

Every MSDeformAttn's sampling_offsets.weight (zero at initialisation, so that all queries would share one offset
pattern) gets a fixed seeded perturbation N(0, 0.13^2): offsets then vary by ~2-3 pixels from query to query, the
access pattern of a trained checkpoint, on the GPU model and the CPU baseline alike.

The JSON line also carries
  roofline     the kernel family with the largest share of GPU time in the timed steps - the fp32 MFMA GEMM
               (csrc/gemm_f32.hip: every 1x1 convolution and Linear): sum of 2*M*N*K flops over its launches /
               sum of their durations, against the dense fp32 matrix peak 157.3 TFLOP/s (MI355X_MICROARCH.md).
               Durations are HIP events that each launch stamps on its own stream (hipExtLaunchKernelGGL) inside
               the timed region.
  roofline_kernels  the same for every stamped family with its share of the step: GEMM by operand form, the fused
               Winograd convolution (executed MFMA flops; the direct form's flops are 2.25x as many), the
               implicit-GEMM convolution, and the MSDA forward kernel at the encoder geometry (csrc/msda_level.hip;
               HBM-bound: algorithmic bytes 4*(N*S*256 + 3*N*Lq*8*4 + N*Lq*256) per launch against 8 TB/s, PMC
               traffic from profiles/).  With the two-stream schedule or the clip pipeline active the kernels share
               the CUs with the other stream, so the durations come from one extra single-stream step.
  cpu_baseline (rank 0, N=1 only) the same path - this repository's host code on CPU tensors with
               the CPU oracle standing in for the two HIP operators - timed on clip 0 of the GPU workload (32 frames
               at full resolution, R = 31; 3 timed passes after a 2-frame warm-up pass, ~31 s each: value = the median,
               best / worst beside it) on the box's host cores; the same
               clip then goes through the HIP path and the differences are reported (`check_vs_hip_path`: floating
               outputs, PostProcess labels / box indices and the ordered temporal top-k picks compared index by index
               outside a 2e-5 score tie margin, with the two ranking heads rescaled - on both sides - so that at least
               four ranks in five lie outside it: the oracle as checker at 800x1333).
  literal_mode (N=1 only) SURVEY.md 8d's secondary figure: the detector's own forward over the 32 frames -> ONE output (frame 0
               current), timed with the reference's protocol (benchmark.py:31-43: a device synchronize around every forward).
  e2e          frames/s x the per-frame algorithmic work of config E (tools/algorithmic_work.py) against the chip's peaks:
               `fp32_mfma_frac` counts the 3x3 stride-1 convolutions in DIRECT-form flops (what the reference computes),
               `fp32_mfma_frac_executed` in the flops the Winograd kernel executes (1/2.25 of them).
  ms_per_step_p50  median GPU time of a step from HIP events recorded after each step on the stream its last kernel
               runs on (the completion events of the clip pipeline's steps - consecutive completions of one lane divided
               by the number of lanes -, else events on the current stream; no host sync inside the timed region).
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

# The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The clip pipeline uses
# two streams per eager lane / one per graph lane beside the default stream: with 4 queues two of the five streams of the
# one-GPU default share a queue and partly serialise (tools/hw_queues_ab.sh, profiles/r04_hw_queues.txt: 391.6 -> 395.8 frames/s
# with 8 queues; 4 frames per rank: 11.45 ms on 3 graph lanes -> 11.13 on 4).  Read when the runtime initialises: set first.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# fp32 means fp32: no reduced-precision matmul / convolution modes
torch.backends.cuda.matmul.allow_tf32 = False
torch.backends.cudnn.allow_tf32 = False

# Fabric-side traffic (HBM + Infinity Cache: 2 x FETCH_SIZE - the gfx950 correction - + WRITE_SIZE, separate PMC passes) per
# kernel family and 32-frame step: profiles/pmc_traffic.json, written by tools/pmc_traffic.py from the PMC passes over this
# workload (tools/profile_round_a.sh); counters cannot be read inside a timed run, so the line quotes the committed passes
with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as _fh:
    _PMC = json.load(_fh)["families"]


def _family_traffic(name):
    f = _PMC[name]
    return f["fetch_bytes_per_step"] + f["write_bytes_per_step"], f["launches_per_step"]


FAMILY_TRAFFIC_PER_STEP = {"gemm": _family_traffic("gemm_f32_kernel"), "wino": _family_traffic("conv_wino_kernel"),
                           "igemm": _family_traffic("conv_igemm_kernel")}
MSDA_TRAFFIC_PER_FRAME = _family_traffic("msda_fused_level")[0] / _family_traffic("msda_fused_level")[1] / 32

# per-frame algorithmic work of config E in all-current mode and the one-pass bytes of the kernel families: generated by
# tools/algorithmic_work.py (a walk over the built model at 800x1333) into tools/algorithmic_work.json; BASELINE.md
# section 2 quotes 4.333 GB / 343.3 GFLOP for the same quantity (the walk gives +1.5 % bytes, -0.1 % flops)
with open(os.path.join(ROOT, "tools", "algorithmic_work.json")) as _fh:
    _WORK = json.load(_fh)["E"]
BYTES_PER_FRAME = _WORK["per_frame_all_current"]["bytes"]
FLOPS_PER_FRAME = _WORK["per_frame_all_current"]["flops"]
FAMILY_WORK = _WORK["kernel_families_all_current"]          # per frame: act / weights bytes, flops, launches
# ... with the 3x3 stride-1 convolutions counted as the Winograd kernel executes them: F(2x2,3x3) = 16 products per 2x2 output
# tile where the direct form has 36
FLOPS_EXECUTED_PER_FRAME = FLOPS_PER_FRAME - FAMILY_WORK["wino"]["flops"] * (1 - 16.0 / 36.0)
BASELINE_MD_PER_FRAME = {"bytes": 4.333e9, "flops": 343.3e9}
HBM_PEAK = 8.0e12
FP32_MFMA_PEAK = 157.3e12
OFFSET_PERTURBATION = 0.13


def build(device, num_ref_frames):
    from models import build_model
    from models.config import transvodpp_args
    from models.ops.modules import MSDeformAttn
    torch.manual_seed(42)
    model, _, _ = build_model(transvodpp_args("LateFusion", num_ref_frames=num_ref_frames, device=str(device)))
    g = torch.Generator().manual_seed(4242)
    with torch.no_grad():                     # per-query sampling offsets like a trained checkpoint's (see docstring)
        for mod in model.modules():
            if isinstance(mod, MSDeformAttn):
                w = mod.sampling_offsets.weight
                w.add_((torch.randn(w.shape, generator=g) * OFFSET_PERTURBATION).to(w.device))
    return model.to(device).eval()


def literal_mode(model, clip, warm=5, iters=30):
    """SURVEY.md 8d's secondary figure: the detector's own forward as the reference times it - the T = 1 + R frames of a clip
    go in, frame 0 is the current frame, ONE output comes back (deformable_detr_multi_plusplus.py:210 of the reference) -
    with the reference's protocol (benchmark.py:31-43: warm-up, then a device synchronize around every forward, perf_counter).
    Not the headline: in all-current mode every frame of the clip gets an output.  -> dictionary for the JSON line"""
    from util.misc_multi import NestedTensor
    samples = NestedTensor(clip, torch.zeros(clip.shape[0], clip.shape[2], clip.shape[3], dtype=torch.bool, device=clip.device))
    times = []
    with torch.no_grad():
        for i in range(warm + iters):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = model(samples)
            torch.cuda.synchronize()
            if i >= warm:
                times.append(time.perf_counter() - t0)
    assert out["pred_logits"].shape[0] == 1 and bool(torch.isfinite(out["pred_logits"]).all())
    times.sort()
    mean = sum(times) / len(times)
    T = clip.shape[0]
    return {"clips_per_s": round(1.0 / mean, 3), "input_frames_per_s": round(T / mean, 2), "ms_per_forward_mean": round(mean * 1e3, 3),
            "ms_per_forward_p50": round(times[len(times) // 2] * 1e3, 3), "iterations": iters, "warmup": warm,
            "note": f"one forward of the detector over {T} frames -> 1 output (frame 0 current, R = {T - 1}), a device synchronize "
                    "around every forward; the spatial stage of all frames + ONE temporal stage - the number comparable to timing "
                    "the reference's forward, not the headline"}


def cpu_baseline(height, width, threads, frames=32, warm_frames=2, passes=3):
    """The same host code on CPU tensors, the oracle as the MSDA / RoIAlign operator: clip 0 of the GPU workload
    (``frames`` frames, seed 42, R = frames - 1), ``passes`` timed passes after a ``warm_frames``-frame warm-up pass
    (SURVEY.md 8d / the reference's benchmark.py:31-43 average several); ``value`` is the median pass, min / max ride along.
    The same clip then goes through the HIP path and is compared (tests/_config_e_check.py, shared with
    tests/test_configs_gpu.py::test_config_e_32_frame_clip_800x1333)."""
    from tests import _config_e_check as chk
    clip = torch.randn(frames, 4, height, width, generator=torch.Generator().manual_seed(42))
    want, heads, seconds = chk.cpu_reference_clip(build, clip, threads, timed_passes=passes, warm_frames=warm_frames)
    srt = sorted(seconds)
    med = srt[len(srt) // 2]
    line = {"value": round(frames / med, 4), "unit": "frames/s", "cores": threads, "kind": "port",
            "passes": len(seconds), "frames_per_s_best": round(frames / srt[0], 4), "frames_per_s_worst": round(frames / srt[-1], 4),
            "seconds_per_pass": [round(x, 2) for x in seconds],
            "sample": f"clip 0 of the GPU workload: {frames} frames at {height}x{width}, all-current mode (R = {frames - 1}), "
                      f"{len(seconds)} timed passes after a {warm_frames}-frame warm-up pass, value = the median pass "
                      f"({med:.1f} s; {sum(seconds):.0f} s of CPU work in all), torch CPU ops + oracle/msda_oracle.c for MSDA "
                      "and RoIAlign"}
    got = chk.hip_path_clip(build, clip, heads)
    line["check_vs_hip_path"] = chk.compare(got, want, heads, height, width)
    return line


def launch_ranks(n):
    """``python bench.py --gpus N`` without a launcher: start the N ranks as fresh child processes (one per GPU) through
    torch.distributed.run with the arguments of this call, as the reference's own launcher forks its ranks
    (/root/reference/tools/launch.py:159-192).  The children inherit stdout, so rank 0's JSON line is this command's
    output; -> the launcher's return code.  Called before this process has touched the GPU; it never execs."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=32, help="clip length T (fixed across N)")
    ap.add_argument("--clips-per-step", type=int, default=1,
                    help="clips served per step, each sharded over the N ranks (1 = one clip in flight, T/N frames per rank and "
                         "step: BASELINE.json's configuration, strong scaling - the default; 0 = N clips per step: every rank runs "
                         "T frames per step, the throughput mode of a stream of clips, weak scaling)")
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--width", type=int, default=1333)
    ap.add_argument("--micro-batch", type=int, default=32,
                    help="frames per pass through the spatial stage (capped at the frames of the rank); larger "
                         "is faster up to the whole block: 8 -> 143.3, 16 -> 131.6, 32 -> 127.3 ms per 32-frame clip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-literal", action="store_true", help="skip the secondary literal-mode figure (one output per clip)")
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--overlap", type=int, default=1,
                    help="two-stream schedule inside a rank: backbones of micro-batch i+1 beside the transformer "
                         "tail of micro-batch i (ClipRunner; same results)")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="clips are a stream: queue each step with ClipRunner.submit, so the transformer tail of clip k - and "
                         "for N > 1 the exchange - runs on a second HIP stream beside the backbones of clip k+1 (same results; "
                         "+4 %% frames/s at 32 frames per GPU, +13 %% at 4; 0 = one clip at a time on one stream; 1 = on, the "
                         "default for every N since round 3: the per-kernel roofline durations then come from one extra "
                         "single-stream step)")
    ap.add_argument("--lanes", type=int, default=0,
                    help="independent stream pairs the clip pipeline deals consecutive steps to (ClipRunner(lanes=...): each lane "
                         "keeps two clips in flight).  0 = 2 eager lanes, or 4 HIP-graph lanes (3 with fewer than 8 hardware queues) "
                         "(tools/rank_step.py, profiles/r04_rank_step.txt: 4 frames per step 12.8 -> 11.6 ms, 8 frames 25.2 -> 21.7, "
                         "16 frames 43.2 -> 42.4, 32 frames 82.5 -> 81.8 ms with two lanes; three add nothing)")
    ap.add_argument("--graph", type=int, default=-1,
                    help="1: the clip pipeline replays every rank step as HIP graphs (ClipRunner(graph=True): spatial stage + "
                         "query/RoI fusion, eager exchange, temporal stage) instead of launching ~1500 kernels from Python - same "
                         "kernels, bit-equal outputs, host time per step from 10.8 ms to 0.3 ms at 4 frames per rank, where the "
                         "eager host is 88 %% busy; a lane is then one stream (4 lanes).  -1 (default) = on for every N > 1 (ranks share "
                         "a host; 12.3 -> 11.6 ms per 4-frame step, level at 8 and 16 frames: profiles/r04_rank_step.txt), off for "
                         "one GPU with 32 frames per step (eager 1 %% ahead)")
    ap.add_argument("--deterministic", type=int, default=0,
                    help="(no-op since round 2: no library convolution is left in the path; every kernel is run-to-run deterministic)")
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the multi-rank "
                         "code path with several ranks on one GPU)")
    ap.add_argument("--conv-autotune", type=int, default=0,
                    help="(no-op since round 2: no library convolution is left in the path)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(a.gpus))       # no GPU call has happened in this process
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus")
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
    clips = a.clips_per_step if a.clips_per_step > 0 else world
    block = []                                                      # this rank's frames of every clip, clip-major
    for c in range(clips):
        clip = torch.randn(a.frames, 4, a.height, a.width, generator=torch.Generator().manual_seed(42 + c))
        block.append(clip[rank * per_rank:(rank + 1) * per_rank].to(device))
        del clip
    mine = torch.cat(block, 0) if clips > 1 else block[0]          # resident in HBM before timing
    del block
    rank_frames = clips * per_rank                                  # frames a rank runs per step
    # graphs by default whenever the clip is sharded (several ranks share one host: 85-88 % of an eager step is Python launch
    # time at 4-16 frames per rank, profiles/r04_rank_step.txt) or the step is short; one GPU with 32 frames stays eager (1 % ahead)
    use_graph = bool(a.graph) if a.graph >= 0 else ((world > 1 or rank_frames <= 4) and a.pipeline >= 1)
    hw_queues = int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4)
    lanes = a.lanes if a.lanes > 0 else ((4 if hw_queues >= 8 else 3) if use_graph else 2)
    runner = ClipRunner(model, micro_batch=min(a.micro_batch, rank_frames), overlap=bool(a.overlap), lanes=lanes, graph=use_graph)
    n_micro = -(-rank_frames // min(a.micro_batch, rank_frames))
    overlapped = bool(a.overlap) and n_micro >= ClipRunner.MIN_OVERLAP_BATCHES
    # N > 1: the exchange sits on the side stream - how torch.distributed's NCCL backend is meant to be used; exercised on
    # RCCL with one rank (tests/test_clip_shard_gpu.py::test_rccl_call_path_with_one_rank) and with two ranks over gloo
    pipelined = n_micro == 1 and a.pipeline >= 1
    dones = []                                                      # completion events of the pipelined steps (HIP events)

    def step():
        if pipelined:
            dones.append(runner.submit(mine, clips=clips)[1])
        else:
            runner(mine, clips=clips)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if pipelined:             # set-up, not steps: every lane's streams / tables / HIP graphs exist before the W warm-up steps
        for _ in range(lanes):
            step()
        barrier()
        dones.clear()
    for _ in range(a.warmup):
        step()
    barrier()
    ops.profile_start()                                            # kernels stamp their own begin/end events
    # step marks without a host sync: with the clip pipeline the completion event of every step (recorded by submit on the
    # lane's tail stream, where the step's last kernel runs; the first interval starts at the last warm-up step's completion),
    # else events on the current stream
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    first_timed = len(dones)
    t0 = time.perf_counter()
    if not pipelined:
        marks[0].record(torch.cuda.current_stream(device))
    for i in range(a.steps):
        step()
        if not pipelined:
            marks[i + 1].record(torch.cuda.current_stream(device))
    barrier()
    dt = time.perf_counter() - t0
    launches = ops.profile_stop()
    stride = 1
    if pipelined:       # consecutive completions on ONE lane (one stream: ordered) are `lanes` steps apart
        stride = lanes
        marks = dones[max(first_timed - lanes, 0):]
    step_ms = sorted(marks[i].elapsed_time(marks[i + stride]) / stride for i in range(len(marks) - stride)) or [dt / a.steps * 1e3]
    timed_region_launches = None
    if overlapped or pipelined:
        # In the timed region the MSDA kernel shares the CUs with the other stream's convolutions, so
        # its stamped duration there is not the kernel's own.  The roofline figures come from one extra
        # single-stream step (same inputs, same kernels) after the timed region; both are reported.
        timed_region_launches = launches
        saved_overlap, runner.overlap = runner.overlap, False
        runner(mine, clips=clips)
        barrier()
        ops.profile_start()
        runner(mine, clips=clips)
        barrier()
        launches = ops.profile_stop()
        runner.overlap = saved_overlap

    # beside the headline (one clip in flight), the throughput mode of a stream of clips: N clips per step, every rank runs
    # its T/N frames of each of them (T frames per step and rank whatever N is; one all-gather per step all the same)
    dt_stream = None
    if world > 1 and clips == 1:
        block = [mine]
        for c in range(1, world):
            clip = torch.randn(a.frames, 4, a.height, a.width, generator=torch.Generator().manual_seed(42 + c))
            block.append(clip[rank * per_rank:(rank + 1) * per_rank].to(device))
            del clip
        many = torch.cat(block, 0)
        del block
        # the same clip pipeline as the one-GPU run (two eager lanes: every rank runs 32 frames per step here) when the block
        # goes through in one pass; one call per step otherwise
        wide = ClipRunner(model, micro_batch=min(a.micro_batch, world * per_rank), overlap=bool(a.overlap), lanes=2)
        piped = pipelined and min(a.micro_batch, world * per_rank) >= world * per_rank
        wide_step = (lambda: wide.submit(many, clips=world)) if piped else (lambda: wide(many, clips=world))
        for _ in range(max(2, a.warmup)):
            wide_step()
        barrier()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            wide_step()
        barrier()
        dt_stream = time.perf_counter() - t1

    t = torch.tensor([dt, dt_stream or 0.0], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t[0].item())
    if dt_stream is not None:
        dt_stream = float(t[1].item())

    if rank == 0:
        fps = clips * a.frames * a.steps / dt
        steps_profiled = a.steps if timed_region_launches is None else 1
        step_s = dt / a.steps
        fam = {-1: ("gemm_f32_kernel, [K,N] operand (1x1 convolutions of the backbones / input_proj)", "mfma"),
               -2: ("gemm_f32_kernel, [N,K] operand (Linear layers)", "mfma"),
               -3: ("conv_wino_kernel (3x3 stride-1 convolutions, Winograd F(2x2,3x3); executed MFMA flops, the direct "
                    "form has 2.25x as many)", "mfma"),
               -4: ("conv_igemm_kernel (7x7/2 stem, 3x3/2, DFormer stem)", "mfma")}
        kernels = []
        full_size = a.frames == 32 and a.height == 800 and a.width == 1333

        def family_bytes(key, frames):       # one-pass bytes of a family per step of `frames` frames (weights once per launch)
            w = FAMILY_WORK[key]
            return w["act"] * frames + w["weights"]

        for tag, (name, bound) in fam.items():
            rec = [(sec, work) for (sec, work, ta, tb) in launches if ta == tag and sec > 0]
            if not rec:
                continue
            tsum, wsum = sum(x for x, _ in rec), sum(w for _, w in rec)
            tr_key = {-3: "wino", -4: "igemm"}.get(tag)
            traffic = algo = None
            if tr_key and full_size:      # per launch, scaled to the rank's frames
                traffic = int(FAMILY_TRAFFIC_PER_STEP[tr_key][0] / FAMILY_TRAFFIC_PER_STEP[tr_key][1] * rank_frames / 32)
                algo = int(family_bytes(tr_key, rank_frames) / (len(rec) / steps_profiled))
            kernels.append({"kernel": name, "bound": bound, "achieved": round(wsum / tsum / 1e12, 2), "peak": FP32_MFMA_PEAK / 1e12,
                            "unit": "TFLOP/s", "frac": round(wsum / tsum / FP32_MFMA_PEAK, 4), "traffic": traffic,
                            "algorithmic_bytes": algo,
                            "traffic_over_algorithmic": round(traffic / algo, 2) if traffic and algo else None,
                            "launches_per_step": round(len(rec) / steps_profiled, 1),
                            "flops_per_step": wsum / steps_profiled, "ms_per_step": round(tsum / steps_profiled * 1e3, 3),
                            "share_of_step": round(tsum / steps_profiled / step_s, 4)})
        gemm = [(sec, work) for (sec, work, ta, tb) in launches if ta in (-1, -2) and sec > 0]
        roof = None
        if gemm:
            tsum, wsum = sum(x for x, _ in gemm), sum(w for _, w in gemm)
            roof = {"bound": "mfma", "kernel": "gemm_f32_kernel (fp32 MFMA GEMM: every 1x1 convolution and Linear; the family "
                                               "with the largest share of GPU time)",
                    "achieved": round(wsum / tsum / 1e12, 2), "peak": FP32_MFMA_PEAK / 1e12, "unit": "TFLOP/s",
                    "frac": round(wsum / tsum / FP32_MFMA_PEAK, 4),
                    "traffic": (int(FAMILY_TRAFFIC_PER_STEP["gemm"][0] / FAMILY_TRAFFIC_PER_STEP["gemm"][1] * rank_frames / 32)
                                if full_size else None),
                    "algorithmic_bytes": int(family_bytes("gemm", rank_frames) / (len(gemm) / steps_profiled)) if full_size else None,
                    "traffic_source": "PMC 2 x FETCH_SIZE + WRITE_SIZE (HBM + Infinity Cache side of L2), average per launch of "
                                      "the family, profiles/pmc_traffic.json (tools/pmc_traffic.py)",
                    "algorithmic_source": "tools/algorithmic_work.py: one-pass bytes of every 1x1 convolution and Linear of a step "
                                          "(inputs + outputs + weights + bottleneck residual reads), average per launch",
                    "launches": len(gemm), "flops_per_launch": wsum / len(gemm), "avg_launch_us": round(tsum / len(gemm) * 1e6, 2),
                    "share_of_step": round(tsum / steps_profiled / step_s, 4),
                    "measured": "HIP events stamped by each launch, timed region; achieved = sum of 2*M*N*K / sum of durations"}
            if roof["traffic"] and roof["algorithmic_bytes"]:
                roof["traffic_over_algorithmic"] = round(roof["traffic"] / roof["algorithmic_bytes"], 2)
        enc = [(sec, nbytes) for (sec, nbytes, lq, s) in launches if lq > 0 and lq == s and sec > 0]
        if enc:
            mean_t = sum(x for x, _ in enc) / len(enc)
            nbytes = enc[0][1]
            msda = {"kernel": "msda_fused_level<2> (MSDA forward, encoder / late-fusion geometry, level in LDS, per-query offsets)",
                    "bound": "hbm", "achieved": round(nbytes / mean_t / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                    "frac": round(nbytes / mean_t / HBM_PEAK, 4),
                    "traffic": int(MSDA_TRAFFIC_PER_FRAME * min(a.micro_batch, rank_frames)),
                    "traffic_source": "PMC FETCH_SIZE/WRITE_SIZE, profiles/pmc_traffic.json",
                    "launches_per_step": round(len(enc) / steps_profiled, 1), "bytes_per_launch": nbytes,
                    "avg_launch_us": round(mean_t * 1e6, 2),
                    "share_of_step": round(mean_t * len(enc) / steps_profiled / step_s, 4)}
            if timed_region_launches is not None:
                shared = [sec for (sec, _, lq, s) in timed_region_launches if lq > 0 and lq == s and sec > 0]
                msda["avg_launch_us_timed_region_shared_cus"] = round(sum(shared) / max(len(shared), 1) * 1e6, 2)
            kernels.append(msda)
        if roof is not None and timed_region_launches is not None:
            roof["measured"] = ("HIP events stamped by each launch, one extra single-stream step after the timed region "
                                "(two-stream schedule / clip pipeline off)")
        line = {
            "metric": "frames/sec at 800x1333 RGB-D, TransVOD++ Late-Fusion", "value": round(fps, 3),
            "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "ms_per_step_p50": round(step_ms[len(step_ms) // 2], 3),
            "higher_is_better": True, "scaling": "strong" if clips == 1 else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "rccl": {"backend": dist.get_backend() if world > 1 else None,
                     "world": dist.get_world_size() if world > 1 else 1},
            "config": {"workload": f"TransVOD++ LateFusion, {a.frames}-frame {a.height}x{a.width} RGB-D clip, "
                                   f"all-current mode (R={a.frames - 1}), L=1 DC5 (S={-(-a.height // 16) * -(-a.width // 16)}), 300 queries, 3 classes",
                       "clips_per_step": clips, "frames_per_gpu": rank_frames, "frames_per_gpu_per_clip": per_rank,
                       "micro_batch": min(a.micro_batch, rank_frames),
                       "parallelism": f"every clip frame-sharded x{world} ({per_rank} frames/GPU), {clips} clip(s) per step, "
                                      f"1 all-gather of the reference query sets per step",
                       "two_stream_overlap": overlapped, "clip_pipeline": pipelined, "pipeline_lanes": lanes if pipelined else 0, "hip_hw_queues": hw_queues,
                       "hip_graphs": use_graph and pipelined and bool(runner._graph_slots) and all(v is not False for v in runner._graph_slots.values())},
            "roofline": roof,
            "roofline_kernels": kernels,
            "e2e": {"hbm_frac": round(fps / world * BYTES_PER_FRAME / HBM_PEAK, 4),
                    "fp32_mfma_frac": round(fps / world * FLOPS_PER_FRAME / FP32_MFMA_PEAK, 4),
                    "fp32_mfma_frac_executed": round(fps / world * FLOPS_EXECUTED_PER_FRAME / FP32_MFMA_PEAK, 4),
                    "flops_per_frame_executed": FLOPS_EXECUTED_PER_FRAME,
                    "bytes_per_frame": BYTES_PER_FRAME, "flops_per_frame": FLOPS_PER_FRAME,
                    "source": "tools/algorithmic_work.py (config E, all-current mode)", "baseline_md": BASELINE_MD_PER_FRAME},
        }
        if dt_stream is not None:
            line["clip_stream_throughput"] = {
                "value": round(world * a.frames * a.steps / dt_stream, 3), "unit": "frames/s", "ms_per_step": round(dt_stream / a.steps * 1e3, 3),
                "clips_per_step": world, "frames_per_gpu": world * per_rank, "scaling": "weak",
                "note": "same run, after the timed region: N clips per step, every rank runs its T/N frames of each (throughput mode of a "
                        "stream of clips); NOT the headline - BASELINE.json's configuration is one clip sharded over the GPUs"}
        if world == 1 and clips == 1 and not a.no_literal:
            line["literal_mode"] = literal_mode(model, mine)
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(a.height, a.width, min(a.cpu_threads, os.cpu_count() or 1))
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
