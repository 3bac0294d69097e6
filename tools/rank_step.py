"""Time ONE rank's share of the 32-frame clip on one GPU, for several (frames per rank, micro-batch,
overlap) choices: frames_forward on the rank's F frames, the all-gather replaced by a local copy of
[T,300,259] (its cost on xGMI is ~10 us), temporal_forward against all T frames' queries.

    python tools/rank_step.py [T=32]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "depth-fusion-in-transformer-based-video-object-detection_amd"))
sys.path.insert(0, ROOT)
from bench import build  # noqa: E402
from models.clip_inference import ClipRunner  # noqa: E402

torch.backends.cuda.matmul.allow_tf32 = False
torch.backends.cudnn.allow_tf32 = False
torch.backends.cudnn.deterministic = os.environ.get("DET", "0") == "1"
T = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda")
model = build(dev, T - 1)
clip = torch.randn(T, 4, 800, 1333, generator=torch.Generator().manual_seed(42))


def rank_step(runner, x):
    local = runner.frames_forward(x)
    F_ = x.shape[0]
    rep = T // F_
    all_ref = local["ref"].repeat(rep, 1, 1)            # stands in for the gathered queries of the other ranks
    all_logits = local["logits"].repeat(rep, 1, 1)
    return runner.temporal_forward(local, all_ref, all_logits, first_frame=0)


def rank_step_weak(runner, x, B):
    """throughput mode: the rank's F frames of each of B clips in one pass (x is [B * F, ...] clip-major)."""
    local = runner.frames_forward(x)
    F_ = x.shape[0] // B
    rep = T // F_

    def pool(t):        # stands in for the gathered queries: every clip's T frames = its own F frames repeated
        return t.view(B, F_, *t.shape[1:]).repeat(1, rep, 1, 1).reshape(B * T, *t.shape[1:])
    return runner.temporal_forward(local, pool(local["ref"]), pool(local["logits"]), first_frame=0, clips=B)


if os.environ.get("WEAK", "1") == "1":
    for world in (2, 4, 8):
        F_ = T // world
        x = torch.cat([clip[:F_]] * world, 0).to(dev)              # B = world clips
        runner = ClipRunner(model, micro_batch=T, overlap=False)
        for _ in range(2):
            rank_step_weak(runner, x, world)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            rank_step_weak(runner, x, world)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 4
        print(f"weak: N={world} GPUs, {world} clips per step, {F_} frames/GPU of each: {dt * 1e3:7.2f} ms/step "
              f"-> {world * T / dt:7.1f} frames/s whole job (no exchange cost)", flush=True)

CASES = os.environ.get('CASES')
for F_, mb, ov in eval(CASES) if CASES else [(32, 8, False), (32, 8, True), (16, 8, False), (16, 8, True), (16, 4, True), (8, 8, False),
                   (8, 4, True), (8, 2, True), (4, 4, False), (4, 2, True), (4, 1, True)]:
    if F_ > T:
        continue
    x = clip[:F_].to(dev)
    runner = ClipRunner(model, micro_batch=mb, overlap=ov)
    for _ in range(2):
        rank_step(runner, x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 4
    for _ in range(n):
        rank_step(runner, x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"frames/rank {F_:2d} (N={T // F_} GPUs)  micro-batch {mb}  overlap {int(ov)}: {dt * 1e3:7.2f} ms/step "
          f"-> {T / dt:7.1f} frames/s whole job", flush=True)

# clips as a stream at the rank's block size (ClipRunner.submit: the tail of step k on a second HIP stream beside the backbones
# of step k + 1) - what `bench.py --pipeline 2` does for N > 1; the exchange is stood in by repeating the rank's own query sets
if os.environ.get("PIPE", "0") == "1":
    # RUNNERS="1,2,3": that many independent ClipRunner pipelines (2 HIP streams each) fed round-robin - more clips in flight
    for R in eval("[" + os.environ.get("RUNNERS", "1") + "]"):
        for F_ in eval(os.environ.get("PIPE_FRAMES", "(4, 8, 16)")):
            x = clip[:F_].to(dev)
            rep = T // F_
            runners = []
            for _ in range(R):
                runner = ClipRunner(model, micro_batch=F_, overlap=False, lanes=int(os.environ.get("LANES", "1")),
                                    graph=os.environ.get("GRAPH", "0") == "1")
                runner.exchange = lambda ref, logits, clips=1, rep=rep: (ref.repeat(rep, 1, 1), logits.repeat(rep, 1, 1))
                runners.append(runner)
            L = int(os.environ.get("LANES", "1"))
            for i in range(3 * R * L):
                runners[i % R].submit(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 8 * R * L
            for i in range(n):
                runners[i % R].submit(x)
            th = (time.perf_counter() - t0) / n
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            mode = "graph" if os.environ.get("GRAPH", "0") == "1" else "eager"
            print(f"pipelined x{R} ({mode}, {L} lane(s) per runner): frames/rank {F_:2d} (N={T // F_} GPUs): {dt * 1e3:7.2f} ms/step -> "
                  f"{T / dt:7.1f} frames/s whole job; host {th * 1e3:5.2f} ms/step in submit", flush=True)
            del runners
