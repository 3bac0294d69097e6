"""GPU, 2 processes sharing cuda:0, gloo transport: the sharded clip path on the HIP kernels gives the
same per-frame outputs as one process running the whole clip.  (RCCL itself needs one GPU per rank;
the 8-GPU run is the driver's.  What is exercised here is everything around the transport: frame
blocks per rank, the packed all-gather message, clip-order reassembly, the batched temporal stage.)"""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "depth-fusion-in-transformer-based-video-object-detection_amd")


def _free_port():
    """A port nobody holds right now (arithmetic on the pid can meet a port another test of this run left in TIME_WAIT)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _build():
    from models import build_model
    from models.config import transvodpp_args
    from tests._param_fill import fill_params_by_name
    model, _, _ = build_model(transvodpp_args(num_ref_frames=3, device="cuda"))
    fill_params_by_name(model, seed=5)
    with torch.no_grad():
        for h in list(model.bbox_embed) + list(model.temp_bbox_embed_list):
            h.layers[-1].weight.mul_(0.2)
    return model.cuda().eval()


def _clip():
    return torch.randn(4, 4, 64, 96, generator=torch.Generator().manual_seed(11))


def _worker(rank, world, port, result_path):
    for p in (PKG, ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from models.clip_inference import ClipRunner
        torch.cuda.set_device(0)
        torch.backends.cudnn.deterministic = True     # library solvers are not run-to-run deterministic
        model = _build()
        clip = _clip()
        per = clip.shape[0] // world
        out = ClipRunner(model, micro_batch=2)(clip[rank * per:(rank + 1) * per].cuda())
        gathered = [None] * world
        dist.all_gather_object(gathered, {k: out[k].cpu() for k in ("pred_logits", "pred_boxes")})
        if rank == 0:
            torch.save(gathered, result_path)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_on_one_gpu_match_single_process(tmp_path):
    port = _free_port()
    result = str(tmp_path / "sharded.pt")
    mp.spawn(_worker, args=(2, port, result), nprocs=2, join=True)
    sharded = torch.load(result)
    from models.clip_inference import ClipRunner
    saved = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    try:
        whole = ClipRunner(_build(), micro_batch=2)(_clip().cuda())
    finally:
        torch.backends.cudnn.deterministic = saved
    logits = torch.cat([s["pred_logits"] for s in sharded], 0)
    boxes = torch.cat([s["pred_boxes"] for s in sharded], 0)
    assert torch.allclose(logits, whole["pred_logits"].cpu(), atol=1e-5)
    assert torch.allclose(boxes, whole["pred_boxes"].cpu(), atol=1e-5)
    top_a = torch.topk(logits.sigmoid().view(4, -1), 100, dim=1)[1]
    top_b = torch.topk(whole["pred_logits"].cpu().sigmoid().view(4, -1), 100, dim=1)[1]
    assert torch.equal(top_a, top_b)


def _worker_full(rank, world, port, result_path, graph):
    """Config E at production size on `world` ranks sharing cuda:0 over gloo: the bench's own pipeline (ClipRunner.submit, two
    lanes or HIP graphs, the exchange on the lane's stream) on three different 32-frame 800x1333 clips."""
    for p in (PKG, ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        from models.clip_inference import ClipRunner
        torch.cuda.set_device(0)
        T = 32
        per = T // world
        model = bench.build(torch.device("cuda", 0), T - 1)
        runner = ClipRunner(model, micro_batch=per, lanes=3 if graph else 2, graph=graph)
        outs = []
        for c in range(3):
            clip = torch.randn(T, 4, 800, 1333, generator=torch.Generator().manual_seed(42 + c))
            outs.append(runner.submit(clip[rank * per:(rank + 1) * per].cuda()))
        res = []
        for out, done in outs:
            done.synchronize()
            res.append({k: out[k].cpu() for k in ("pred_logits", "pred_boxes")})
        gathered = [None] * world
        dist.all_gather_object(gathered, res)
        if rank == 0:
            torch.save(gathered, result_path)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("graph", [False, True])
def test_config_e_sharded_over_two_ranks_at_production_size(tmp_path, graph):
    """BASELINE.json configs[4] as the N > 1 bench runs it - 32-frame 800x1333 RGB-D clips, 16 frames per rank, one all-gather of
    the reference query sets per clip, the clip pipeline with two eager lanes / three HIP-graph lanes - on two ranks over gloo
    (all this box allows: RCCL needs one GPU per rank), against one process running the whole clip: every frame's logits and
    boxes within 2e-5 (the GEMM tiling depends on the rows per launch), for three clips in flight."""
    import bench
    from models.clip_inference import ClipRunner
    port = _free_port()
    result = str(tmp_path / "sharded_full.pt")
    mp.spawn(_worker_full, args=(2, port, result, graph), nprocs=2, join=True)
    sharded = torch.load(result)                      # [rank][clip]{...}
    model = bench.build(torch.device("cuda", 0), 31)
    runner = ClipRunner(model, micro_batch=32)
    for c in range(3):
        clip = torch.randn(32, 4, 800, 1333, generator=torch.Generator().manual_seed(42 + c))
        whole = runner(clip.cuda())
        for k in ("pred_logits", "pred_boxes"):
            got = torch.cat([sharded[r][c][k] for r in range(2)], 0)
            assert got.shape == whole[k].shape
            err = (got - whole[k].cpu()).abs().max().item()
            assert err < 2e-5, (c, k, err)


def test_two_stream_schedule_gives_identical_outputs():
    """overlap=True (backbones of micro-batch i+1 beside the transformer tail of micro-batch i on two HIP
    streams) only changes the schedule: same kernels, same inputs -> the same bits, run after run."""
    from models.clip_inference import ClipRunner
    model = _build()
    clip = torch.randn(6, 4, 64, 96, generator=torch.Generator().manual_seed(12)).cuda()
    # library solvers for some small convolutions are not run-to-run deterministic on this stack
    # (tools/determinism_check.py); ask for deterministic ones so that "same bits" can be asserted
    saved = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    try:
        serial = ClipRunner(model, micro_batch=2, overlap=False)(clip)
        again = ClipRunner(model, micro_batch=2, overlap=False)(clip)
        assert torch.equal(serial["pred_logits"], again["pred_logits"]), "serial schedule is not deterministic"
        runner = ClipRunner(model, micro_batch=2, overlap=True)
        for _ in range(3):                      # repeated calls reuse the streams and the allocator's blocks
            piped = runner(clip)
            torch.cuda.synchronize()
            for k in ("pred_logits", "pred_boxes"):
                assert torch.equal(piped[k], serial[k]), k
            for a, b in zip(piped["topk"], serial["topk"]):
                assert torch.equal(a, b)
    finally:
        torch.backends.cudnn.deterministic = saved


@pytest.mark.parametrize("lanes,graph", [(1, False), (2, False), (3, False), (1, True), (3, True)])
def test_clip_pipeline_submit_matches_call(lanes, graph):
    """ClipRunner.submit (backbones of clip k+1 on one HIP stream beside the tail of clip k on another, at
    most two clips in flight per lane; ``lanes`` stream pairs dealt round-robin; ``graph``: every step a HIP-graph replay)
    returns what __call__ returns, for a stream of different clips - bit for bit."""
    from models.clip_inference import ClipRunner
    model = _build()
    clips = [torch.randn(4, 4, 64, 96, generator=torch.Generator().manual_seed(40 + i)).cuda() for i in range(9)]
    saved = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True         # see test_two_stream_schedule_gives_identical_outputs
    try:
        runner = ClipRunner(model, micro_batch=4, lanes=lanes, graph=graph)
        want = [runner(c) for c in clips]
        torch.cuda.synchronize()
        handles = [runner.submit(c) for c in clips]   # queued back to back, nothing waited for in between
        for (out, done), ref in zip(handles, want):
            done.synchronize()
            for k in ("pred_logits", "pred_boxes"):
                assert torch.equal(out[k], ref[k]), k
            for a, b in zip(out["topk"], ref["topk"]):
                assert torch.equal(a, b)
    finally:
        torch.backends.cudnn.deterministic = saved


@pytest.mark.timeout(900)
@pytest.mark.parametrize("clips_per_step,pipeline,launcher", [(None, 1, False), (0, 1, True), (1, 0, True)])
def test_bench_py_two_ranks_gloo_rehearsal(tmp_path, clips_per_step, pipeline, launcher):
    """bench.py's own N > 1 code path (rank / world from the environment, frame sharding, barrier, MAX over ranks,
    rank 0 prints the line) with two ranks sharing this GPU over gloo - a rehearsal of the driver's RCCL launch on a
    small clip.  The line must parse and describe a 2-rank run: by default ONE clip in flight sharded over the ranks
    (BASELINE.json's configuration, strong scaling) plus the N-clips-per-step throughput figure as an extra key;
    --clips-per-step 0 makes the throughput mode the measured one (weak scaling); by default the exchange sits on the
    side stream of the clip pipeline (ClipRunner.submit), --pipeline 0 runs one clip at a time on one stream.
    ``launcher=False`` is the driver's N = 1 command form with N changed - plain ``python bench.py --gpus 2`` - which must
    start its two ranks by itself (bench.py:launch_ranks) and give the same line."""
    import json
    import subprocess
    port = _free_port()
    cmd = [sys.executable]
    if launcher:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(port)]
    cmd += [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--frames", "4",
           "--height", "128", "--width", "160", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--pipeline", str(pipeline)]
    if clips_per_step is not None:
        cmd += ["--clips-per-step", str(clips_per_step)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=850, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["value"] > 0
    assert line["rccl"] == {"backend": "gloo", "world": 2}
    assert line["config"]["frames_per_gpu_per_clip"] == 2 and line["roofline"]["bound"] == "mfma"
    # 2 frames per rank and step (<= 4): the pipelined steps are HIP-graph replays on 4 lanes (bench.py asks the HIP runtime
    # for 8 hardware queues), the exchange between the two graphs of a step goes over the process group eagerly; the
    # N-clips-per-step mode (4 frames per rank) too
    graphs = pipeline >= 1
    assert line["config"]["hip_hw_queues"] == int(os.environ.get("GPU_MAX_HW_QUEUES", "8"))
    assert line["config"]["clip_pipeline"] == (pipeline >= 1)
    assert line["config"]["pipeline_lanes"] == ((4 if line["config"]["hip_hw_queues"] >= 8 else 3) if pipeline else 0)
    assert line["config"]["hip_graphs"] == graphs
    assert line["ms_per_step_p50"] > 0
    if clips_per_step == 0:
        assert line["scaling"] == "weak" and line["config"]["clips_per_step"] == 2 and line["config"]["frames_per_gpu"] == 4
        assert "clip_stream_throughput" not in line
    else:
        assert line["scaling"] == "strong" and line["config"]["clips_per_step"] == 1 and line["config"]["frames_per_gpu"] == 2
        extra = line["clip_stream_throughput"]
        assert extra["value"] > 0 and extra["frames_per_gpu"] == 4 and extra["scaling"] == "weak"


def test_clips_per_call_match_clip_by_clip_on_the_hip_path():
    """clips=2 in one call (the rank's frames of both clips in one pass through every kernel) gives each clip the
    outputs it gets alone: bit-equal picks, floating outputs to rounding (batch size changes the GEMM tiling only)."""
    from models.clip_inference import ClipRunner
    model = _build()
    clips = [torch.randn(4, 4, 64, 96, generator=torch.Generator().manual_seed(60 + i)).cuda() for i in range(2)]
    runner = ClipRunner(model, micro_batch=8)
    both = runner(torch.cat(clips, 0), clips=2)
    for b, clip in enumerate(clips):
        alone = runner(clip)
        for k in ("pred_logits", "pred_boxes"):
            assert torch.allclose(both[k][b * 4:(b + 1) * 4], alone[k], atol=1e-5), (b, k)


def test_clip_outputs_are_run_to_run_deterministic():
    """No library solver is left in the inference path (round 2), and no kernel of it accumulates with atomics: the same
    clip gives the same bits run after run, default settings (no cudnn.deterministic)."""
    from models.clip_inference import ClipRunner
    saved = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = False
    try:
        runner = ClipRunner(_build(), micro_batch=4)
        clip = _clip().cuda()
        first = runner(clip)
        for _ in range(3):
            again = runner(clip)
            assert torch.equal(first["pred_logits"], again["pred_logits"]) and torch.equal(first["pred_boxes"], again["pred_boxes"])
            for a, b in zip(first["topk"], again["topk"]):
                assert torch.equal(a, b)
    finally:
        torch.backends.cudnn.deterministic = saved


_RCCL_ONE_RANK = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["DFX_PKG"]); sys.path.insert(0, os.environ["DFX_ROOT"])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ["DFX_PORT"], RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
from tests.test_clip_shard_gpu import _build, _clip
from models.clip_inference import ClipRunner
model = _build()
clip = _clip().cuda()
want = ClipRunner(model, micro_batch=4)(clip)                       # no process group: no exchange
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))  # RCCL
runner = ClipRunner(model, micro_batch=4, gather_on_one_rank=True)   # run the collective although world == 1
got = runner(clip)                                                   # all_gather_into_tensor on the current stream
outs = [runner.submit(clip) for _ in range(3)]                       # ... and on the side stream of the clip pipeline
calls = []
class Spy(ClipRunner):                                               # which stream every collective runs on, and what was captured by then
    def exchange(self, ref, logits, clips=1):
        calls.append((torch.cuda.current_stream().stream_id, {k[0] for k in self._graph_slots}))
        return super().exchange(ref, logits, clips)
grunner = Spy(model, micro_batch=4, gather_on_one_rank=True, lanes=3, graph=True)
outs += [grunner.submit(clip) for _ in range(7)]                     # ... and between the two HIP graphs of a step, three lanes
torch.cuda.synchronize()
dist.barrier()
assert len(grunner._graph_slots) == 3 and all(isinstance(v, dict) and "g2" in v for v in grunner._graph_slots.values()), "captures fell back"
# the watchdog polls a collective's completion event; HIP refuses that while the event's stream captures: a lane's stream
# carries no collective before the lane has captured (the communicator is set up on the caller's stream)
lanes = {s.stream_id: k for k, s in grunner._streams.items() if isinstance(k, tuple) and isinstance(k[0], str) and isinstance(k[1], int)}
assert len(lanes) == 3 and sum(sid in lanes for sid, _ in calls) == 7 and any(sid not in lanes for sid, _ in calls)
assert all(lanes[sid] in captured for sid, captured in calls if sid in lanes), "a collective ran on a lane's stream before the lane captured"
for o in [got] + [o for o, _ in outs]:
    for k in ("pred_logits", "pred_boxes"):
        assert torch.equal(o[k], want[k]), k
dist.destroy_process_group()
print("RCCL_ONE_RANK_OK")
"""


@pytest.mark.timeout(600)
def test_rccl_call_path_with_one_rank():
    """The exchange through RCCL itself (backend "nccl") with ONE rank - all this box allows: process-group initialisation on
    the device, `all_gather_into_tensor` of the packed query sets on the current stream, on the side stream of the clip
    pipeline (`ClipRunner.submit`) and between the two HIP graphs of a graph-mode step (captures with the backend's watchdog
    thread alive: thread-local capture mode), a barrier, bit-equal outputs.  What it cannot show is the transport between GPUs."""
    import subprocess
    env = dict(os.environ, DFX_PKG=PKG, DFX_ROOT=ROOT, DFX_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", _RCCL_ONE_RANK], capture_output=True, text=True, timeout=550, env=env)
    said = [ln for ln in out.stderr.splitlines() if any(w in ln for w in ("what()", "xception", "rror", "terminate"))][:12]
    assert out.returncode == 0 and "RCCL_ONE_RANK_OK" in out.stdout, (out.returncode, out.stdout[-300:], said, out.stderr[-1500:])


@pytest.mark.timeout(600)
@pytest.mark.parametrize("H,W", [(64, 96), (800, 1333)])
def test_literal_forward_is_frame_0_of_the_all_current_clip(H, W):
    """The two modes of SURVEY.md 8d on one clip: the detector's own forward (T frames in, frame 0 current, the others its
    references in clip order, ONE output: bench.py's secondary `literal_mode` figure) is what all-current mode - the headline -
    emits for frame 0.  Both run the HIP path; the rows per GEMM launch differ (micro-batching), hence a tolerance."""
    import bench
    from models.clip_inference import ClipRunner
    from util.misc_multi import NestedTensor
    T = 4
    dev = torch.device("cuda", 0)
    model = bench.build(dev, T - 1)
    clip = torch.randn(T, 4, H, W, generator=torch.Generator().manual_seed(5)).to(dev)
    with torch.no_grad():
        one = model(NestedTensor(clip, torch.zeros(T, H, W, dtype=torch.bool, device=dev)))
    every = ClipRunner(model, micro_batch=T)(clip)
    assert one["pred_logits"].shape[0] == 1 and every["pred_logits"].shape[0] == T
    assert (one["pred_logits"][0] - every["pred_logits"][0]).abs().max().item() < 2e-5
    assert (one["pred_boxes"][0] - every["pred_boxes"][0]).abs().max().item() < 2e-5
    rep = bench.literal_mode(model, clip, warm=1, iters=2)
    assert rep["iterations"] == 2 and rep["input_frames_per_s"] == pytest.approx(T * rep["clips_per_s"], rel=1e-2)
