"""CPU, 2 processes over gloo: a clip sharded by frames across ranks gives, after the one
all-gather of reference query sets, the same per-frame outputs as one process running the whole clip
(SURVEY.md section 8e).  On the GPU node the same code runs over RCCL (backend "nccl")."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]
PKG = os.path.join(ROOT, "depth-fusion-in-transformer-based-video-object-detection_amd")


def _patch_cpu_ops():
    for p in (PKG, ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    from oracle import msda_oracle
    import models.ops.functions.ms_deform_attn_func as f
    from dfx import ops
    f.MSDeformAttnFunction = msda_oracle.OracleMSDAFunction

    def roi_align(inp, rois, output_size, spatial_scale, sampling_ratio, aligned=True, channels_last=False):
        size = output_size if isinstance(output_size, int) else output_size[0]
        if channels_last:
            out = msda_oracle.roi_align(inp.permute(0, 3, 1, 2).contiguous(), rois, size, spatial_scale,
                                        sampling_ratio, aligned)
            return out.flatten(2).transpose(1, 2).contiguous()
        return msda_oracle.roi_align(inp, rois, size, spatial_scale, sampling_ratio, aligned)

    ops.roi_align = roi_align


def _build():
    from models import build_model
    from models.config import transvodpp_args
    from tests._param_fill import fill_params_by_name
    model, _, _ = build_model(transvodpp_args(num_ref_frames=3, device="cpu"))
    fill_params_by_name(model, seed=5)
    with torch.no_grad():
        for h in list(model.bbox_embed) + list(model.temp_bbox_embed_list):
            h.layers[-1].weight.mul_(0.2)
    return model.eval()


def _worker(rank, world, port, result_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    _patch_cpu_ops()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from models.clip_inference import ClipRunner
        model = _build()
        clip = torch.randn(4, 4, 64, 96, generator=torch.Generator().manual_seed(11))
        per = clip.shape[0] // world
        out = ClipRunner(model, micro_batch=2)(clip[rank * per:(rank + 1) * per])
        gathered = [None] * world
        dist.all_gather_object(gathered, {k: out[k] for k in ("pred_logits", "pred_boxes")})
        if rank == 0:
            torch.save(gathered, result_path)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_clip_matches_single_process(tmp_path, cpu_msda):
    from tests.test_models_golden import cpu_roi  # noqa: F401
    port = _free_port()
    result = str(tmp_path / "sharded.pt")
    mp.spawn(_worker, args=(2, port, result), nprocs=2, join=True)
    sharded = torch.load(result)
    # single process, whole clip
    _patch_cpu_ops()
    from models.clip_inference import ClipRunner
    model = _build()
    clip = torch.randn(4, 4, 64, 96, generator=torch.Generator().manual_seed(11))
    whole = ClipRunner(model, micro_batch=2)(clip)
    logits = torch.cat([s["pred_logits"] for s in sharded], 0)
    boxes = torch.cat([s["pred_boxes"] for s in sharded], 0)
    assert torch.allclose(logits, whole["pred_logits"], atol=1e-5)
    assert torch.allclose(boxes, whole["pred_boxes"], atol=1e-5)
    # identical box indices after post-processing (top-100 over query x class)
    top_a = torch.topk(logits.sigmoid().view(4, -1), 100, dim=1)[1]
    top_b = torch.topk(whole["pred_logits"].sigmoid().view(4, -1), 100, dim=1)[1]
    assert torch.equal(top_a, top_b)


# ---- several clips per call (a stream of clips served B at a time, every clip still sharded over the ranks) -----------
def _clips(n):
    return [torch.randn(4, 4, 64, 96, generator=torch.Generator().manual_seed(21 + c)) for c in range(n)]


def _worker_multi(rank, world, port, result_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    _patch_cpu_ops()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from models.clip_inference import ClipRunner
        model = _build()
        clips = _clips(3)
        per = clips[0].shape[0] // world
        block = torch.cat([c[rank * per:(rank + 1) * per] for c in clips], 0)       # clip-major: [B * F, ...]
        out = ClipRunner(model, micro_batch=3)(block, clips=len(clips))
        gathered = [None] * world
        dist.all_gather_object(gathered, {k: out[k] for k in ("pred_logits", "pred_boxes")})
        if rank == 0:
            torch.save(gathered, result_path)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_three_clips_per_call_on_two_ranks_match_clip_by_clip(tmp_path, cpu_msda):
    """clips=3 on 2 ranks: each rank holds 2 frames of each clip; one all-gather; every frame must come out as when
    its clip is run alone in one process (a frame never sees another clip's queries)."""
    port = _free_port()
    result = str(tmp_path / "multi.pt")
    mp.spawn(_worker_multi, args=(2, port, result), nprocs=2, join=True)
    sharded = torch.load(result)                                    # [rank] -> [B * F, ...] clip-major
    _patch_cpu_ops()
    from models.clip_inference import ClipRunner
    model = _build()
    runner = ClipRunner(model, micro_batch=2)
    for b, clip in enumerate(_clips(3)):
        whole = runner(clip)
        for k in ("pred_logits", "pred_boxes"):
            got = torch.cat([s[k][b * 2:(b + 1) * 2] for s in sharded], 0)         # rank-major = clip order
            assert torch.allclose(got, whole[k], atol=1e-5), (b, k)
        got = torch.cat([s["pred_logits"][b * 2:(b + 1) * 2] for s in sharded], 0)
        assert torch.equal(torch.topk(got.sigmoid().view(4, -1), 100, dim=1)[1],
                           torch.topk(whole["pred_logits"].sigmoid().view(4, -1), 100, dim=1)[1])


def test_clips_per_call_single_process_matches_clip_by_clip(cpu_msda):
    """world = 1: clips=2 is two whole clips in one call."""
    _patch_cpu_ops()
    from models.clip_inference import ClipRunner
    model = _build()
    clips = _clips(2)
    runner = ClipRunner(model, micro_batch=4)
    both = runner(torch.cat(clips, 0), clips=2)
    for b, clip in enumerate(clips):
        whole = runner(clip)
        for k in ("pred_logits", "pred_boxes"):
            assert torch.allclose(both[k][b * 4:(b + 1) * 4], whole[k], atol=1e-5), (b, k)
        for a, w in zip(both["topk"], whole["topk"]):
            assert torch.equal(a[b * 4:(b + 1) * 4], w)
