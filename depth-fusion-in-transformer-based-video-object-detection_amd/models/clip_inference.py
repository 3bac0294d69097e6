"""Clip inference for TransVOD++ with every frame of the clip as "current" frame, optionally with
the clip's frames spread over several GPUs (one process per GPU, RCCL over xGMI).

The reference forward consumes a clip of T = 1+R frames and emits ONE output, for frame 0
(/root/reference/models/deformable_transformer_multi_plusplus.py:404-406,601).  Running it T times
with each frame in front would redo the whole spatial stage T times.  This runner computes, for
every frame t, exactly what the reference forward would output for the clip re-ordered as
[t, other frames in clip order] - but each frame's spatial stage (backbones, Late Fusion, encoder,
decoder) and its two query/RoI fusion passes run ONCE:

  1. spatial stage on the rank's own frames (micro-batched);
  2. ``frame_stage``: class logits + both RoI-fused query sets ("cur": plain memory, "ref":
     memory + positional embedding - the reference treats the two roles differently);
  3. one all-gather of the "ref" query sets [Q,256] and class logits [Q,classes] of every frame
     (the only inter-GPU exchange: ~0.31 MB per frame);
  4. ``temporal_stage`` for each owned frame against all other frames' gathered queries, then the
     final temporal heads.

Frames are independent until step 3 (frozen / eval-mode norms, per-sample attention), so the
clip shards by contiguous blocks of frames with weights replicated.

Several clips per call (``clips=B``): a rank that owns only T/world frames of a clip runs small kernels
(4 frames per GPU at world = 8).  A stream of clips can be served B at a time instead: the rank's block is then
its F frames of EACH of the B clips, [B*F, C, H, W] clip-major, steps 1+2 run on all B*F frames in one pass, step 3
is still ONE all-gather (of all B clips' query sets), and step 4 lets every frame see the T-1 other frames of ITS OWN
clip only.  Per clip the results are those of ``clips=1``.
"""
import torch
import torch.distributed as dist

from util.misc import NestedTensor

from .detector_common import apply_box_head
from .fused import enable_fused_inference


class ClipRunner:
    MIN_OVERLAP_BATCHES = 3

    def __init__(self, model, micro_batch=4, group=None, fused=None, overlap=True, gather_on_one_rank=False, lanes=1, graph=False):
        """model: models.deformable_detr_multi_plusplus.DeformableDETR in eval mode.
        fused: use the GPU-only fused inference routes (models/fused.py); default = model is on a GPU.
        overlap: on a GPU, run the backbones of micro-batch i+1 on one HIP stream while the transformer
        and the query/RoI fusion of micro-batch i (many short kernels that leave most CUs idle) run on
        another; results are identical, only the schedule changes.  Used when the block has at least
        MIN_OVERLAP_BATCHES micro-batches (measured, tools/rank_step.py: 4 micro-batches of 8 frames
        143.3 -> 133.1 ms, 2 micro-batches 72.8 -> 74.1 ms).
        gather_on_one_rank: run the collective of ``exchange`` although the group has one rank (RCCL's call path on a
        one-GPU box); a one-rank group skips it otherwise.
        lanes: independent stream pairs ``submit`` deals consecutive clips to, round-robin (each lane keeps two clips in
        flight).  A rank that owns few frames of a clip runs kernels that leave CUs idle (partial last rounds of tiles,
        the 300-query tail): with two lanes the 4-frame rank step of an 8-GPU run takes 11.6 ms instead of 12.8 (14.6 on one
        stream), the 8-frame step 21.7 instead of 25.2 (profiles/r04_rank_step.txt); three lanes add nothing, and neither do
        two at 32 frames per rank.
        graph: ``submit`` replays the rank step as HIP graphs (one per lane and input shape: the spatial stage + query/RoI
        fusion, then - after the eager exchange when the clip is sharded - the temporal stage) instead of launching its ~1500
        kernels from Python: the same kernels and bit-equal outputs, the host side of a 4-frame step drops from ~9 ms to
        well under 1 ms (8 ranks share one host).  A lane is then ONE stream with one clip in flight; use 3 lanes, or 4 with
        GPU_MAX_HW_QUEUES=8 (the HIP runtime maps a process's streams onto 4 hardware queues by default: INTEGRATION.md).  Masked
        (padded) clips take the eager route."""
        self.model = model
        self.gather_on_one_rank = gather_on_one_rank
        self.lanes = max(1, int(lanes))
        self.graph = bool(graph)
        self._graph_slots = {}
        self._lane_carried_collectives = set()
        self._next_lane = 0
        self._lane_warm = set()
        self.micro_batch = micro_batch
        self.group = group
        self.overlap = overlap
        self._streams = {}
        if fused is None:
            fused = next(model.parameters()).is_cuda
        enable_fused_inference(model, fused)
        self._const = {}       # small index tensors, built once (each build is a blocking H2D copy)

    def _cached(self, key, build):
        if key not in self._const:
            self._const[key] = build()
        return self._const[key]

    # ---- steps 1+2 for a block of frames -------------------------------------------------------
    # the two halves of steps 1+2 for ALL frames of the block in one pass (what ``submit`` pipelines)
    def _block_consts(self, frames, mask):
        m = self.model
        F_, _, H, W = frames.shape
        dev = str(frames.device)
        if mask is None:        # the runner's own all-valid mask: the same tensor object on every call (util/memo.py)
            mask = self._cached(("mask", F_, H, W, dev), lambda: torch.zeros((F_, H, W), dtype=torch.bool, device=frames.device))
        whwh = self._cached(("whwh", W, H, dev), lambda: torch.as_tensor(
            (W, H, W, H), dtype=torch.long, device=frames.device).repeat(1, m.num_queries, 1))
        return mask, whwh

    def _encode_block(self, frames, mask=None):
        mask, whwh = self._block_consts(frames, mask)
        return self.model._encode_inputs(NestedTensor(frames, mask)), whwh

    def _tail_block(self, staged):
        m, tr = self.model, self.model.transformer
        (srcs, masks, pos, d_srcs, d_masks, d_pos, rgbd), whwh = staged
        st = tr._spatial_stage(srcs, masks, pos, d_srcs, d_masks, d_pos, m.query_embed.weight, rgbd)
        fs = tr.frame_stage(st["hs"][-1], st["inter_references"][-1], st["memory"], st["lvl_pos_embed_flatten"],
                            st["last_hw"], whwh, m.class_embed[-1], m.bbox_embed[-1], roles=("cur", "ref"))
        return {"cur": fs["cur"], "ref": fs["ref"], "logits": fs["logits"], "ref_last": st["inter_references"][-1],
                "hs_last": st["hs"][-1], "memory": st["memory"], "valid_ratios": st["valid_ratios"], "spatial_shapes": st["spatial_shapes"],
                "level_start_index": st["level_start_index"]}

    @torch.no_grad()
    def frames_forward(self, frames, mask=None):
        """frames [F,4|3,H,W] (this rank's frames) -> dict of per-frame tensors."""
        m, tr = self.model, self.model.transformer
        F_, _, H, W = frames.shape
        own_mask = mask is None
        if own_mask:
            mask = self._cached(("mask", F_, H, W, str(frames.device)),
                                lambda: torch.zeros((F_, H, W), dtype=torch.bool, device=frames.device))
        whwh = self._cached(("whwh", W, H, str(frames.device)), lambda: torch.as_tensor(
            (W, H, W, H), dtype=torch.long, device=frames.device).repeat(1, m.num_queries, 1))
        keep = {k: [] for k in ("cur", "ref", "logits", "ref_last", "hs_last", "memory", "valid_ratios")}
        meta = []

        def encode(sl):
            # the runner's own all-valid mask is handed over as the SAME tensor object on every call, so
            # everything derived from it (resized masks, positional embeddings, valid ratios) is built once
            # (util/memo.py); a caller's mask is passed through as it comes
            msl = mask[sl] if not own_mask else self._cached(
                ("mask_slice", F_, H, W, str(frames.device), sl.start, sl.stop), lambda: mask[sl])
            return m._encode_inputs(NestedTensor(frames[sl], msl))

        def tail(enc):
            srcs, masks, pos, d_srcs, d_masks, d_pos, rgbd = enc
            st = tr._spatial_stage(srcs, masks, pos, d_srcs, d_masks, d_pos, m.query_embed.weight, rgbd)
            fs = tr.frame_stage(st["hs"][-1], st["inter_references"][-1], st["memory"],
                                st["lvl_pos_embed_flatten"], st["last_hw"], whwh, m.class_embed[-1],
                                m.bbox_embed[-1], roles=("cur", "ref"))
            keep["cur"].append(fs["cur"])
            keep["ref"].append(fs["ref"])
            keep["logits"].append(fs["logits"])
            keep["ref_last"].append(st["inter_references"][-1])
            keep["hs_last"].append(st["hs"][-1])
            keep["memory"].append(st["memory"])
            keep["valid_ratios"].append(st["valid_ratios"])
            meta.append((st["spatial_shapes"], st["level_start_index"]))

        slices = [slice(s, min(F_, s + self.micro_batch)) for s in range(0, F_, self.micro_batch)]
        overlapped = self.overlap and frames.is_cuda and len(slices) >= self.MIN_OVERLAP_BATCHES
        if overlapped:
            # two streams: backbones of micro-batch i+1 alongside the transformer tail of micro-batch i.
            # Every tensor that crosses streams stays referenced (held / keep) until both streams have
            # been joined below, so the caching allocator cannot hand its memory out early.
            cur = torch.cuda.current_stream(frames.device)
            if frames.device not in self._streams:
                self._streams[frames.device] = (torch.cuda.Stream(frames.device), torch.cuda.Stream(frames.device))
            s_back, s_tail = self._streams[frames.device]
            s_back.wait_stream(cur)
            s_tail.wait_stream(cur)
            held = []

            def encode_async(sl):
                with torch.cuda.stream(s_back):
                    enc = encode(sl)
                    ev = torch.cuda.Event()
                    ev.record(s_back)
                return enc, ev

            pending = encode_async(slices[0])
            for i in range(len(slices)):
                enc, ev = pending
                if i + 1 < len(slices):
                    pending = encode_async(slices[i + 1])      # queued first: the GPU always has large kernels waiting
                with torch.cuda.stream(s_tail):
                    s_tail.wait_event(ev)
                    tail(enc)
                held.append(enc)
            cur.wait_stream(s_back)
            cur.wait_stream(s_tail)
        else:
            for sl in slices:
                tail(encode(sl))
        meta = meta[-1]
        out = {k: (v[0] if len(v) == 1 else torch.cat(v, 0)) for k, v in keep.items()}      # (cat of ONE tensor is a copy)
        out["spatial_shapes"], out["level_start_index"] = meta
        if overlapped:
            out["_held"] = held       # keeps the cross-stream tensors alive until the caller drops the result
        return out

    # ---- step 3 ---------------------------------------------------------------------------------
    def exchange(self, ref, logits, clips=1):
        """all-gather the per-frame reference query sets and logits over the clip's ranks.
        ref [B*F,Q,C], logits [B*F,Q,K] (clip-major: the rank's F frames of clip 0, of clip 1, ...) ->
        ([B*T,Q,C], [B*T,Q,K]), clip-major with each clip's T = world * F frames in clip order (rank-major)."""
        if not (dist.is_available() and dist.is_initialized()):
            return ref, logits
        world = dist.get_world_size(self.group)
        if world == 1 and not self.gather_on_one_rank:
            return ref, logits
        BF, Q, C = ref.shape
        assert BF % clips == 0, "the rank's block must hold the same number of frames of every clip"
        F_ = BF // clips
        K = logits.shape[-1]
        packed = torch.cat([ref, logits], dim=-1).contiguous()            # one message per rank
        gathered = torch.empty((world * BF, Q, C + K), dtype=packed.dtype, device=packed.device)
        dist.all_gather_into_tensor(gathered, packed, group=self.group)
        if clips > 1:      # [world, B, F] -> [B, world, F]: every clip's frames together, in clip order
            gathered = gathered.view(world, clips, F_, Q, C + K).transpose(0, 1).reshape(world * BF, Q, C + K)
        return gathered[..., :C].contiguous(), gathered[..., C:].contiguous()

    # ---- step 4 ---------------------------------------------------------------------------------
    @torch.no_grad()
    def temporal_forward(self, local, all_ref, all_logits, first_frame, clips=1, others=None):
        """Outputs for each local frame as the current frame, all local frames in one batched pass.
        local: the rank's B*F frames (clip-major); all_ref / all_logits [B*T,...] from ``exchange``; local frame i of
        clip b is frame first_frame + i of that clip and sees the clip's other T-1 frames - or, with ``others`` [F,R]
        (long, rows of the pools), the reference frames the caller names, in that order (``VideoStream``)."""
        m, tr = self.model, self.model.transformer
        assert all_ref.shape[0] % clips == 0 and local["cur"].shape[0] % clips == 0
        T = all_ref.shape[0] // clips
        F_ = local["cur"].shape[0] // clips
        dev = all_ref.device
        if others is None:
            others = self._cached(("others", T, first_frame, F_, clips, str(dev)), lambda: torch.as_tensor(
                [[b * T + j for j in range(T) if j != first_frame + i] for b in range(clips) for i in range(F_)],
                dtype=torch.long, device=dev))                                     # [B*F, T-1] rows of the pools, clip order
        final_hs, final_refs, _, picks = tr.temporal_stage(
            local["cur"], local["ref_last"], local["memory"], all_ref, all_logits, others,
            local["spatial_shapes"], local["level_start_index"], local["valid_ratios"],
            m.temp_class_embed_list, m.temp_bbox_embed_list)
        return {"pred_logits": m.temp_class_embed_list[2](final_hs),
                "pred_boxes": apply_box_head(m.temp_bbox_embed_list[2], final_hs, final_refs), "topk": picks,
                "topk_scores": tr.last_pick_scores, "final_hs": final_hs}

    @torch.no_grad()
    def __call__(self, frames, mask=None, clips=1):
        """frames: this rank's contiguous block of the clip, [T/world, C, H, W] - or, with ``clips=B``, its block of
        each of B clips, [B * T/world, C, H, W] clip-major.
        -> {"pred_logits" [B*F,Q,classes], "pred_boxes" [B*F,Q,4]} for the rank's frames."""
        rank = dist.get_rank(self.group) if dist.is_available() and dist.is_initialized() else 0
        assert frames.shape[0] % clips == 0, "the block must hold the same number of frames of every clip"
        local = self.frames_forward(frames, mask)
        all_ref, all_logits = self.exchange(local["ref"], local["logits"], clips)
        return self.temporal_forward(local, all_ref, all_logits, first_frame=rank * (frames.shape[0] // clips), clips=clips)

    # ---- a stream of clips ------------------------------------------------------------------------
    @torch.no_grad()
    def submit(self, frames, mask=None, clips=1):
        """Pipelined ``__call__`` for a stream of clips (serving): this clip's backbones are queued on one
        HIP stream, its transformer / query-RoI fusion / exchange / temporal stage on a second one, so
        the tail of clip k - many short kernels that leave most CUs idle - runs beside the backbones of
        clip k+1.  Same kernels, same results; only the schedule changes.

        -> (outputs, done): ``outputs`` as ``__call__``; they are valid once ``done`` (a HIP event) has
        completed - ``done.synchronize()``, ``torch.cuda.current_stream().wait_event(done)`` or a device
        synchronize.  At most two clips are in flight per lane (``lanes`` of the constructor: consecutive clips go to the
        lanes round-robin); a further submit on a full lane waits for that lane's oldest clip."""
        if not frames.is_cuda:
            out = self(frames, mask, clips)
            return out, None
        if self.graph and mask is None:
            return self._submit_graph(frames, clips)
        dev = frames.device
        rank = dist.get_rank(self.group) if dist.is_available() and dist.is_initialized() else 0
        lane = self._next_lane % self.lanes
        self._next_lane += 1
        key = (dev, lane)
        if key not in self._streams:
            # (the first lane is also filed under the device alone: the pair frames_forward's two-stream schedule uses)
            self._streams[key] = self._streams[dev] if lane == 0 and dev in self._streams else (torch.cuda.Stream(dev), torch.cuda.Stream(dev))
            if lane == 0:
                self._streams[dev] = self._streams[key]
        s_back, s_tail = self._streams[key]
        inflight = self.__dict__.setdefault("_inflight", {}).setdefault(key, [])
        while inflight and inflight[0][-1].query():
            inflight.pop(0)
        if len(inflight) >= 2:
            inflight.pop(0)[-1].synchronize()
        s_back.wait_stream(torch.cuda.current_stream(dev))           # the caller's frames
        saved_overlap, self.overlap = self.overlap, False             # one schedule at a time
        try:
            with torch.cuda.stream(s_back):
                staged = self._encode_block(frames, mask)
                ready = torch.cuda.Event()
                ready.record(s_back)
            with torch.cuda.stream(s_tail):
                s_tail.wait_event(ready)
                local = self._tail_block(staged)
                all_ref, all_logits = self.exchange(local["ref"], local["logits"], clips)
                out = self.temporal_forward(local, all_ref, all_logits, first_frame=rank * (frames.shape[0] // clips),
                                            clips=clips)
                done = torch.cuda.Event(enable_timing=True)
                done.record(s_tail)
            # the outputs were allocated on the tail stream and are consumed on the caller's (after ``done``): tell the caching
            # allocator, so that a block the caller frees is not handed out on the tail stream while the caller's kernels still read it
            cur = torch.cuda.current_stream(dev)
            for v in out.values():
                for t in (v if isinstance(v, (list, tuple)) else (v,)):
                    if torch.is_tensor(t):
                        t.record_stream(cur)
        finally:
            self.overlap = saved_overlap
        inflight.append((frames, staged, local, out, done))           # cross-stream tensors stay referenced until done
        if key not in self._lane_warm:
            # a lane's first clip completes before the next lane starts: the small tensors built once and shared by all lanes
            # (index tables, mask-derived positions: self._cached, util/memo.py) are then complete for every later reader
            self._lane_warm.add(key)
            done.synchronize()
        return out, done


    # ---- the same as HIP graphs ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def _submit_graph(self, frames, clips=1):
        dev = frames.device
        sharded = dist.is_available() and dist.is_initialized() and (dist.get_world_size(self.group) > 1 or self.gather_on_one_rank)
        sharded = sharded or "exchange" in self.__dict__          # (an instance-level stand-in for the exchange: tools/rank_step.py)
        rank = dist.get_rank(self.group) if dist.is_available() and dist.is_initialized() else 0
        first = rank * (frames.shape[0] // clips)
        lane = self._next_lane % self.lanes
        self._next_lane += 1
        skey = (str(dev), lane)
        if skey not in self._streams:
            self._streams[skey] = torch.cuda.Stream(dev)
        stream = self._streams[skey]
        key = (skey, tuple(frames.shape), frames.dtype, clips, sharded)
        slot = self._graph_slots.get(key)
        cur = torch.cuda.current_stream(dev)
        saved_overlap, self.overlap = self.overlap, False
        try:
            if slot is None:
                slot = self._capture_slot(frames, clips, first, sharded, stream, cur, skey in self._lane_carried_collectives)
                self._graph_slots[key] = slot
            if slot is False:                                             # capture failed once (reported then): eager route
                self.overlap = saved_overlap
                saved_graph, self.graph = self.graph, False
                self._next_lane -= 1
                try:
                    return self.submit(frames, None, clips)
                finally:
                    self.graph = saved_graph
            stream.wait_stream(cur)                                       # the caller's frames
            with torch.cuda.stream(stream):
                slot["in"].copy_(frames, non_blocking=True)
                slot["g1"].replay()
                if sharded:
                    ar, al = self.exchange(slot["local"]["ref"], slot["local"]["logits"], clips)
                    self._lane_carried_collectives.add(skey)
                    slot["ref"].copy_(ar)
                    slot["logits"].copy_(al)
                    slot["g2"].replay()
                # the graph's own output buffers are overwritten by the lane's next clip: hand out copies
                o = slot["out"]
                out = {k: (v.clone() if torch.is_tensor(v) else [t.clone() for t in v]) for k, v in o.items()}
                done = torch.cuda.Event(enable_timing=True)
                done.record(stream)
            for v in out.values():                                        # (as in ``submit``: consumed on the caller's stream)
                for t in (v if isinstance(v, (list, tuple)) else (v,)):
                    if torch.is_tensor(t):
                        t.record_stream(cur)
        finally:
            self.overlap = saved_overlap
        hold = self.__dict__.setdefault("_ginflight", {}).setdefault(skey, [])
        hold.append((frames, done))                                       # the input stays referenced until its copy has run
        while len(hold) > 2:
            hold.pop(0)
        return out, done


    def _pools_stand_in(self, ref, logits, clips):
        """Tensors with the shapes (and clip-major order) of ``exchange``'s results, WITHOUT a collective: the rank's own rows
        repeated.  For the eager passes that precede a capture."""
        if "exchange" in self.__dict__:                       # (an instance-level stand-in that launches no collective either)
            return self.exchange(ref, logits, clips)
        world = dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1
        if world == 1:
            return ref, logits
        F_ = ref.shape[0] // clips

        def rep(t):
            return t.view(clips, F_, *t.shape[1:]).repeat(1, world, 1, 1).reshape(world * t.shape[0], *t.shape[1:])
        return rep(ref), rep(logits)

    def _capture_slot(self, frames, clips, first, sharded, stream, cur, lane_carried_collectives=False):
        """Graphs of one (lane, input shape): eager passes on the lane's stream first - every table built once (self._cached,
        util/memo.py), every kernel's one-time set-up done: a capture must not meet a blocking copy - then the capture(s), into
        one memory pool.  Other threads (the collective backend's watchdog) may call the runtime meanwhile: thread-local
        capture mode.  -> the slot, or False when the capture failed (said once on stderr; the caller runs eagerly).

        The eager passes launch NO collective (``_pools_stand_in`` gives the temporal stage pools of the right shape; the
        communicator is set up by one tiny exchange on the caller's stream).  torch.distributed's NCCL backend records a
        completion event per collective on the stream it runs on (the lane's, for the blocking all-gather of ``exchange``) and
        its watchdog thread polls that event until it has seen the collective finish - up to ~100 ms later; the HIP runtime
        refuses the query of an event whose stream is capturing (hipErrorCapturedEvent: the capture is lost and the watchdog
        takes the process down - one run in five of the one-rank RCCL test while the eager passes still ran the exchange).  So a
        lane's stream has carried no collective when it captures.  A lane that captures a SECOND input shape has carried those
        of its replays: the device is drained and the watchdog given time to retire them.
        (Measured and dropped: the captures, or the eager passes, on a further stream shared by the lanes - the lanes' graphs then
        partly serialise, 13.0 instead of 11.5 ms per 4-frame step; the exchange of a replay on a stream beside the lane: 12.6.)"""
        import sys
        import time as _time
        slot, err = None, None
        try:
            if sharded and not getattr(self, "_communicator_ready", False) and "exchange" not in self.__dict__:
                tiny = torch.zeros((clips, 1, 1), dtype=frames.dtype, device=frames.device)
                self.exchange(tiny, tiny, clips)               # (on the caller's stream, which is never captured here)
                self._communicator_ready = True
            stream.wait_stream(cur)
            with torch.cuda.stream(stream):
                for _ in range(2):
                    local = self._tail_block(self._encode_block(frames))
                    ar, al = self._pools_stand_in(local["ref"], local["logits"], clips)
                    self.temporal_forward(local, ar, al, first_frame=first, clips=clips)
                pools = (ar.clone(), al.clone()) if sharded else None        # static buffers of the gathered pools
            torch.cuda.synchronize(frames.device)
            if lane_carried_collectives:
                _time.sleep(0.5)
            slot = {"in": frames.clone()}
            slot["g1"] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(slot["g1"], stream=stream, capture_error_mode="thread_local"):
                local = self._tail_block(self._encode_block(slot["in"]))
                if not sharded:
                    slot["out"] = self.temporal_forward(local, local["ref"], local["logits"], first_frame=first, clips=clips)
            slot["local"] = local
            if sharded:
                slot["ref"], slot["logits"] = pools
                slot["g2"] = torch.cuda.CUDAGraph()
                with torch.cuda.graph(slot["g2"], stream=stream, pool=slot["g1"].pool(), capture_error_mode="thread_local"):
                    slot["out"] = self.temporal_forward(local, slot["ref"], slot["logits"], first_frame=first, clips=clips)
        except RuntimeError as e:                                          # (a failed capture leaves the eager route intact)
            err = str(e).splitlines()[0][:200]
            torch.cuda.synchronize(frames.device)
        ok = err is None
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            # one decision for all ranks: a rank replaying graphs and a rank launching eagerly would still issue the same
            # collectives, but a rank that failed half-way must not be the only one to know
            flag = torch.tensor([int(ok)], dtype=torch.int32, device=frames.device if dist.get_backend(self.group) == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            ok = bool(int(flag.item()))
        if not ok:
            print(f"ClipRunner: HIP-graph capture failed ({err or 'on another rank'}); this shape runs eagerly", file=sys.stderr, flush=True)
            return False
        return slot


class VideoStream:
    """The reference's VIDEO inference mode as a stream (inference.py:750-765, 879-883): frame t of a video is detected
    with the ``num_ref_frames`` frames ``sample_reference_ids(t, video, R)`` as its reference frames - the window
    [t - R, t + R] without t, first R entries, i.e. the R PREVIOUS frames for every t >= R, and the first R + 1 frames of the
    video without t for the frames before that (repeated when the video is shorter) - where ``ClipRunner.__call__`` serves a
    fixed clip in which every frame sees all others.

    Frames arrive in blocks (``push``); each frame's spatial stage and query/RoI fusion run once, its "ref" query set and
    class logits go into a bank of the last R + 1 frames, and a frame's temporal stage runs as soon as its reference set is
    complete: at once for t >= R, when frame R has arrived (or the video has ended) for the first R frames.  With several
    ranks every ``push`` takes the rank's contiguous block of the world * F new frames (rank-major = frame order), the
    per-frame sets are exchanged with the same single all-gather as a clip's, every rank keeps the whole bank and emits the
    outputs of its own frames."""

    def __init__(self, runner, num_ref_frames=None, filter_key_img=True):
        self.runner = runner
        self.R = int(num_ref_frames if num_ref_frames is not None else runner.model.transformer.num_ref_frames)
        self.filter_key_img = filter_key_img
        self.seen = 0              # frames of the video pushed so far (all ranks)
        self.bank = {}             # frame index -> (ref [Q,C], logits [Q,K])
        self.pending = {}          # frame index -> the frame's own tensors, until its output is emitted
        self.meta = None

    def _reference_ids(self, t, n):
        from .inference_io import sample_reference_ids
        return sample_reference_ids(t, list(range(n)), self.R, self.filter_key_img)

    def _ready(self, t, ended):
        """Reference set of frame t computable from the frames seen so far?  (the window reaches up to frame t + R, but
        only its first R entries are used)"""
        if ended:
            return True
        need = t if t >= self.R else self.R          # highest index among the first R entries of the window
        if not self.filter_key_img:
            need = t if t >= self.R else self.R - 1
        return self.seen > max(need, t)

    @torch.no_grad()
    def push(self, frames, mask=None, last=False):
        """frames [F,C,H,W]: this rank's block of the next world * F frames of the video; ``last``: the video ends with
        this block.  -> [(frame index, {"pred_logits" [Q,K], "pred_boxes" [Q,4]}), ...] for the rank's frames whose
        outputs became computable, in frame order.
        Every rank passes the SAME F on a call (frame ``seen + rank * F + i``; the all-gather carries equal blocks): a video
        whose length is not a multiple of world * F ends with a push of fewer frames per rank - down to F = 1 - and, when
        fewer than ``world`` frames are left, with the last of them repeated on the ranks that have none and ``pad`` = the
        number of repeated frames (they are computed and dropped).  A differing F is caught here, before the collective."""
        return self._push(frames, mask, last, 0)

    @torch.no_grad()
    def push_tail(self, frames, pad, mask=None):
        """The video's last block when fewer than world * F frames are left: every rank still passes F frames, the final
        ``pad`` frames of the world * F (rank-major) being repeats that are computed and dropped.  Ends the video."""
        return self._push(frames, mask, True, int(pad))

    def _push(self, frames, mask, last, pad):
        r = self.runner
        world = dist.get_world_size(r.group) if dist.is_available() and dist.is_initialized() else 1
        rank = dist.get_rank(r.group) if world > 1 else 0
        F_ = frames.shape[0]
        if world > 1:       # unequal blocks would hang or mis-index the gather: one tiny collective buys a clear error
            sizes = torch.tensor([F_, pad], dtype=torch.long, device=frames.device if dist.get_backend(r.group) == "nccl" else "cpu")
            lo, hi = sizes.clone(), sizes.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=r.group)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=r.group)
            if not torch.equal(lo, hi):
                raise ValueError(f"VideoStream.push: every rank must pass the same number of frames and the same pad "
                                 f"(this rank: {F_}, {pad}; over the ranks: {lo.tolist()} .. {hi.tolist()})")
        assert 0 <= pad < world * F_, "pad counts repeated frames at the end of the world * F block"
        local = r.frames_forward(frames, mask)
        all_ref, all_logits = r.exchange(local["ref"], local["logits"])
        base = self.seen
        real = world * F_ - pad
        for j in range(real):
            self.bank[base + j] = (all_ref[j], all_logits[j])
        for i in range(F_):
            if rank * F_ + i < real:
                # (slices of a block are cloned when it holds several frames: a view would pin the whole block's memory
                # tensor [F,S,C] until the first R frames' outputs are emitted)
                self.pending[base + rank * F_ + i] = {k: (local[k][i].clone() if F_ > 1 else local[k][i])
                                                     for k in ("cur", "ref_last", "memory", "valid_ratios")}
        self.meta = (local["spatial_shapes"], local["level_start_index"])
        self.seen = base + real
        return self._emit(last)

    def _emit(self, ended):
        r = self.runner
        m = r.model
        ready = [t for t in sorted(self.pending) if self._ready(t, ended)]
        out = []
        if ready:
            n = self.seen
            ids = sorted(self.bank)
            row = {t: i for i, t in enumerate(ids)}
            refs = [self._reference_ids(t, n if ended else max(n, t + self.R + 1)) for t in ready]
            dev = self.bank[ids[0]][0].device
            others = torch.as_tensor([[row[j] for j in ref] for ref in refs], dtype=torch.long, device=dev)
            pool_ref = torch.stack([self.bank[t][0] for t in ids])
            pool_logits = torch.stack([self.bank[t][1] for t in ids])
            local = {k: torch.stack([self.pending[t][k] for t in ready]) for k in ("cur", "ref_last", "memory", "valid_ratios")}
            local["spatial_shapes"], local["level_start_index"] = self.meta
            res = r.temporal_forward(local, pool_ref, pool_logits, 0, others=others)
            for i, t in enumerate(ready):
                out.append((t, {"pred_logits": res["pred_logits"][i], "pred_boxes": res["pred_boxes"][i]}))
                del self.pending[t]
        # frames no later frame can refer to: everything more than R behind the oldest frame still to be emitted / to come
        # (the first R frames refer to frames 0 .. R: nothing goes while one of them is pending)
        oldest = min(list(self.pending) + [self.seen])
        for t in [t for t in self.bank if t < oldest - self.R]:
            del self.bank[t]
        return out
