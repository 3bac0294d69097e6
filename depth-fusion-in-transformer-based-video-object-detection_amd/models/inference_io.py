"""The steps on either side of the hot path, host-side Python.

Before the path
  ``sample_reference_ids``   which frames of a video serve as reference frames of frame ``img_id``
                             (/root/reference/inference.py:750-765)
  ``assemble_clip``          current + reference frames (RGB or RGB-D) stacked on the channel axis,
                             the [T*C,H,W] tensor the clip collate splits again (inference.py:778-794,
                             util/misc_multi.py:319-340)
  ``load_checkpoint``        the reference's checkpoint wire format {'model': state_dict, ...} incl. the
                             temporal / spatial weight merging of main_multi.py:333-394 and
                             inference.py:807-823
The caller itself
  ``FrameInference``         the body of the reference's ``infer()`` loop (inference.py:879-956) as one chain on the GPU:
                             preprocessing kernel -> model (single frame) or ClipRunner (TransVOD++, every frame of the
                             clip as current frame) -> post-filter -> box rescale -> label lines
After the path
  ``filter_detections``      softmax over classes, keep queries whose class-1 ("hand") probability exceeds
                             ``keep_prob`` (inference.py:918-930)
  ``rescale_bboxes``         (cx,cy,w,h) in [0,1] -> pixel (x1,y1,x2,y2) of the original image (:475-490)
  ``yolo_lines``             the label-file lines ``Hand cx cy w h p`` (:951-956)
"""
import torch

from util.box_ops import box_cxcywh_to_xyxy

TEMPORAL_KEYS = {
    "vid_multi": ["temporal_query", "temporal_decoder", "temp_bbox_embed"],
    "vid_multi_plusplus": ["temporal_query", "dynamic_layer", "temporal_decoder", "temp_bbox_embed"],
}


def sample_reference_ids(img_id, video_img_ids, num_ref_frames, filter_key_img=True):
    """IDs of the reference frames of ``img_id``: the window [img_id - R, img_id + R] clipped to the
    video, optionally without the frame itself, repeated until it holds R entries, first R taken."""
    left = max(video_img_ids[0], img_id - num_ref_frames)
    right = min(video_img_ids[-1], img_id + num_ref_frames)
    window = list(range(left, right + 1))
    if filter_key_img and img_id in window:
        window.remove(img_id)
    if not window:
        raise ValueError("a one-frame video has no reference frames")
    while len(window) < num_ref_frames:
        window.extend(window)
    return window[:num_ref_frames]


def assemble_clip(rgb_frames, depth_frames=None):
    """[3,H,W] RGB (and [1,H,W] depth) tensors of the current frame followed by its reference frames
    -> one [T*C,H,W] tensor, C = 4 with depth."""
    parts = []
    for i, rgb in enumerate(rgb_frames):
        assert rgb.shape[0] == 3, "Image should have 3 RGB channels."
        if depth_frames is None:
            parts.append(rgb)
        else:
            depth = depth_frames[i]
            assert depth.shape[0] == 1, "Depth should have 1 channel."
            parts.append(torch.cat([rgb, depth], dim=0))
    return torch.cat(parts, dim=0)


def merge_checkpoints(checkpoint, temporal_checkpoint=None, spatial_checkpoint=None, dataset_file="vid_multi_plusplus"):
    """state_dict of ``checkpoint`` with the temporal modules taken from a TransVOD(++) checkpoint and
    every tensor of a separately fine-tuned spatial checkpoint laid over it (later wins)."""
    state = dict(checkpoint["model"])
    if temporal_checkpoint is not None and dataset_file in TEMPORAL_KEYS:
        wanted = TEMPORAL_KEYS[dataset_file]
        state.update({k: v for k, v in temporal_checkpoint["model"].items() if any(w in k for w in wanted)})
    if spatial_checkpoint is not None:
        state.update(spatial_checkpoint["model"])
    return state


def load_checkpoint(model, resume, spatial_weights=None, transvod_temporal_weights=None,
                    dataset_file="vid_multi_plusplus", map_location="cpu"):
    """Load a reference checkpoint file (and optional temporal / spatial companions) into ``model`` with
    ``strict=False``; returns (missing_keys, unexpected_keys) with the thop counters filtered out."""
    load = lambda p: torch.load(p, map_location=map_location, weights_only=False)  # noqa: E731
    state = merge_checkpoints(load(resume),
                              load(transvod_temporal_weights) if transvod_temporal_weights else None,
                              load(spatial_weights) if spatial_weights else None, dataset_file)
    missing, unexpected = model.load_state_dict(state, strict=False)
    unexpected = [k for k in unexpected if not (k.endswith("total_params") or k.endswith("total_ops"))]
    return list(missing), unexpected


@torch.no_grad()
def filter_detections(outputs, keep_prob=0.5, batch_index=0):
    """-> (probs [K], boxes [K,4] normalised cxcywh, query indices [K]) of the kept queries."""
    probas = outputs["pred_logits"].softmax(-1)[batch_index]
    keep = probas[:, 1] > keep_prob
    idx = keep.nonzero(as_tuple=False).flatten()
    return probas[idx, 1], outputs["pred_boxes"][batch_index][idx], idx


def rescale_bboxes(out_bbox, size):
    """size = (width, height) of the original image."""
    img_w, img_h = size
    scale = torch.tensor([img_w, img_h, img_w, img_h], dtype=torch.float32, device=out_bbox.device)
    return box_cxcywh_to_xyxy(out_bbox) * scale


def yolo_lines(normalized_boxes, probs, label="Hand"):
    return [f"{label} {cx:.8f} {cy:.8f} {w:.8f} {h:.8f} {p:.8f}"
            for (cx, cy, w, h), p in zip(normalized_boxes.tolist(), probs.tolist())]


class FrameInference:
    """The per-image body of the reference's inference loop (inference.py:879-956) chained on the GPU:

        uint8 frames -> fused resize / normalise / pad kernel (models/preprocess.py, row f4)
                     -> ``model(NestedTensor)`` for the single-frame detectors, or ``ClipRunner`` for TransVOD(++)
                        clips (every frame of the clip is a current frame, models/clip_inference.py)
                     -> ``softmax(-1)[:, 1] > keep_prob`` -> kept boxes, pixel boxes of the ORIGINAL image,
                        label lines ``Hand cx cy w h p`` (row f1)

    One result dict per frame: probs [K], boxes [K,4] (normalised cxcywh), queries [K] (indices of the kept queries),
    boxes_px [K,4] (x1,y1,x2,y2 in the original image), lines (list of str; empty = the reference writes no file)."""

    def __init__(self, model, keep_prob=0.5, size=600, max_size=1333, micro_batch=32):
        from .clip_inference import ClipRunner
        from .fused import enable_fused_inference
        from .preprocess import ClipPreprocessor
        self.model = model.eval()
        self.keep_prob = keep_prob
        dev = next(model.parameters()).device
        self.pre = ClipPreprocessor(size, max_size, device=dev)
        self.is_clip_model = hasattr(model.transformer, "temporal_stage")
        if self.is_clip_model:
            self.runner = ClipRunner(model, micro_batch=micro_batch)
        else:
            enable_fused_inference(model, dev.type == "cuda")

    def _finish(self, outputs, b, size_wh):
        probs, boxes, idx = filter_detections(outputs, self.keep_prob, b)
        return {"probs": probs, "boxes": boxes, "queries": idx, "boxes_px": rescale_bboxes(boxes, size_wh),
                "lines": yolo_lines(boxes, probs)}

    @torch.no_grad()
    def image(self, rgb_u8, depth_u8=None):
        """One image through a single-frame detector (``dataset_file == 'vid_single'``, inference.py:884-889)."""
        nt = self.pre([rgb_u8], None if depth_u8 is None else [depth_u8])
        return self._finish(self.model(nt), 0, (rgb_u8.shape[1], rgb_u8.shape[0]))

    @torch.no_grad()
    def clip(self, rgb_frames, depth_frames=None):
        """A clip of T frames through TransVOD(++): frame t's result is the reference's for the clip
        [t, the other frames in clip order] (inference.py:879-883 with that clip)."""
        nt = self.pre(rgb_frames, depth_frames)
        padded = bool(nt.mask.any())
        out = self.runner(nt.tensors, nt.mask if padded else None)
        return [self._finish(out, t, (f.shape[1], f.shape[0])) for t, f in enumerate(rgb_frames)]
