"""Shared recipe for the checkpoint-merge fixture: synthetic ``{'model': state_dict}`` files whose tensor
VALUES name their origin (1 = the resumed checkpoint, 2 = the TransVOD temporal checkpoint, 3 = the spatial fine-tune),
a zero-initialised target model whose key names hit every rule of the merge, and the cases to run.

tools/gen_golden_inference.py runs the REFERENCE's own resume block (main_multi.py, ``if args.resume:`` ... the
``total_ops`` filter, eval branch) on them; tests/test_inference_io.py runs models.inference_io.load_checkpoint."""
import os

import torch
from torch import nn

# (dataset_file, temporal checkpoint given, spatial checkpoint given)
CASES = [("vid_multi_plusplus", True, True), ("vid_multi_plusplus", True, False), ("vid_multi", True, False),
         ("vid_multi", True, True), ("vid_multi_plusplus", False, True), ("vid_multi_plusplus", False, False),
         ("vid_single", True, False)]

NAMES = ["backbone", "input_proj", "class_embed", "bbox_embed", "transformer_encoder", "temporal_query_layer1",
         "temporal_decoder1", "temp_class_embed", "temp_bbox_embed", "dynamic_layer_for_current_query1", "only_in_model"]


class Target(nn.Module):
    def __init__(self):
        super().__init__()
        for n in NAMES:
            setattr(self, n, nn.Linear(2, 2))
        for p in self.parameters():
            nn.init.zeros_(p)


def write_checkpoints(folder):
    keys = [k for k in Target().state_dict() if not k.startswith("only_in_model")]
    full = lambda v: {k: torch.full((2, 2) if k.endswith("weight") else (2,), float(v)) for k in keys}  # noqa: E731
    base = full(1)
    base["stray_module.weight"] = torch.ones(1)                # unexpected for the model
    base["backbone.total_ops"] = torch.zeros(1)                # thop counters: filtered from the report
    temporal = full(2)
    temporal["temporal_decoder1.total_params"] = torch.zeros(1)
    temporal["dynamic_layer_for_current_query9.weight"] = torch.ones(1)
    spatial = {k: v for k, v in full(3).items() if k.startswith(("backbone", "input_proj", "temp_bbox_embed"))}
    paths = {}
    for name, sd in (("base", base), ("temporal", temporal), ("spatial", spatial)):
        paths[name] = os.path.join(folder, f"{name}.pth")
        torch.save({"model": sd, "epoch": 7}, paths[name])
    return paths


def describe(model, missing, unexpected):
    """-> JSON-able summary: origin of every tensor the model ended up with, and the reported key lists."""
    return {"origin": {k: float(v.flatten()[0]) for k, v in model.state_dict().items()},
            "missing": sorted(missing), "unexpected": sorted(unexpected)}


# ---- DFormer partial load (dformer_backbone.py:161-198) ---------------------------------------------------------------
def write_dformer_checkpoint(path, depth_backbone):
    """A synthetic DFormer checkpoint ``{'state_dict': ...}``: for every conv / BatchNorm of the depth stem the tensors of
    the depth branch (``backbone.downsample_layers_e.<i>.<j>.*``, values i*100 + j*10 + {1: weight, 2: bias, 3: running_mean,
    4: running_var}), the same names of the RGB branch (``downsample_layers.``, must be ignored), one tensor of the wrong
    shape and one unrelated key."""
    sd = {}
    for name, mod in depth_backbone.named_modules():
        if not isinstance(mod, (nn.Conv2d, nn.BatchNorm2d)):
            continue
        _, i, j = name.split(".")
        base = int(i) * 100 + int(j) * 10
        for branch, offset in (("downsample_layers_e", 0.0), ("downsample_layers", 0.5)):
            key = f"backbone.{branch}.{i}.{j}"
            sd[key + ".weight"] = torch.full_like(mod.weight, base + 1 + offset)
            sd[key + ".bias"] = torch.full_like(mod.bias, base + 2 + offset)
            if isinstance(mod, nn.BatchNorm2d):
                sd[key + ".running_mean"] = torch.full_like(mod.running_mean, base + 3 + offset)
                sd[key + ".running_var"] = torch.full_like(mod.running_var, base + 4 + offset)
                sd[key + ".num_batches_tracked"] = torch.tensor(7)
    sd["backbone.downsample_layers_e.0.0.weight"] = torch.full((16, 1, 5, 5), 999.0)      # wrong shape: weight stays
    sd["decode_head.conv_seg.weight"] = torch.ones(3)
    torch.save({"state_dict": sd}, path)


def describe_dformer(depth_backbone):
    return {k: float(v.flatten()[0]) for k, v in depth_backbone.state_dict().items()}


# ---- collate inputs (util/misc.py:304-356, util/misc_multi.py:304-345) -------------------------------------------------
def collate_inputs():
    """Ragged seeded images [3,H,W] and clips [(1+R)*C,H,W] (RGB-D: C = 4; RGB: C = 3; 'nosplit': plain 3-d tensors)."""
    g = torch.Generator().manual_seed(77)
    imgs = [torch.randn(3, h, w, generator=g) for h, w in ((11, 17), (13, 9), (8, 20))]
    clips = {"rgbd": [torch.randn(3 * 4, h, w, generator=g) for h, w in ((10, 14), (12, 9))],
             "rgb": [torch.randn(2 * 3, h, w, generator=g) for h, w in ((7, 7), (5, 12), (9, 3))],
             "nosplit": [torch.randn(3, h, w, generator=g) for h, w in ((6, 8), (4, 10))]}
    return imgs, clips
