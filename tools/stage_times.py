"""Per-stage GPU time of one micro-batch of the TransVOD++ Late-Fusion path (HIP events)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "depth-fusion-in-transformer-based-video-object-detection_amd"))
sys.path.insert(0, ROOT)
from bench import build  # noqa: E402
from models.clip_inference import ClipRunner  # noqa: E402
from util.misc import NestedTensor  # noqa: E402

torch.backends.cuda.matmul.allow_tf32 = False
torch.backends.cudnn.allow_tf32 = False
F_ = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda")
model = build(dev, 31)
runner = ClipRunner(model, micro_batch=F_)
x = torch.randn(F_, 4, 800, 1333, device=dev)
mask = torch.zeros(F_, 800, 1333, dtype=torch.bool, device=dev)
marks = []


def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append((name, e))


@torch.no_grad()
def once():
    marks.clear()
    m, tr = model, model.transformer
    mark("start")
    rgb = NestedTensor(x[:, :3], mask)
    body = m.backbone[0].body
    h = body.stem(rgb.tensors, True); mark("resnet stem (conv7x7+bias+relu+maxpool)")
    h = body.run_stage(body.layer1, h, True); mark("resnet layer1")
    h = body.run_stage(body.layer2, h, True); mark("resnet layer2")
    h = body.run_stage(body.layer3, h, True); mark("resnet layer3")
    h = body.run_stage(body.layer4, h, True); mark("resnet layer4 (DC5)")
    d = m.depth_backbone(NestedTensor(x[:, 3:4], mask)); mark("dformer stem + pos")
    srcs, masks, pos, d_srcs, d_masks, d_pos, rgbd = m._encode_inputs(NestedTensor(x, mask)); mark("(_encode_inputs again: backbones + input_proj)")
    st = tr._spatial_stage(srcs, masks, pos, d_srcs, d_masks, d_pos, m.query_embed.weight, rgbd); mark("transformer spatial stage (LF + 6 enc + 6 dec)")
    whwh = torch.as_tensor((1333, 800, 1333, 800), dtype=torch.long, device=dev).repeat(1, m.num_queries, 1)
    fs = tr.frame_stage(st["hs"][-1], st["inter_references"][-1], st["memory"], st["lvl_pos_embed_flatten"],
                        st["last_hw"], whwh, m.class_embed[-1], m.bbox_embed[-1]); mark("frame stage (2x RoIAlign + RCNNHead)")
    T = 32
    all_ref = fs["ref"].repeat(T // F_, 1, 1); all_lg = fs["logits"].repeat(T // F_, 1, 1)
    local = dict(cur=fs["cur"], ref_last=st["inter_references"][-1], memory=st["memory"], spatial_shapes=st["spatial_shapes"],
                 level_start_index=st["level_start_index"], valid_ratios=st["valid_ratios"])
    runner.temporal_forward(local, all_ref, all_lg, 0); mark("temporal stage (3x topk+TQE+TDTD), R=31")


for _ in range(3):
    once()
torch.cuda.synchronize()
tot = 0.0
for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
    ms = e0.elapsed_time(e1)
    print(f"{ms:8.2f} ms  {ms / F_:6.3f} ms/frame  {n1}")
print("note: '_encode_inputs again' repeats the backbones; subtract the first six lines from it to get input_proj + pos")

# finer: spatial stage pieces
import time
from models.transformer_layers import get_reference_points  # noqa: E402
