"""Algorithmic bytes and flops of the path, per frame and per op family, for BASELINE.json's configurations
("the builder must commit the generator script for these numbers"):

    python tools/algorithmic_work.py            # table for configs A-E, writes tools/algorithmic_work.json
    python tools/algorithmic_work.py --check    # also compares with BASELINE.md section 2 / SURVEY.md 8d

How it counts.  The detector is BUILT through ``build_model`` and RUN once on CPU tensors at 800 x 1333 (config A:
794 x 600) with the oracle as the two custom operators, under a ``TorchDispatchMode`` that sees every ATen call with
its operand shapes; module hooks tell which part of the model an op belongs to.  Accounting rule of BASELINE.md section 2:

  * a contraction (convolution, Linear / matmul / bmm), a normalisation (LayerNorm, GroupNorm), a sampling operator
    (MSDA core, RoIAlign) reads its activation inputs once and writes its output once: 4 bytes per element;
    parameters are counted apart ("weights"), once per frame
  * batch-norm in eval mode / FrozenBatchNorm, bias, ReLU / GELU / sigmoid, dropout, the stem's max-pool and every other
    elementwise op are folded into their producer: no traffic
  * a residual or positional add of two activation tensors of the same shape costs one extra read of one operand
  * attention is counted flash-style: the score matrix of softmax(QK^T)V never travels (the product before a softmax
    does not write its output, the product after it does not read it)
  * flops: 2 per multiply-accumulate of a contraction, 10 per (query, head, level, point, channel) of the MSDA core
    (4 bilinear corners x 2 + weighting); normalisations, elementwise ops and RoIAlign's averages are not counted

TransVOD++ (configs D, E) is counted per OUTPUT frame: one frame's spatial stage + its query/RoI fusion + one temporal
stage against R reference frames' query sets.  The reference forward pays one RoIAlign + RCNNHead pass per frame
("literal" figure); in all-current mode (every frame of the clip is also a current frame, models/clip_inference.py) each
frame needs the pass twice, once on plain memory and once on memory + positions: the "all-current" figure, which is what
bench.py's `e2e` fractions use.

The same walk yields, per kernel family of csrc/ (fp32 MFMA GEMM = 1x1 convolutions + Linears, Winograd = 3x3 stride-1
convolutions, implicit GEMM = the other convolutions, MSDA core), the one-pass bytes bench.py prints next to the measured
traffic (`algorithmic_bytes`).
"""
import argparse
import collections
import json
import os
import sys

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "depth-fusion-in-transformer-based-video-object-detection_amd")
for _p in (PKG, ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)

OUT_JSON = os.path.join(ROOT, "tools", "algorithmic_work.json")

# BASELINE.md section 2 / SURVEY.md 8d (activation MB, weight MB, GFLOP per frame)
PUBLISHED = {"A": None, "baseline": (3413, 138, 312.2), "B": (3613, 140, 317.5), "C": (3946, 144, 326.6), "D": (3688, 213, 327.2),
             "E": (3909, 215, 334.6)}
PUBLISHED_ALL_CURRENT_E = (4.333e9, 343.3e9)

PARTS = (("backbone.", "resnet50"), ("depth_backbone.", "dformer"), ("input_proj", "input_proj"),
         ("transformer.depth_encoder_layer", "late_fusion"), ("transformer.encoder.fusion_layers", "encoder_cross_fusion"),
         ("transformer.encoder", "encoder"), ("transformer.decoder", "decoder"),
         ("transformer.dynamic_layer_for_current_query", "query_roi_fusion"), ("transformer.temporal_roi", "query_roi_fusion"),
         ("transformer.temporal_", "temporal"), ("temp_", "temporal"), ("class_embed", "heads"), ("bbox_embed", "heads"),
         ("transformer.reference_points", "decoder"), ("transformer", "transformer_glue"))


def part_of(path):
    for prefix, name in PARTS:
        if path.startswith(prefix):
            return name
    return "other"


class Walk(TorchDispatchMode):
    """Records (part, family, activation bytes, weight bytes, flops) for every counted op of one forward."""

    def __init__(self, model):
        super().__init__()
        self.params = set()
        for t in list(model.parameters()) + list(model.buffers()):
            self.params.add(t.untyped_storage().data_ptr())
        self.stack = ["(top)"]
        self.rows = []            # dicts
        self.section = "spatial"
        self._pending_softmax_producer = None
        self._softmax_out = None
        self.handles = []
        for name, mod in model.named_modules():
            if name:
                self.handles.append(mod.register_forward_pre_hook(lambda m, a, n=name: self.stack.append(n)))
                self.handles.append(mod.register_forward_hook(lambda m, a, o: (self.stack.pop(), None)[1]))

    def close(self):
        for h in self.handles:
            h.remove()

    # ---- helpers ----
    def is_param(self, t):
        return isinstance(t, torch.Tensor) and t.untyped_storage().data_ptr() in self.params

    def add(self, family, act_in, act_out, weights, flops, note=""):
        self.rows.append(dict(section=self.section, part=part_of(self.stack[-1]), module=self.stack[-1], family=family,
                              act_bytes=4 * (act_in + act_out), in_bytes=4 * act_in, out_bytes=4 * act_out,
                              weight_bytes=4 * weights, flops=flops, note=note))
        return self.rows[-1]

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func._schema.name
        try:
            self.count(name, args, out)
        except Exception as e:       # a tool, not the product: say what was not understood and go on
            print(f"  [not counted] {name}: {e}", file=sys.stderr)
        return out

    def count(self, name, args, out):
        T = torch.Tensor
        if name in ("aten::convolution", "aten::_convolution", "aten::mkldnn_convolution", "aten::convolution_overrideable",
                    "aten::_slow_conv2d_forward", "aten::slow_conv_dilated2d", "aten::thnn_conv2d", "aten::_conv_depthwise2d"):
            x, w = args[0], args[1]
            stride = args[3] if name in ("aten::convolution", "aten::_convolution", "aten::convolution_overrideable") else None
            if name in ("aten::_slow_conv2d_forward",):
                stride = args[4]
            if name == "aten::slow_conv_dilated2d":
                stride = args[4]
            if name == "aten::mkldnn_convolution":
                stride = args[4]
            b = args[2] if len(args) > 2 and isinstance(args[2], T) else None
            Co, Ci, kh, kw = w.shape
            s = int(stride[0]) if stride else 1
            fam = "gemm_kn" if (kh == 1 and kw == 1 and s == 1) else ("wino" if (kh == 3 and kw == 3 and s == 1) else "igemm")
            self.add(fam, x.numel(), out.numel(), w.numel() + (b.numel() if b is not None else 0),
                     2.0 * out.numel() * Ci * kh * kw, f"conv {Ci}->{Co} {kh}x{kw}/{s} {tuple(out.shape[-2:])}")
        elif name in ("aten::addmm", "aten::mm", "aten::linear"):
            if name == "aten::addmm":
                _, a, b = args[:3]
            elif name == "aten::linear":
                a, b = args[0], args[1].t()
            else:
                a, b = args[:2]
            K = a.shape[-1]
            wa, wb = self.is_param(a), self.is_param(b)
            act_in = (0 if wa else a.numel()) + (0 if wb else b.numel())
            wts = (a.numel() if wa else 0) + (b.numel() if wb else 0)
            if name == "aten::addmm" and self.is_param(args[0]):
                wts += args[0].numel()
            self.add("gemm_nk", act_in, out.numel(), wts, 2.0 * out.numel() * K, f"linear {tuple(a.shape)} x {tuple(b.shape)}")
        elif name in ("aten::bmm", "aten::baddbmm"):
            a, b = (args[1], args[2]) if name == "aten::baddbmm" else (args[0], args[1])
            row = self.add("bmm", a.numel() + b.numel(), out.numel(), 0, 2.0 * out.numel() * a.shape[-1],
                           f"bmm {tuple(a.shape)} x {tuple(b.shape)}")
            row["_out_ptr"] = out.untyped_storage().data_ptr()
            row["_in_ptrs"] = (a.untyped_storage().data_ptr(), b.untyped_storage().data_ptr())
            row["_a_numel"] = a.numel()
            if self._softmax_out is not None and self._softmax_out in row["_in_ptrs"]:
                # product after a softmax: the probabilities are not read from memory (flash-style)
                row["act_bytes"] -= 4 * a.numel()
                row["in_bytes"] -= 4 * a.numel()
                self._softmax_out = None
        elif name in ("aten::_softmax", "aten::softmax"):
            x = args[0]
            ptr = x.untyped_storage().data_ptr()
            for row in reversed(self.rows[-6:]):
                if row.get("_out_ptr") == ptr and row["family"] == "bmm":
                    row["act_bytes"] -= row["out_bytes"]         # product before a softmax: scores are not written
                    row["out_bytes"] = 0
                    self._softmax_out = out.untyped_storage().data_ptr()
                    break
        elif name in ("aten::native_layer_norm", "aten::native_group_norm"):
            x = args[0]
            y = out[0]
            self.add("norm", x.numel(), y.numel(), sum(a.numel() for a in args[1:] if isinstance(a, T) and self.is_param(a)), 0.0,
                     name.split("::")[1])
        elif name in ("aten::max_pool2d_with_indices", "aten::max_pool2d"):
            # folded into its producer like every elementwise op: the convolution in front writes the pooled map
            y = out[0] if isinstance(out, (tuple, list)) else out
            for row in reversed(self.rows[-3:]):
                if row["family"] in ("igemm", "wino", "gemm_kn") and row["out_bytes"] == 4 * args[0].numel():
                    row["act_bytes"] -= row["out_bytes"] - 4 * y.numel()
                    row["out_bytes"] = 4 * y.numel()
                    row["note"] += " (+ max-pool folded)"
                    break
        elif name in ("aten::add", "aten::add_"):
            a, b = args[0], args[1]
            if (isinstance(a, T) and isinstance(b, T) and a.shape == b.shape and a.numel() >= 65536 and a.is_floating_point()
                    and not self.is_param(a) and not self.is_param(b)):
                self.add("residual", a.numel(), 0, 0, 0.0, f"add {tuple(a.shape)}")

    # custom operators (not ATen): called by the patched entry points below
    def msda(self, value, shapes, loc, aw, out):
        N, S, M, D = value.shape
        Lq = out.shape[1]
        L, P = shapes.shape[0], loc.numel() // (N * Lq * M * shapes.shape[0] * 2)
        self.add("msda", N * S * M * D + 3 * N * Lq * M * L * P, N * Lq * M * D, 0, 10.0 * N * Lq * M * L * P * D,
                 f"msda N={N} Lq={Lq} S={S} L={L} P={P}")

    def roi(self, inp, out):
        self.add("roi_align", inp.numel(), out.numel(), 0, 0.0, f"roi_align {tuple(inp.shape)} -> {tuple(out.shape)}")


def install_custom_ops(walk_ref):
    """CPU operators = the oracle (test infrastructure; this tool never runs in the product path), wrapped to be counted."""
    import models.ops.functions.ms_deform_attn_func as f
    from dfx import ops
    from oracle import msda_oracle

    class Counted:
        @staticmethod
        def apply(value, shapes, lsi, loc, aw, step):
            out = msda_oracle.OracleMSDAFunction.apply(value, shapes, lsi, loc, aw, step)
            if walk_ref[0] is not None:
                walk_ref[0].msda(value, shapes, loc, aw, out)
            return out

    def roi_align(inp, rois, output_size, spatial_scale, sampling_ratio, aligned=True, channels_last=False):
        size = output_size if isinstance(output_size, int) else output_size[0]
        x = inp.permute(0, 3, 1, 2).contiguous() if channels_last else inp
        w, walk_ref[0] = walk_ref[0], None           # the layout copy above is not part of the path
        out = msda_oracle.roi_align(x, rois, size, spatial_scale, sampling_ratio, aligned)
        walk_ref[0] = w
        if w is not None:
            w.roi(inp, out)
        return out.flatten(2).transpose(1, 2).contiguous() if channels_last else out

    f.MSDeformAttnFunction, ops.roi_align = Counted, roi_align


def kernel_families(rows, passes_of_qrf=1):
    """One-pass bytes / flops per frame of the kernel families of csrc/ that bench.py reports (`algorithmic_bytes`):
    gemm = every 1x1 stride-1 convolution and every Linear incl. the residual reads of the ResNet bottlenecks (they ride in
    the GEMM's epilogue); wino = 3x3 stride-1 convolutions; igemm = the other convolutions."""
    out = {k: dict(act=0.0, weights=0.0, flops=0.0, ops=0) for k in ("gemm", "wino", "igemm", "msda")}
    for r in rows:
        mult = passes_of_qrf if r["section"] == "qrf" else 1
        fam = {"gemm_kn": "gemm", "gemm_nk": "gemm", "wino": "wino", "igemm": "igemm", "msda": "msda"}.get(r["family"])
        if r["family"] == "residual" and r["part"] == "resnet50":
            fam = "gemm"
        if fam is None:
            continue
        out[fam]["act"] += r["act_bytes"] * mult
        out[fam]["weights"] += r["weight_bytes"]
        out[fam]["flops"] += r["flops"] * mult
        out[fam]["ops"] += mult if r["family"] != "residual" else 0
    return out


def summarise(rows, passes_of_qrf=1):
    """-> totals and per-part / per-family tables for one frame."""
    tot = collections.OrderedDict(act=0.0, weights=0.0, flops=0.0)
    parts, fams = collections.OrderedDict(), collections.OrderedDict()
    for r in rows:
        mult = passes_of_qrf if r["section"] == "qrf" else 1
        act = r["act_bytes"] * mult
        fl = r["flops"] * mult
        tot["act"] += act
        tot["weights"] += r["weight_bytes"]
        tot["flops"] += fl
        p = parts.setdefault(r["part"], dict(act=0.0, weights=0.0, flops=0.0))
        p["act"] += act
        p["weights"] += r["weight_bytes"]
        p["flops"] += fl
        f_ = fams.setdefault(r["family"], dict(act=0.0, weights=0.0, flops=0.0, ops=0))
        f_["act"] += act
        f_["weights"] += r["weight_bytes"]
        f_["flops"] += fl
        f_["ops"] += mult
    return tot, parts, fams


@torch.no_grad()
def walk_config(cfg, verbose=False):
    from models import build_model
    from models.config import single_args, transvodpp_args
    from util.misc import NestedTensor
    torch.manual_seed(42)
    H, W = (794, 600) if cfg == "A" else (800, 1333)
    if cfg in ("A", "B", "C", "baseline"):
        fusion = {"A": "Baseline", "baseline": "Baseline", "B": "LateFusion", "C": "Encoder_CrossFusion"}[cfg]
        model, _, _ = build_model(single_args(fusion, device="cpu"))
        R = 0
    else:
        R = 7 if cfg == "D" else 31
        model, _, _ = build_model(transvodpp_args("Baseline" if cfg == "D" else "LateFusion", num_ref_frames=R, device="cpu"))
    model.eval()
    C = 3 if cfg in ("A", "D", "baseline") else 4
    x = torch.randn(1, C, H, W)
    mask = torch.zeros(1, H, W, dtype=torch.bool)
    ref = [None]
    install_custom_ops(ref)
    walk = Walk(model)
    ref[0] = walk
    try:
        with walk:
            if R == 0:
                model(NestedTensor(x, mask))
            else:
                from models.clip_inference import ClipRunner
                runner = ClipRunner(model, micro_batch=1, fused=False)
                m, tr = model, model.transformer
                enc = m._encode_inputs(NestedTensor(x, mask))
                srcs, masks, pos, d_srcs, d_masks, d_pos, rgbd = enc
                st = tr._spatial_stage(srcs, masks, pos, d_srcs, d_masks, d_pos, m.query_embed.weight, rgbd)
                whwh = torch.as_tensor((W, H, W, H), dtype=torch.long).repeat(1, m.num_queries, 1)
                walk.section = "qrf"
                fs = tr.frame_stage(st["hs"][-1], st["inter_references"][-1], st["memory"], st["lvl_pos_embed_flatten"],
                                    st["last_hw"], whwh, m.class_embed[-1], m.bbox_embed[-1], roles=("cur",))
                walk.section = "temporal"
                local = dict(cur=fs["cur"], ref_last=st["inter_references"][-1], memory=st["memory"],
                             spatial_shapes=st["spatial_shapes"], level_start_index=st["level_start_index"],
                             valid_ratios=st["valid_ratios"])
                pool = torch.cat([fs["cur"]] * (R + 1), 0)            # shapes only: R reference frames' query sets
                logits = torch.cat([fs["logits"]] * (R + 1), 0)
                runner.temporal_forward(local, pool, logits, 0)
    finally:
        ref[0] = None
        walk.close()
    return walk.rows


def config_entry(cfg, verbose=False):
    """The JSON entry of one configuration (what tools/algorithmic_work.json holds under its key; tests/test_algorithmic_work.py
    re-derives config E's and compares) -> (entry, rows, totals, parts, families)."""
    rows = walk_config(cfg, verbose)
    tot, parts, fams = summarise(rows, 1)
    entry = {"per_frame": {"activation_bytes": tot["act"], "weight_bytes": tot["weights"], "bytes": tot["act"] + tot["weights"],
                           "flops": tot["flops"]},
             "parts": {k: v for k, v in parts.items()}, "families": {k: v for k, v in fams.items()},
             "kernel_families": kernel_families(rows, 1)}
    if cfg in ("D", "E"):
        tot2, parts2, fams2 = summarise(rows, 2)
        entry["per_frame_all_current"] = {"activation_bytes": tot2["act"], "weight_bytes": tot2["weights"],
                                          "bytes": tot2["act"] + tot2["weights"], "flops": tot2["flops"]}
        entry["families_all_current"] = {k: v for k, v in fams2.items()}
        entry["kernel_families_all_current"] = kernel_families(rows, 2)
    if PUBLISHED.get(cfg):
        pa, pw, pf = PUBLISHED[cfg]
        entry["baseline_md"] = {"activation_MB": pa, "weight_MB": pw, "GFLOP": pf}
    return entry, rows, tot, parts, fams


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="A,baseline,B,C,D,E",
                    help="A = the sample image's size 794x600; baseline = the same detector at 800x1333 (the row BASELINE.md quotes)")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()
    doc = {}
    wanted = a.configs.split(",")
    for cfg in wanted:
        entry, rows, tot, parts, fams = config_entry(cfg, a.verbose)
        doc[cfg] = entry
        line = f"config {cfg}: activations {tot['act'] / 1e6:8.1f} MB  weights {tot['weights'] / 1e6:6.1f} MB  {tot['flops'] / 1e9:7.1f} GFLOP per frame"
        if PUBLISHED.get(cfg):
            pa, pw, pf = PUBLISHED[cfg]
            line += (f"   (BASELINE.md: {pa} MB + {pw} MB, {pf} GFLOP: {tot['act'] / 1e6 / pa - 1:+.1%} / "
                     f"{tot['weights'] / 1e6 / pw - 1:+.1%} / {tot['flops'] / 1e9 / pf - 1:+.1%})")
        print(line)
        for k, v in parts.items():
            print(f"    {k:22s} {v['act'] / 1e6:8.1f} MB act {v['weights'] / 1e6:6.1f} MB w {v['flops'] / 1e9:7.2f} GFLOP")
        print("    -- by kernel family")
        for k, v in fams.items():
            print(f"    {k:22s} {v['act'] / 1e6:8.1f} MB act {v['weights'] / 1e6:6.1f} MB w {v['flops'] / 1e9:7.2f} GFLOP  {v['ops']} ops")
        if cfg in ("D", "E"):
            ac = entry["per_frame_all_current"]
            print(f"    all-current mode (2 query/RoI fusion passes per frame): {ac['bytes'] / 1e9:.3f} GB, {ac['flops'] / 1e9:.1f} GFLOP per frame")
        if a.verbose:
            for r in rows:
                print(f"      {r['section']:8s} {r['part']:18s} {r['family']:9s} {r['act_bytes'] / 1e6:9.2f} MB {r['flops'] / 1e9:8.3f} GF  {r['note']}  [{r['module']}]")
    if set(wanted) >= {"A", "baseline", "B", "C", "D", "E"}:
        with open(OUT_JSON, "w") as fh:
            json.dump(doc, fh, indent=1, sort_keys=True)
        print("wrote", OUT_JSON)
    if a.check and "E" in doc:
        ac = doc["E"]["per_frame_all_current"]
        print(f"config E all-current: {ac['bytes'] / 1e9:.3f} GB / {ac['flops'] / 1e9:.1f} GFLOP  vs BASELINE.md {PUBLISHED_ALL_CURRENT_E[0] / 1e9:.3f} GB / "
              f"{PUBLISHED_ALL_CURRENT_E[1] / 1e9:.1f} GFLOP")


if __name__ == "__main__":
    main()
