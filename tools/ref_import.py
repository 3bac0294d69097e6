"""Import the reference's Python path in THIS container (build container only).

/root/reference does not exist on the GPU box, so nothing under tests/,
bench.py or __graft_entry__ imports this file; only the fixture generators in
tools/ do.  The reference's compiled extension, torchvision and mmcv are absent
here, so the import uses the stub recipe recorded in SURVEY.md section 8c:

  1. an empty ``MultiScaleDeformableAttention`` module (the compiled op is CUDA-only);
  2. a ``torchvision`` stub (version string + ``ops.boxes.box_area``);
  3. ``models`` as a namespace package, so models/__init__.py (which pulls
     torchvision.models and mmcv) is not executed;
  4. ``MSDeformAttnFunction`` replaced by a shim that feeds the reference's own
     ``ms_deform_attn_core_pytorch`` after the CUDA launcher's flat re-indexing
     (ms_deform_attn_cuda.cu:40-48), so the three temporal-decoder calls with a
     [1,300,8,R,4,2] location tensor behave as they do on the CUDA path;
  5. an ``mmcv.ops.RoIAlign`` stub whose forward is handed in by the caller
     (RoIAlign is third-party and unpinned, see oracle/msda_oracle.c).
"""
import sys
import types

import torch

REF = "/root/reference"


def install(roi_align_fn=None):
    if "models.ops.functions.ms_deform_attn_func" in sys.modules:
        return sys.modules["models.ops.functions.ms_deform_attn_func"]
    sys.path.insert(0, REF)
    sys.modules["MultiScaleDeformableAttention"] = types.ModuleType("MultiScaleDeformableAttention")

    tv = types.ModuleType("torchvision")
    tv.__version__ = "0.8.0"
    tvo = types.ModuleType("torchvision.ops")
    tvob = types.ModuleType("torchvision.ops.boxes")
    tvob.box_area = lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    tvm = types.ModuleType("torchvision.models")
    tvmu = types.ModuleType("torchvision.models._utils")
    tvmu.IntermediateLayerGetter = object
    tv.ops, tvo.boxes, tv.models, tvm._utils = tvo, tvob, tvm, tvmu
    sys.modules.update({"torchvision": tv, "torchvision.ops": tvo, "torchvision.ops.boxes": tvob,
                        "torchvision.models": tvm, "torchvision.models._utils": tvmu})

    mm = types.ModuleType("mmcv")
    mmo = types.ModuleType("mmcv.ops")

    class RoIAlign(torch.nn.Module):
        def __init__(self, output_size, spatial_scale=1.0, sampling_ratio=0, pool_mode="avg", aligned=True):
            super().__init__()
            self.output_size, self.spatial_scale = output_size, spatial_scale
            self.sampling_ratio, self.aligned = sampling_ratio, aligned

        def forward(self, x, rois):
            return roi_align_fn(x, rois, self.output_size, self.spatial_scale, self.sampling_ratio, self.aligned)

    mmo.RoIAlign = RoIAlign
    mm.ops = mmo
    sys.modules.update({"mmcv": mm, "mmcv.ops": mmo})

    pkg = types.ModuleType("models")
    pkg.__path__ = [REF + "/models"]
    sys.modules["models"] = pkg

    import models.ops.functions.ms_deform_attn_func as fmod
    import models.ops.modules.ms_deform_attn as mmod

    core = fmod.ms_deform_attn_core_pytorch

    class FlatShim:
        """Stands where the CUDA op stood: same call signature, CUDA flat indexing."""

        @staticmethod
        def apply(value, shapes, lsi, loc, aw, im2col_step):
            N, S, M, D = value.shape
            L, Lq, P = shapes.shape[0], loc.shape[1], loc.shape[4]
            loc = loc.reshape(-1)[: N * Lq * M * L * P * 2].view(N, Lq, M, L, P, 2)
            aw = aw.reshape(-1)[: N * Lq * M * L * P].view(N, Lq, M, L, P)
            return core(value, shapes, loc, aw)

    mmod.MSDeformAttnFunction = FlatShim
    fmod.FlatShim = FlatShim
    return fmod


def import_inference():
    """The reference's inference.py as a module (after ``install``).  It imports cv2, pycocotools, matplotlib, tqdm and
    torchvision.transforms at module level; none exists in this image and none is used by what the fixture generators call,
    so they are satisfied by EMPTY placeholder modules (import-only names)."""
    if "inference" in sys.modules:
        return sys.modules["inference"]

    def placeholder(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class Any_:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return self

    tv = sys.modules["torchvision"]
    tvt = placeholder("torchvision.transforms", ToPILImage=Any_, Resize=Any_, Compose=Any_, ToTensor=Any_, Normalize=Any_)
    tvf = placeholder("torchvision.transforms.functional", resize=lambda image, size: ("resized", tuple(size)))
    tvt.functional = tvf
    tv.transforms = tvt
    placeholder("cv2")
    placeholder("tqdm", tqdm=lambda x, *a, **k: x)
    mpl = placeholder("matplotlib")
    mpl.pyplot = placeholder("matplotlib.pyplot")
    pc = placeholder("pycocotools")
    pc.coco = placeholder("pycocotools.coco", COCO=Any_)
    pc.mask = placeholder("pycocotools.mask")
    ds = placeholder("datasets")
    ds.__path__ = []
    ds.coco_video_parser = placeholder("datasets.coco_video_parser", CocoVID=Any_)
    sys.modules["models"].build_model = None                 # inference.py does ``from models import build_model``
    import inference
    return inference
