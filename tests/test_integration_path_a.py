"""INTEGRATION.md path A, executed in the build container: the REFERENCE's own ``models/ops/functions/ms_deform_attn_func.py``
and ``models/ops/modules/ms_deform_attn.py`` import this repository's ``MultiScaleDeformableAttention.py`` (the ctypes shim over
libdfx.so) where they import their compiled pybind11 module (/root/reference/models/ops/functions/ms_deform_attn_func.py:18-38),
nothing of the reference changed: the module resolves to the shim, both entry points are found with the header's parameter
lists (ms_deform_attn.h:20-38,41-61), and CPU tensors raise the reference's "Not implemented on the CPU" through the reference's
autograd Function and through its MSDeformAttn module.  (On a GPU the same import runs the HIP kernels; the numerics of that
path are tests/test_msda_gpu.py's.)  Build container only: /root/reference does not exist on the GPU box - skipped there.

Runs in a subprocess: the reference's ``models`` package must not meet this repository's in one interpreter.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "depth-fusion-in-transformer-based-video-object-detection_amd")
REF = "/root/reference"

SCRIPT = r"""
import inspect, os, sys, types
REF, PKG = sys.argv[1], sys.argv[2]
# what `cd <reference> && PYTHONPATH=<PKG> python main_multi.py` gives: the script's directory first, then PYTHONPATH
sys.path[:0] = [REF, PKG]
pkg = types.ModuleType("models")            # models/__init__.py pulls torchvision / mmcv (absent here): namespace import
pkg.__path__ = [REF + "/models"]
sys.modules["models"] = pkg
import torch
import models.ops.functions.ms_deform_attn_func as f          # the REFERENCE's file: `import MultiScaleDeformableAttention as MSDA`
from models.ops.modules import MSDeformAttn                    # the REFERENCE's module
assert f.__file__.startswith(REF) and inspect.getfile(MSDeformAttn).startswith(REF)
assert f.MSDA.__file__ == os.path.join(PKG, "MultiScaleDeformableAttention.py"), f.MSDA.__file__
fwd, bwd = f.MSDA.ms_deform_attn_forward, f.MSDA.ms_deform_attn_backward
assert list(inspect.signature(fwd).parameters) == ["value", "spatial_shapes", "level_start_index", "sampling_loc", "attn_weight", "im2col_step"]
assert list(inspect.signature(bwd).parameters) == ["value", "spatial_shapes", "level_start_index", "sampling_loc", "attn_weight", "grad_output", "im2col_step"]
v = torch.zeros(1, 4, 8, 32); s = torch.tensor([[2, 2]]); l = torch.tensor([0])
loc = torch.zeros(1, 3, 8, 1, 4, 2); aw = torch.zeros(1, 3, 8, 1, 4)
for call in (lambda: f.MSDeformAttnFunction.apply(v, s, l, loc, aw, 64),
             lambda: bwd(v, s, l, loc, aw, torch.zeros(1, 3, 256), 64),
             lambda: MSDeformAttn(256, 1, 8, 4)(torch.zeros(1, 3, 256), torch.rand(1, 3, 1, 2), torch.zeros(1, 4, 256), s, l)):
    try:
        call()
    except RuntimeError as e:
        assert "Not implemented on the CPU" in str(e), str(e)
    else:
        raise AssertionError("CPU tensors must be refused")
print("PATH_A_OK")
"""


@pytest.mark.skipif(not os.path.isdir(REF), reason="build container only: the reference is not on the GPU box")
def test_reference_modules_import_the_shim_unchanged():
    out = subprocess.run([sys.executable, "-c", SCRIPT, REF, PKG], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "PATH_A_OK" in out.stdout, out.stderr[-2000:]
