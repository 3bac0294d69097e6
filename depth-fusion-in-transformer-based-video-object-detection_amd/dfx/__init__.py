"""dfx -- host-side plumbing of the MI355X deformable-attention path.

``dfx._lib`` loads the gfx950 C-ABI library (include/dfx_msda.h) with ctypes and
``dfx.ops`` hands torch CUDA(HIP) tensors to it as raw device pointers on torch's
current stream.  There is no CPU implementation here: every operator raises when
the library is missing or a tensor is not on the GPU.
"""
from ._lib import abi_version, library_path, load  # noqa: F401
