"""musediffusion_amd: MI355X-native hot path of MuseDiffusion (sampling loops + training losses
over the Transformer denoiser) behind the reference's `MuseDiffusion.models` plugin surface.

The tensor work runs in hand-written HIP kernels for gfx950 in `csrc/` (C-ABI: include/musehip.h),
loaded by `musediffusion_amd._lib`.  There is no CPU fallback: anything that needs a kernel raises
if the library is missing.
"""
__version__ = "0.1.0"
