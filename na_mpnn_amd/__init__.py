"""NA-MPNN encoder / decoder hot path on MI355X (hand-written HIP behind the reference's nn.Module surface)."""
import os as _os

# Kernel-argument segments in DEVICE memory.  The HIP runtime places them in host memory by default, so every scalar load of a
# kernel argument that the compiler did not hoist to the kernel's first instructions is a PCIe round trip; the kernels of this
# package take their arguments as by-value structs of 0.6 - 1.6 KB that are read at the point of use.  Measured on the cfg2
# forward: 0.523 -> 0.499 ms exact fp32, 0.332 -> 0.311 ms split-bf16 (profiles/r03c).  The runtime reads the variable when it
# initialises (the first HIP call of the process), so this takes effect when the package is imported before that;
# NAMP_KEEP_HOST_KERNARG=1 leaves the runtime's default alone.
# The variable is process-wide (torch's own kernels see it too) and only takes effect if set before the HIP runtime initialises:
# importing this package AFTER a HIP call leaves the runtime's default in force — a warning says so (README "Environment").
if _os.environ.get("NAMP_KEEP_HOST_KERNARG") != "1":
    if "HIP_FORCE_DEV_KERNARG" not in _os.environ:
        import sys as _sys
        _t = _sys.modules.get("torch")
        if _t is not None and getattr(getattr(_t, "cuda", None), "is_initialized", lambda: False)():
            import warnings as _w
            _w.warn("na_mpnn_amd imported after the HIP runtime was initialised: HIP_FORCE_DEV_KERNARG=1 cannot take effect any more "
                    "(kernel arguments stay in host memory: ~5 % slower small-batch forwards). Import na_mpnn_amd first, or export the "
                    "variable; NAMP_KEEP_HOST_KERNARG=1 silences this.", RuntimeWarning, stacklevel=2)
    _os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
