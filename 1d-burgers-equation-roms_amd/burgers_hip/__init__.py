"""burgers_hip -- host side of the MI355X batched Burgers FOM/ROM time-stepper.

lib    ctypes binding of libburgers_hip.so (C ABI in include/burgers_hip.h); no CPU fallback
build  in-tree hipcc build (gfx950)
fom    batched FOM (bg_fom_run) on torch device tensors
rom    batched POD / quadratic-manifold / POD-ANN steppers (bg_rom_reduce, bg_lu_solve)
pod    offline bases (SVD + energy truncation, quadratic ridge fit) and .npy contracts
dist   mu-sharding across ranks and the snapshot all-gather (RCCL on GPU, gloo on CPU)
"""
__all__ = ["lib", "build", "fom", "rom", "pod", "dist"]
