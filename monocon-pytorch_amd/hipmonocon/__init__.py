import os as _os

# Multi-process GPU work on this platform needs dmabuf IPC (RCCL's peer mappings, CUDA tensors shared across processes):
# without it hipIpcGetMemHandle fails with "invalid argument".  The runtime reads the variable when it initialises, i.e. at the
# first device call of the process, so it is set here -- the package is imported before its library touches a device -- and
# only when the caller's environment does not say otherwise.
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
