"""The consumer of the SALSA features: the SELD CRNN (PANN-ResNet22-style encoder + 2-layer BiGRU + SED / DOA heads) on
PyTorch-ROCm, bf16 autocast + channels-last, data-parallel over the GPUs of a node with RCCL gradient all-reduce.
This is a "next" row of SURVEY.md section 8(f): it runs on MIOpen/rocBLAS through torch, not on hand-written kernels."""
from .model import SeldCRNN, interpolate_tensor  # noqa: F401
from .loss import seld_loss  # noqa: F401
