"""The consumer of the SALSA features: the SELD CRNN (PANN-ResNet22-style encoder + 2-layer BiGRU + SED / DOA heads) on
PyTorch-ROCm, bf16 autocast + channels-last, data-parallel over the GPUs of a node with RCCL gradient all-reduce.
Convolutions, BatchNorm, pools and the GRU scans run on the hand-written HIP kernels of salsa_amd/csrc (include/salsa_nn.h,
include/salsa_gru.h); torch supplies autograd, the decoder GEMMs (hipBLASLt), the optimizer and DDP over RCCL.  Reference
checkpoints load through SeldCRNN.load_reference_state_dict (crnn/checkpoint.py)."""
from .model import SeldCRNN, interpolate_tensor  # noqa: F401
from .loss import seld_loss  # noqa: F401
