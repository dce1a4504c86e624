"""umfa_torch -- PyTorch-ROCm binding of the MI355X flash-attention kernels (in-stream, zero-copy)."""
from .ops import attention_encode, attention_forward, context, last_kernel

__all__ = ["attention_forward", "attention_encode", "context", "last_kernel"]
