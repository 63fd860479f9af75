"""Language-model side of MSR3D's training step: the self-contained pieces
built so far -- the fused per-sequence cross-entropy and the LoRA-augmented linear layer."""
from .lora import LoRALinear  # noqa: F401
from .losses import seq_mean_cross_entropy  # noqa: F401
