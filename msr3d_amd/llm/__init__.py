"""Language-model side of MSR3D's training step: the fused per-sequence
cross-entropy, the LoRA-augmented linear layer, one LoRA-Llama decoder layer assembled from them, and the stack
(layers + final norm + head + loss) a training step runs."""
from .decoder import LoRALlamaDecoderLayer  # noqa: F401
from .lora import LoRALinear  # noqa: F401
from .losses import seq_mean_cross_entropy  # noqa: F401
from .stack import FrozenLinear, LoRALlamaStack  # noqa: F401
from .checkpoint import hf_state_dict, load_hf_state_dict, peft_adapter_state_dict, reference_trainer_state_dict  # noqa: F401
