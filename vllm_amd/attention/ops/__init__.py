from .prefix_prefill import context_attention_fwd  # noqa: F401
from .triton_unified_attention import unified_attention  # noqa: F401
