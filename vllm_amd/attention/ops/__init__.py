from .prefix_prefill import context_attention_fwd  # noqa: F401
