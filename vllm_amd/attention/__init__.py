from .paged_attn import PagedAttention, PagedAttentionMetadata  # noqa: F401
