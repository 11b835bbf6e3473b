from .triton_attn import (TritonAttentionBackend, TritonAttentionImpl,  # noqa: F401
                          TritonAttentionMetadata, TritonAttentionMetadataBuilder)
