from .block_table import BlockTable, MultiGroupBlockTable  # noqa: F401
