from enum import Enum, IntEnum  # noqa: F401  (import-only stand-in)
