"""Stand-in for aenum: the stdlib enum plus an ``extend_enum`` that refuses (never reached by default)."""
from enum import *  # noqa: F401,F403
from enum import Enum, EnumMeta, IntEnum  # noqa: F401


def extend_enum(enumeration, name, *args, **kwargs):
    raise NotImplementedError("extend_enum is not available in the aenum stand-in")
