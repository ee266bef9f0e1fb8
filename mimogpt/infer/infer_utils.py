"""Drop-in module path of the reference config helper; the implementation is selftoktokenizer_amd.config."""
from selftoktokenizer_amd.config import AttrDict as EasyDict, parse_args_from_yaml  # noqa: F401

__all__ = ["parse_args_from_yaml"]
