"""flac-codec_amd: MI355X-native FLAC encode hot path behind the reference's
FlacSampleWriter / encode::Options surface (tuffy/flac-codec 1.3.2).

Import name: `flac_codec_amd` (a symlink to this directory, since `-` is not a valid
Python identifier).  Everything here drives hand-written gfx950 kernels through the C ABI
in include/*.h; there is no CPU fallback.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
