"""CPU-side checks of the C-ABI shared library: it loads, and it exports every symbol the
headers in include/ declare (no compute calls here: there is no GPU on the CPU box)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(flac(?:gpu|enc)_[a-z0-9_]+)\s*\(", text))


def test_library_loads_and_exports_every_declared_symbol():
    from flac_codec_amd import _lib

    _lib.lib()
    exported = _lib.exported_symbols()
    for header in os.listdir(os.path.join(ROOT, "include")):
        if not header.endswith(".h"):
            continue
        missing = declared_functions(header) - exported
        assert not missing, f"{header}: not exported: {sorted(missing)}"


def test_struct_layouts_match_header():
    import ctypes as C

    from flac_codec_amd import _lib

    assert C.sizeof(_lib.SubframePlan) == 12 + 12 + 128 + 64 + 64
    assert C.sizeof(_lib.FramePlan) == 8
    assert C.sizeof(_lib.GpuOptions) == 20
