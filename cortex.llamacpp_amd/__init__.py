"""cortex.llamacpp_amd — MI355X-native GGUF inference backend (hand-written HIP for gfx950).

The product is the C-ABI shared library `lib/libmi355_llama.so` (include/mi355_llama.h); this package
is the ctypes view of it used by tests, bench.py and the smoke entry point.  There is no CPU
fallback: importing works anywhere, but every compute call raises without the library and a GPU.
"""
from . import binding, gguf_synth  # noqa: F401
from .binding import Backend, Clip, Context, Engine, Model, MI355Error, lib_path, load_library  # noqa: F401
