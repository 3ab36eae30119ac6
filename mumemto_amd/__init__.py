"""mumemto_amd -- MI355X-native multi-MUM / multi-MEM engine behind the
mumemto_library C API.  The Python layer is a thin ctypes mirror of the
reference's interface for the hot path (mumemto_library/mumemto_api.hpp:29-57,
python_bindings/): same function names, argument meaning and error behaviour.
All compute happens in mumemto_amd/lib/libmumemto.so (hand-written HIP for
gfx950).  There is no CPU fallback: if the library is missing or no GPU is
usable, calls raise.
"""
from .binding import (  # noqa: F401
    Comm,
    DevicePartition,
    Engine,
    MumemtoError,
    Params,
    library_path,
    load_library,
    mumemto_mem,
    mumemto_mum,
)

__version__ = "0.1.0"
