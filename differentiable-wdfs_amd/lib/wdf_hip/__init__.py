"""wdf_hip -- host side of the MI355X WDF engine (ctypes -> libwdf_hip.so)."""
from . import binding  # noqa: F401
from .binding import WdfHipError  # noqa: F401
