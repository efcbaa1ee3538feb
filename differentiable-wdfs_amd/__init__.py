"""differentiable-wdfs_amd -- MI355X-native engine for the wdf_py hot path.

The directory name is the project's; it is not a Python identifier, so import it with
    import importlib; wdfamd = importlib.import_module("differentiable-wdfs_amd")
or, as the reference scripts do with wdf_py/lib (lpf.py:5-9), put
`differentiable-wdfs_amd/lib` on sys.path and `import tf_wdf as wdf`.
"""
import os
import sys

LIB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib")
if LIB_DIR not in sys.path:
    sys.path.insert(0, LIB_DIR)

import wdf_hip  # noqa: E402,F401
from wdf_hip import binding, engine  # noqa: E402,F401
