"""Drop-in for wdf_py/diode_clipper/diode_config.py: the named diode configurations
(diode_config.py:5-31).  Is / nabla of the 1N4148 are the SPICE-model values the reference
quotes there; N_up / N_down count the series diodes in each direction."""
from collections import namedtuple

DiodeConfig = namedtuple("DiodeConfig", ["name", "Is", "nabla", "Vt", "N_up", "N_down"],
                         defaults=["", 1.0e-9, 1.0, 25.85e-3, 1, 1])

default_diode = DiodeConfig("DefaultDiode")


def _1n4148(n_up, n_down):
    return DiodeConfig(f"1N4148 ({n_up}U-{n_down}D)", Is=4.352e-9, nabla=1.906, N_up=n_up, N_down=n_down)


diode_1n4148_1u1d = _1n4148(1, 1)
diode_1n4148_1u2d = _1n4148(1, 2)
diode_1n4148_1u3d = _1n4148(1, 3)
diode_1n4148_2u2d = _1n4148(2, 2)
diode_1n4148_2u3d = _1n4148(2, 3)
diode_1n4148_3u3d = _1n4148(3, 3)
