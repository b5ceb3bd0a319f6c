"""``Field`` — rsoccer_gym/Entities/Field.py:3-21: the 17 floats of ``get_field_params()`` (rsim.py:49-50 builds it as
``Field(**dict)``, so exactly these keys, all required, in the order of the C-ABI's table — include/rsx.h: rsx_get_field_params)."""
from dataclasses import make_dataclass

from rsoccer_amd._lib import FIELD_KEYS


def _field_class():
    cls = make_dataclass("Field", [(k, float) for k in FIELD_KEYS])   # no defaults: a missing key is an error, as in the reference
    cls.__doc__ = "Field and robot geometry as returned by get_field_params(): " + ", ".join(FIELD_KEYS) + "."
    cls.__module__ = __name__
    return cls


Field = _field_class()
