"""How the records exchanged between the tasks and the simulator adapters are built.

Attribute names, defaults and order are those of the reference's dataclasses (rsoccer_gym/Entities/Ball.py:3-10, Robot.py:4-23,
Field.py:3-21) — task code reads and writes them by name and ``Field(**simulator.get_field_params())`` needs exactly the 17 keys.
Each class is built from a field table in the module the reference keeps it in (``Entities/Ball.py``, ``Robot.py``, ``Field.py``:
one place to see the layout, units alongside); this module holds the builder.
"""
from dataclasses import field, make_dataclass
from typing import Optional

OptFloat = Optional[float]


def record(name, module, spec, doc):
    """a dataclass ``name`` living in ``module`` from rows of (attribute, type, default)"""
    cls = make_dataclass(name, [(n, t, field(default=d)) for n, t, d in spec])
    cls.__doc__ = doc
    cls.__module__ = module
    return cls
