"""Rule library, variant "lrp": mirror of modules/layers_lrp.py (and the BERT copy).  Differs from
rules.py only in Linear (separate S1 = R/Z1, S2 = R/Z2: layers_lrp.py:199-200) and Add (plain
RelPropSimple without the per-sample rescale: layers_lrp.py:98-100)."""
from . import rules as _r
from .rules import *  # noqa: F401,F403
from .rules import __all__  # noqa: F401
from .rules import StopRelprop  # noqa: F401


class RelProp(_r.RelProp):
    variant = "lrp"


class Linear(_r.Linear):
    variant = "lrp"


class Add(_r.Add):
    variant = "lrp"


class einsum(_r.einsum):
    variant = "lrp"


class MatMul(_r.MatMul):
    variant = "lrp"
