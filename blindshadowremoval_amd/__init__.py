"""MI355X-native GSC shadow-removal generator forward (drop-in for the reference's ``Generator`` /
``FSRNet.test*`` inference path).  See DESIGN.md."""
from .weights import generator_variable_shapes, init_weights  # noqa: F401

__all__ = ["Generator", "GeneratorTSM", "generator_variable_shapes", "init_weights"]


def __getattr__(name):
    if name in ("Generator", "GeneratorTSM"):
        from . import model
        return getattr(model, name)
    raise AttributeError(name)
