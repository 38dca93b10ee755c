from . import functional, solver, kernel, corrector, strategy, scheduler
from .optimizer import GaussNewton
from .optimizer import GaussNewton as GN
from .optimizer import LevenbergMarquardt
from .optimizer import LevenbergMarquardt as LM
from .posegraph import PCG
solver.PCG = PCG


def freeze_gc():
    """Keep Python's cyclic garbage collector out of the optimisation loop.

    With torch imported a full (generation-2) collection walks ~10^6 live objects: a 40-70 ms pause, i.e. tens of
    LM steps at this library's speed (measured on InvNet, 10^6 problems: 60 steps in 92 ms with the collector's
    default behaviour, 18 ms without the pauses).  ``gc.freeze()`` moves everything alive *now* into the permanent
    generation, so later collections only scan objects created afterwards.  Call it once after the models and
    optimizers are built; objects frozen here are still released by reference counting, only reference *cycles*
    among them are no longer reclaimed."""
    import gc
    gc.collect()
    gc.freeze()
