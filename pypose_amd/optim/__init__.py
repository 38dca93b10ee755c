from . import functional, solver, kernel, corrector, strategy, scheduler
from .optimizer import GaussNewton
from .optimizer import GaussNewton as GN
from .optimizer import LevenbergMarquardt
from .optimizer import LevenbergMarquardt as LM
from .posegraph import PCG
solver.PCG = PCG
