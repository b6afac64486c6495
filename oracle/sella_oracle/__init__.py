"""ORACLE package (test infrastructure): NumPy/SciPy restatement of the Sella hot path."""
from .mgs import modified_gram_schmidt, mgs_inplace            # noqa: F401
from .secant import symmetrize_Y, update_H                       # noqa: F401
from .davidson import exact, rayleigh_ritz, correction           # noqa: F401
from .hessian_ops import (FiniteDifferenceHessian, OperatorSum,  # noqa: F401
                          QuasiNewtonHessian)
from .stepsolve import (get_stepper, get_restricted_step,        # noqa: F401
                        QuasiNewtonStep, RFOStep, PRFOStep,
                        TrustRegionStep, PerAtomStep, NaiveStep,
                        QuasiNewtonIRCStep, IRCTrustRegionStep, MaxInternalStepOracle)
from . import sparse_internal                                    # noqa: F401
