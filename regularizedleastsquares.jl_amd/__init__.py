"""MI355X (gfx950) backend for the iterative inner loop of RegularizedLeastSquares.jl.

Host-side mirror of the reference interface for the hot path only (CGNR / FISTA / ADMM on a dense
operator, BLAS-1, prox L1 / L2 / L21 / TV / Positive / Real); every numeric step runs in
librls_mi355x.so (hand-written HIP, C ABI in include/rls_mi355x.h).  Julia's `f!` is spelled `f_`.

The directory name contains a dot, so it cannot be imported with a plain `import` statement:
use the `rls_amd` shim at the repository root (`import rls_amd`).
"""
from ._lib import LIB_PATH, RLSError, load  # noqa: F401
from .arrays import (Context, DeviceMatrix, DeviceVector, NormalOperator, OperatorHandle, ProdOp, WeightingOp,  # noqa: F401
                     default_context, normalOperator)
from .regularization import (AbstractParameterizedRegularization, AbstractProjectionRegularization,  # noqa: F401
                             AbstractRegularization, GradientOp, L1Regularization, L2Regularization,
                             L21Regularization, LLRRegularization, MeasurementBasedNormalization, NoNormalization, NuclearRegularization,
                             PositiveRegularization, RealRegularization, SystemMatrixBasedNormalization,
                             TVRegularization, NormalizedRegularization, innerreg, lam, norm, normalize, prox_, scalefactor,
                             AbstractNestedRegularization, AbstractScaledRegularization, AutoScaledRegularization,
                             ClampedScalingTransform, FixedParameterRegularization, FixedScaledRegularization,
                             IdentityTransform, MaskedRegularization, MinMaxTransform, PlugAndPlayRegularization,
                             PnPRegularization, ProjectionRegularization, TransformedRegularization, ZTransform,
                             findfirst, findsink, findsinks, is_projection, sink, sinktype)
from .solvers import (ADMM, CGNR, DiagonalPreconditioner, FISTA, POGM, Kaczmarz, KaczmarzState, OptISTA, SplitBregman, AbstractKrylovSolver,
                      AbstractPrimalDualSolver, AbstractProximalGradientSolver, AbstractRowActionSolver,
                      applicableSolverList, isapplicable, AbstractLinearSolver, AdmmBatchedState, BatchedState, FistaBatchedState, CompareSolutionCallback, MultiThreadingState,  # noqa: F401
                      SequentialState, StoreConvergenceCallback, StoreSolutionCallback, createLinearSolver, init_,
                      iterate, linearSolverList, power_iterations, solve_, solve_group_, solverconvergence, solversolution,
                      solverstate)
from . import multigpu  # noqa: F401,E402
from .multigpu import (CommRowShardedADMM, CommRowShardedCGNR, CommRowShardedFISTA, ConcurrentSolves, MultiSolve, RowShardedADMM, RowShardedCGNR, RowShardedFISTA,  # noqa: F401,E402
                       shard_columns, shard_rows)
