from .base import (PnPSolver, ADMMSolver, IADMMSolver, HQSSolver, PGSolver, APGSolver,  # noqa: F401
                   REDADMMSolver, AMPSolver)
