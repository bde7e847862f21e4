"""MI355X-native batched augmented-Lagrangian MPC/QP solver (drop-in for the
``qpth.AL_mpc`` path of anonymous-author-918/deq-mpc-corl).

This module is the importable alias of the ``deq-mpc-corl_amd/`` directory (a
hyphen cannot appear in a Python module name); all code lives there.
"""
import os as _os

__path__.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                                 "deq-mpc-corl_amd"))

from .qpth.al_utils import LinDx, QuadCost  # noqa: E402,F401
from .qpth.AL_mpc import MPC  # noqa: E402,F401
from .problems import AffineDynamics, PendulumDynamics, synthetic_problem  # noqa: E402,F401
from .dynamics import (Cartpole1lDynamics, Cartpole1lV2Dynamics, Cartpole2lDynamics, FlyingCartpoleDynamics,  # noqa: E402,F401
                       Pendulum1lDynamics, RexQuadrotorDynamics)
