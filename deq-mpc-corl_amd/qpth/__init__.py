"""`qpth`-compatible import surface: the modules deqmpc/policies.py:5-8 imports -
`qpth.qp_wrapper`, `qpth.AL_mpc`, `qpth.AL_mpc_custom.Obstacle_MPC`, `qpth.al_utils` - all resolve to this
package. Use the module shadowing of INTEGRATION.md section 1 (sys.modules entries named `qpth...`), or
put a directory holding a `qpth` symlink to this directory in front of the reference's tree on sys.path."""
from . import al_utils  # noqa: F401
from . import AL_mpc  # noqa: F401
from . import AL_mpc_custom  # noqa: F401
from . import qp_wrapper  # noqa: F401
