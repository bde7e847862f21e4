"""`qpth`-compatible import surface: put `deq-mpc-corl_amd/` in front of
the reference's tree on sys.path and `import qpth.AL_mpc as al_mpc`,
`import qpth.al_utils as al_utils` in deqmpc/policies.py:5-8 resolve to this package."""
from . import al_utils  # noqa: F401
from . import AL_mpc  # noqa: F401
