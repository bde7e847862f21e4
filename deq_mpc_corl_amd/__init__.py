"""Importable alias of the ``deq-mpc-corl_amd/`` package directory (a hyphen cannot
appear in a Python module name). All code lives in ``deq-mpc-corl_amd/``."""
import os as _os

__path__.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                                 "deq-mpc-corl_amd"))
