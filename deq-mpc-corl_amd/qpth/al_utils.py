"""Problem containers with the reference's names (qpth/al_utils.py:8-13).

``QuadCost(C, c, f)``: C is [B,T,n,n] (only its diagonal is used, AL_mpc.py:250),
c is [B,T,n], f is [B,T] (constant term, logging only).
``LinDx(F, f)``: affine dynamics x_{t+1} = F_t [x_t;u_t] + f_t with F [B,T-1,nx,n],
f [B,T-1,nx] (batch-first, as AL_mpc.MPC.rollout indexes it, AL_mpc.py:527-529).
"""
from collections import namedtuple

QuadCost = namedtuple("QuadCost", "C c f")
LinDx = namedtuple("LinDx", "F f")

QuadCost.__new__.__defaults__ = (None,) * len(QuadCost._fields)
LinDx.__new__.__defaults__ = (None,) * len(LinDx._fields)
