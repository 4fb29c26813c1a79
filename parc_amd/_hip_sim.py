"""ctypes declarations of the simulator entry points (include/parc_sim.h)."""
import ctypes

from . import _hip

c_vp, c_int, c_f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float


def declare(L):
    L.parc_sim_abi.restype = c_int
    L.parc_sim_step.restype = c_int
    L.parc_sim_step.argtypes = [c_vp, c_vp, _hip.TerrainS, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_f]
    L.parc_sim_step_tick.restype = c_int
    L.parc_sim_step_tick.argtypes = [c_vp, c_vp, _hip.TerrainS, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_f, c_vp, c_vp, c_f]
    L.parc_sim_refresh_bodies.restype = c_int
    L.parc_sim_refresh_bodies.argtypes = [c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_vp]
    L.parc_sim_refresh_bodies_masked.restype = c_int
    L.parc_sim_refresh_bodies_masked.argtypes = [c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]
