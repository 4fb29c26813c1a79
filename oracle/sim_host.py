"""ctypes binding of the HOST build of the simulator core (oracle/sim_host.cpp) -- TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PARC_SIM_HOST_LIB: another build of the same sources (oracle/Makefile `sanitize`: libparc_sim_host_asan.so / _poison.so)
_LIB = os.environ.get("PARC_SIM_HOST_LIB") or os.path.join(_HERE, "_build", "libparc_sim_host.so")
# which formulation HostSim.step runs by default: "core" = one env per lane (parc_sim_core.h), "bpl" = the body-per-lane kernel
# the product launches (parc_sim_bpl.h) under the 16-fiber lane emulation of sim_host_bpl.cpp
DEFAULT_VARIANT = os.environ.get("PARC_SIM_HOST_VARIANT", "core")
_lib = None


class TerrainS(ctypes.Structure):
    _fields_ = [("hf", ctypes.c_void_p), ("dim_x", ctypes.c_int32), ("dim_y", ctypes.c_int32), ("min_x", ctypes.c_float),
                ("min_y", ctypes.c_float), ("dx", ctypes.c_float), ("dy", ctypes.c_float)]


def lib():
    global _lib
    if _lib is None:
        subprocess.check_call(["make", "-C", _HERE] + (["sanitize"] if "PARC_SIM_HOST_LIB" in os.environ else []), stdout=subprocess.DEVNULL)
        _lib = ctypes.CDLL(_LIB)
    return _lib


def set_fill(byte):
    """Work arrays of both formulations start from this byte pattern (0xFF = NaN); -1 = leave them as they come."""
    lib().sim_host_set_fill(ctypes.c_int(byte))
    lib().sim_host_bpl_set_fill(ctypes.c_int(byte))


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class HostSim:
    """State arrays in the Isaac Gym layouts, stepped on the CPU."""

    def __init__(self, model_struct, n, hf, min_point, dxdy, num_bodies=15, dof_size=28, variant=None):
        self.m = model_struct
        self.variant = variant or DEFAULT_VARIANT
        self.n, self.B, self.D = n, num_bodies, dof_size
        self.hf = np.ascontiguousarray(hf, dtype=np.float32)
        self.ter = TerrainS(_p(self.hf), self.hf.shape[0], self.hf.shape[1], float(min_point[0]), float(min_point[1]),
                            float(dxdy[0]), float(dxdy[1]))
        self.root_state = np.zeros((n, 13), np.float32)
        self.root_state[:, 6] = 1.0
        self.dof_state = np.zeros((n, dof_size, 2), np.float32)
        self.rigid_body_state = np.zeros((n, num_bodies, 13), np.float32)
        self.contact_forces = np.zeros((n, num_bodies, 3), np.float32)
        self.env_offsets = np.zeros((n, 3), np.float32)
        self.act_lo = np.full(dof_size, -10.0, np.float32)
        self.act_hi = np.full(dof_size, 10.0, np.float32)

    def step(self, action, n_sub=4, h=1.0 / 120.0):
        action = np.ascontiguousarray(action, dtype=np.float32)
        fn = lib().sim_host_step if self.variant == "core" else lib().sim_host_step_bpl
        fn(ctypes.byref(self.m), self.ter, self.n, _p(self.root_state), _p(self.dof_state), _p(self.rigid_body_state),
                            _p(self.contact_forces), _p(self.env_offsets), _p(action), _p(self.act_lo), _p(self.act_hi),
                            ctypes.c_int(n_sub), ctypes.c_float(h))

    def refresh_bodies(self):
        lib().sim_host_refresh_bodies(ctypes.byref(self.m), self.n, _p(self.root_state), _p(self.dof_state), _p(self.rigid_body_state),
                                      _p(self.contact_forces))

    def penetration(self):
        """[n, num_spheres] penetration depth of every collision sample sphere at the current root / dof state (0 = free)."""
        out = np.zeros((self.n, int(self.m.num_spheres)), np.float32)
        lib().sim_host_penetration(ctypes.byref(self.m), self.ter, self.n, _p(self.root_state), _p(self.dof_state), _p(self.env_offsets), _p(out))
        return out
