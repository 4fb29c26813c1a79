"""The simulator ON THE DEVICE behind the interface of oracle.sim_host.HostSim (numpy state arrays in the Isaac Gym layouts, step(),
refresh_bodies(), the model struct as `.m`), so that tests/test_sim_invariants.py holds the PRODUCT kernel - sim_step_bpl_kernel through
parc_sim_step of the C ABI - to the same closed forms, conservation laws and to the independent float64 inverse dynamics as the two
host builds.  The model struct is uploaded at every call: a test that edits `.m` (gains, gravity, radii) is seen by the next step."""
import numpy as np
import torch

from parc_amd import _hip

DEV = "cuda:0"


class DeviceSim:
    variant = "device"

    def __init__(self, model_struct, n, hf, min_point, dxdy, num_bodies=15, dof_size=28):
        self.m = model_struct
        self.n, self.B, self.D = n, num_bodies, dof_size
        self.hf = np.ascontiguousarray(hf, dtype=np.float32)
        self._hf = torch.tensor(self.hf, device=DEV)
        self._ter = _hip.terrain_struct(self._hf, [float(min_point[0]), float(min_point[1])], [float(dxdy[0]), float(dxdy[1])])
        self.root_state = np.zeros((n, 13), np.float32)
        self.root_state[:, 6] = 1.0
        self.dof_state = np.zeros((n, dof_size, 2), np.float32)
        self.rigid_body_state = np.zeros((n, num_bodies, 13), np.float32)
        self.contact_forces = np.zeros((n, num_bodies, 3), np.float32)
        self.env_offsets = np.zeros((n, 3), np.float32)
        self.act_lo = np.full(dof_size, -10.0, np.float32)
        self.act_hi = np.full(dof_size, 10.0, np.float32)

    def _model(self):
        return torch.frombuffer(bytearray(bytes(self.m)), dtype=torch.uint8).to(DEV)

    def _up(self, a):
        # (a dof-less model still hands the kernels valid pointers: one spare element)
        t = torch.tensor(np.ascontiguousarray(a, dtype=np.float32), device=DEV)
        return t if t.numel() else torch.zeros(4, device=DEV)

    def step(self, action, n_sub=4, h=1.0 / 120.0):
        m = self._model()
        rs, ds, rb, cf = self._up(self.root_state), self._up(self.dof_state), self._up(self.rigid_body_state), self._up(self.contact_forces)
        eo, act, lo, hi = self._up(self.env_offsets), self._up(action), self._up(self.act_lo), self._up(self.act_hi)
        assert rs.shape == (self.n, 13) and rb.shape == (self.n, self.B, 13) and cf.shape == (self.n, self.B, 3) and eo.shape == (self.n, 3)
        assert self.D == 0 or (ds.shape == (self.n, self.D, 2) and act.shape == (self.n, self.D))
        p = _hip.ptr
        _hip.check(_hip.lib().parc_sim_step(_hip.stream(), _hip.c_vp(m.data_ptr()), self._ter, self.n, p(rs), p(ds), p(rb), p(cf), p(eo), p(act),
                                            p(lo), p(hi), int(n_sub), float(h)), "parc_sim_step")
        torch.cuda.synchronize()
        self.root_state[:] = rs.cpu().numpy()
        if self.D:
            self.dof_state[:] = ds.cpu().numpy()
        self.rigid_body_state[:] = rb.cpu().numpy()
        self.contact_forces[:] = cf.cpu().numpy()

    def refresh_bodies(self):
        m = self._model()
        rs, ds, rb, cf = self._up(self.root_state), self._up(self.dof_state), self._up(self.rigid_body_state), self._up(self.contact_forces)
        p = _hip.ptr
        _hip.check(_hip.lib().parc_sim_refresh_bodies(_hip.stream(), _hip.c_vp(m.data_ptr()), self.n, None, 0, p(rs), p(ds), p(rb), p(cf)),
                   "parc_sim_refresh_bodies")
        torch.cuda.synchronize()
        self.rigid_body_state[:] = rb.cpu().numpy()
        self.contact_forces[:] = cf.cpu().numpy()
