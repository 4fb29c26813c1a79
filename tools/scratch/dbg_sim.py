import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from parc_amd import _hip
from parc_amd.sim_model import SimModel
from parc_amd.anim.kin_char_model import KinCharModel
from parc_amd.assets import humanoid_spec
from oracle.sim_host import HostSim
DEV="cuda:0"
def T(x, dtype=torch.float32): return torch.tensor(np.asarray(x), dtype=dtype, device=DEV)
km = KinCharModel(DEV); km.load_char_file(humanoid_spec.write_mjcf())
sm = SimModel(km)
L=_hip.lib()
def run(case, n=4, steps=1, nsub=1):
    rng = np.random.default_rng(2)
    hf = np.full((40,40), -100.0 if case=="fall" else 0.0, np.float32)
    host = HostSim(sm.struct, n, hf, [-4.0,-4.0],[0.4,0.4])
    host.root_state[:,2]=1.2
    if case in ("vel","all"):
        host.root_state[:,7:13]=rng.standard_normal((n,6))*0.3
        host.dof_state[:,:,1]=rng.standard_normal((n,28))*0.5
    if case in ("pos","all"):
        host.dof_state[:,:,0]=rng.standard_normal((n,28))*0.2
    rs, ds = T(host.root_state.copy()), T(host.dof_state.copy())
    rb, cf = torch.zeros((n,15,13),device=DEV), torch.zeros((n,15,3),device=DEV)
    eo, lo, hi = T(host.env_offsets), T(host.act_lo), T(host.act_hi)
    d_hf=T(hf); ter=_hip.terrain_struct(d_hf,[-4.0,-4.0],[0.4,0.4])
    for s in range(steps):
        act=np.zeros((n,28),np.float32)
        host.step(act,n_sub=nsub,h=1/120)
        a=T(act)
        _hip.check(L.parc_sim_step(_hip.stream(), sm.device_ptr(DEV), ter, n, _hip.ptr(rs), _hip.ptr(ds), _hip.ptr(rb), _hip.ptr(cf), _hip.ptr(eo), _hip.ptr(a), _hip.ptr(lo), _hip.ptr(hi), nsub, 1/120), "x")
        torch.cuda.synchronize()
    print(case, "root", np.abs(rs.cpu().numpy()-host.root_state).max(), "dof", np.abs(ds.cpu().numpy()-host.dof_state).max(), "rb", np.abs(rb.cpu().numpy()-host.rigid_body_state).max())
    if np.abs(rs.cpu().numpy()-host.root_state).max()>1e-3:
        print(" dev", rs[0].cpu().numpy()); print(" host", host.root_state[0])
for c in ("fall","vel","pos","all"):
    run(c)
run("ground", steps=3, nsub=4)

def run1(dofs, val=0.3, n=1, nsub=1, vel=False):
    hf = np.full((40,40), -100.0, np.float32)
    host = HostSim(sm.struct, n, hf, [-4.0,-4.0],[0.4,0.4])
    host.root_state[:,2]=1.2
    for d in dofs:
        host.dof_state[:,d,1 if vel else 0]=val
    rs, ds = T(host.root_state.copy()), T(host.dof_state.copy())
    rb, cf = torch.zeros((n,15,13),device=DEV), torch.zeros((n,15,3),device=DEV)
    eo, lo, hi = T(host.env_offsets), T(host.act_lo), T(host.act_hi)
    d_hf=T(hf); ter=_hip.terrain_struct(d_hf,[-4.0,-4.0],[0.4,0.4])
    act=np.zeros((n,28),np.float32)
    host.step(act,n_sub=nsub,h=1/120)
    a=T(act)
    _hip.check(L.parc_sim_step(_hip.stream(), sm.device_ptr(DEV), ter, n, _hip.ptr(rs), _hip.ptr(ds), _hip.ptr(rb), _hip.ptr(cf), _hip.ptr(eo), _hip.ptr(a), _hip.ptr(lo), _hip.ptr(hi), nsub, 1/120), "x")
    torch.cuda.synchronize()
    dd = np.abs(ds.cpu().numpy()-host.dof_state)[0]
    print("dofs", dofs, "vel" if vel else "pos", "root diff", np.abs(rs.cpu().numpy()-host.root_state).max(), "dof diff max", dd.max(), "at", np.unravel_index(dd.argmax(), dd.shape))
    if dd.max()>1e-3:
        print("  dev vel ", ds[0,:,1].cpu().numpy().round(3)); print("  host vel", host.dof_state[0,:,1].round(3))
print("---- single-dof probes")
run1([17]); run1([9]); run1([14]); run1([15]); run1([0]); run1([0,1,2]); run1([3]); run1([6]); run1([18])
run1([17], vel=True); run1([14], vel=True)
