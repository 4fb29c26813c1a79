"""The two clip edits between stage 2's optimiser and the tracker's dataset (zmotion_editing_tools/motion_edit_lib.py of the reference:
flip_motion_about_XZ_plane :514-610, remove_hesitation_frames :1242-1319) against fixture G22 (reference outputs on CPU)."""
import pytest
import torch

from conftest import golden
from test_hip_parity import DEV, T, close, km  # noqa: F401  (km is a fixture)

pytestmark = pytest.mark.gpu


def test_g22_mirrored_clip(km):
    from parc_amd.zmotion_editing_tools import motion_edit_lib as medit
    g = golden("g22_motion_edit")
    f, c = medit.flip_motion_about_XZ_plane(T(g["frames"]), km, contact_frames=T(g["contacts"]))
    close(c, g["flip_contacts"], atol=0, rtol=0)
    close(f[:, 0:6], g["flip_frames"][:, 0:6], atol=0, rtol=0)
    # dofs come back from a quaternion -> dof map: compare the rotations they stand for (exponential maps of angle > pi have
    # a shorter equivalent) and the values themselves where they agree
    jr = km.dof_to_rot(f[:, 6:].contiguous())
    jr_ref = km.dof_to_rot(T(g["flip_frames"][:, 6:]))
    dot = (jr * jr_ref).sum(-1).abs()
    assert float((1.0 - dot).max()) < 1e-6
    close(f[:, 6:], g["flip_frames"][:, 6:], atol=3e-5, rtol=0)
    # mirroring twice is the identity on poses
    f2, c2 = medit.flip_motion_about_XZ_plane(f, km, contact_frames=c)
    close(c2, g["contacts"], atol=0, rtol=0)
    close(f2[:, 0:6], g["frames"][:, 0:6], atol=0, rtol=0)
    jr2, jr0 = km.dof_to_rot(f2[:, 6:].contiguous()), km.dof_to_rot(T(g["frames"][:, 6:]))
    assert float((1.0 - (jr2 * jr0).sum(-1).abs()).max()) < 1e-6
    # a clip handed over on the host comes back on the host (parc_2_kin_gen.py:469 passes CPU tensors)
    fh = medit.flip_motion_about_XZ_plane(torch.tensor(g["frames"]), km)
    assert fh.device.type == "cpu" and torch.allclose(fh, f.cpu(), atol=0)


def test_g22_hesitation_frames_removed(km):
    from parc_amd.zmotion_editing_tools import motion_edit_lib as medit
    g = golden("g22_motion_edit")
    nf, nc = medit.remove_hesitation_frames(T(g["hes_frames"]), T(g["hes_contacts"]), km)
    assert nf.shape == g["hes_out_frames"].shape == (52, 34)                 # the 8-frame stretch goes, the 3-frame one stays
    close(nf, g["hes_out_frames"], atol=0, rtol=0)
    close(nc, g["hes_out_contacts"], atol=0, rtol=0)
    nf2, _ = medit.remove_hesitation_frames(T(g["hes_frames"]), T(g["hes_contacts"]), km, hesitation_val=0.3, hesitation_min_seq_len=2)
    assert nf2.shape[0] == int(g["hes_out2_count"][0])
    same, _ = medit.remove_hesitation_frames(T(g["frames"]), T(g["contacts"]), km, hesitation_val=1e-4)
    assert same.shape[0] == g["frames"].shape[0]
