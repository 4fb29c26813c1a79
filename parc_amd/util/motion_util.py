"""Batches of kinematic poses as named tensors (the exchange format between the env, the motion generator and the editing tools).

Mirror of the reference's ``util/motion_util.py`` (MotionFrames :6-143, cat_motion_frames :145-193, motion_frames_from_mlib_format
:195-221): six optional fields, every operation applies to the fields that are present.
"""
import torch

from . import torch_util

FIELDS = ("root_pos", "root_rot", "joint_rot", "body_pos", "body_rot", "contacts")


class MotionFrames:
    def __init__(self, root_pos=None, root_rot=None, joint_rot=None, body_pos=None, body_rot=None, contacts=None):
        self.root_pos, self.root_rot, self.joint_rot = root_pos, root_rot, joint_rot
        self.body_pos, self.body_rot, self.contacts = body_pos, body_rot, contacts

    def _map(self, fn):
        return MotionFrames(**{f: (None if getattr(self, f) is None else fn(getattr(self, f))) for f in FIELDS})

    def init_blank_frames(self, char_model, history_length, batch_size=1):
        """identity poses [batch, history, ...] on the model's device"""
        dev, nb = char_model._device, char_model.get_num_joints()
        z = lambda *shape: torch.zeros((batch_size, history_length) + shape, dtype=torch.float32, device=dev)
        self.root_pos, self.root_rot, self.joint_rot = z(3), z(4), z(nb - 1, 4)
        self.body_pos, self.body_rot, self.contacts = z(nb, 3), z(nb, 4), z(nb)
        self.root_rot[..., 3] = 1.0
        self.joint_rot[..., 3] = 1.0

    def get_mlib_format(self, char_model):
        """-> (frames [..., 6 + dofs] = root position, root exponential map, joint dofs;  contacts)"""
        frames = torch.cat([self.root_pos, torch_util.quat_to_exp_map(self.root_rot), char_model.rot_to_dof(self.joint_rot)], dim=-1)
        return frames, self.contacts

    def get_slice(self, in_slice):
        return self._map(lambda x: x[:, in_slice])

    def unsqueeze(self, dim):
        return self._map(lambda x: x.unsqueeze(dim))

    def squeeze(self, dim):
        return self._map(lambda x: x.squeeze(dim))

    def expand_first_dim(self, b):
        return self._map(lambda x: x.expand(b, *x.shape[1:]))

    def get_idx(self, idx):
        return self._map(lambda x: x[idx])

    def set_vals(self, other, ids):
        for f in FIELDS:
            mine = getattr(self, f)
            if mine is not None:
                mine[ids] = getattr(other, f)[ids].clone()

    def store(self, other):
        """in place: every field takes `other`'s values (same shapes); the tensors keep their addresses (captured rollout steps)"""
        for f in FIELDS:
            mine = getattr(self, f)
            if mine is not None:
                mine.copy_(getattr(other, f))

    def set_vals_masked(self, other, mask):
        """set_vals for the rows where the bool mask [batch] is set, as fixed-shape work (no index list)"""
        for f in FIELDS:
            mine = getattr(self, f)
            if mine is not None:
                m = mask.reshape([-1] + [1] * (mine.dim() - 1))
                mine.copy_(torch.where(m, getattr(other, f), mine))

    def get_copy(self, new_device):
        return self._map(lambda x: x.clone().to(device=new_device))

    def set_device(self, device):
        for f in FIELDS:
            if getattr(self, f) is not None:
                setattr(self, f, getattr(self, f).to(device=device))


def cat_motion_frames(motion_frames_list):
    """concatenate along the frame axis (dim 1); the first element decides which fields exist"""
    first = motion_frames_list[0]
    assert first.root_pos.dim() == 3
    return MotionFrames(**{f: (None if getattr(first, f) is None else torch.cat([getattr(m, f) for m in motion_frames_list], dim=1)) for f in FIELDS})


def motion_frames_from_mlib_format(mlib_motion_frames, char_model, contacts=None):
    root_pos = mlib_motion_frames[..., 0:3]
    root_rot = torch_util.exp_map_to_quat(mlib_motion_frames[..., 3:6])
    joint_rot = char_model.dof_to_rot(mlib_motion_frames[..., 6:])
    body_pos, body_rot = char_model.forward_kinematics(root_pos, root_rot, joint_rot)
    return MotionFrames(root_pos=root_pos, root_rot=root_rot, joint_rot=joint_rot, body_pos=body_pos, body_rot=body_rot, contacts=contacts)
