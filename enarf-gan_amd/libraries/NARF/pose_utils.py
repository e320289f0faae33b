"""transform_pose with the reference's signature (libraries/NARF/pose_utils.py:129-148).

Kept for callers that build part frames themselves; TriPlaneNARF.forward does the same arithmetic inside
enarf_prepare (one launch) instead."""
import torch


def transform_pose(pose_to_camera, bone_length, origin_location, parent_id):
    par = torch.as_tensor(list(parent_id)[1:], dtype=torch.long, device=pose_to_camera.device)
    mid = (pose_to_camera[:, 1:, :, 3:] + pose_to_camera[:, par, :, 3:]) / 2
    if origin_location == "center":
        pose_to_camera = torch.cat([pose_to_camera[:, 1:, :, :3], mid], dim=-1)
    elif origin_location == "center_fixed":
        pose_to_camera = torch.cat([pose_to_camera[:, par, :, :3], mid], dim=-1)
    elif origin_location == "center+head":
        bone_length = torch.cat([bone_length, torch.ones(bone_length.shape[0], 1, 1, device=bone_length.device)], dim=1)
        _pose = torch.cat([pose_to_camera[:, par, :, :3], mid], dim=-1)
        pose_to_camera = torch.cat([_pose, pose_to_camera[:, 15][:, None]], dim=1)
    return pose_to_camera, bone_length
