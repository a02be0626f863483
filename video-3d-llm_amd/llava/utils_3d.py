"""llava.utils_3d (reference: llava/utils_3d.py:3-13)."""
import numpy as np


def convert_pc_to_box(obj_pc):
    """Axis-aligned box of a point set [N, >=3] -> (center[3], size[3])."""
    xyz = np.asarray(obj_pc)[:, :3]
    lo, hi = xyz.min(axis=0), xyz.max(axis=0)
    return [(lo[i] + hi[i]) / 2 for i in range(3)], [hi[i] - lo[i] for i in range(3)]
