"""Writes the on-disk form of a scene the way the dataset holds it (16-bit depth PNGs, 4 x 4 pose text files, RGB JPEGs/PNGs)
from the arrays stored in tests/golden/world_coords.npz or from synthetic arrays - shared by the a4 tests and the eval-harness
test.  Test infrastructure."""
import os

import numpy as np
from PIL import Image


def write_frames(folder, depth_u16, poses, rgb=None, step=10, ext=".jpg"):
    """depth [V,H,W] uint16, poses [V,4,4] f64 -> files <folder>/<frame>.png / .txt (/ .jpg); returns the .jpg paths
    (video_utils.py:214,221 derive the depth / pose paths from the image path)."""
    os.makedirs(folder, exist_ok=True)
    files = []
    for v in range(depth_u16.shape[0]):
        base = os.path.join(folder, f"{v * step:05d}")
        Image.fromarray(np.ascontiguousarray(depth_u16[v])).save(base + ".png")
        np.savetxt(base + ".txt", poses[v])
        if rgb is not None:
            Image.fromarray(rgb[v]).save(base + ext, quality=95)
        files.append(base + ext)
    return files
