"""llava.model.llava_arch mixin methods of the 3-D path on HIP (reference: llava/model/llava_arch.py).

`LlavaMetaForCausalLM` here carries the methods the video + `avg-discrete-sin3d` + bilinear-pool + `grid`
newline branch of prepare_inputs_labels_for_multimodal uses; each keeps the reference's name, arguments and
return convention and launches one kernel of libv3d_hip.so.  Branches the eval scripts never reach
(anyres / unpad / faster-video / mrope / llava3d / sample9 / minmax / mlp PE) are not provided.
"""
import math

import torch

from v3d import ops


class LlavaMetaForCausalLM:
    """Host needs: self.config (mm_spatial_pool_mode, voxel_size, min_xyz_range, max_xyz_range),
    self.get_model().image_newline, self.get_vision_tower().num_patches_per_side."""

    def get_2dPool(self, image_feature, stride=2):
        """[F, 729, C] -> [F, 196, C]  (llava_arch.py:191-210, bilinear mode)."""
        if getattr(self.config, "mm_spatial_pool_mode", "bilinear") != "bilinear":
            raise NotImplementedError("only mm_spatial_pool_mode == 'bilinear' is on the accelerated path")
        side = self.get_vision_tower().num_patches_per_side
        n = math.ceil(side / stride)
        F_, _, C = image_feature.shape
        return ops.visual_tokens(image_feature, side=side, n=n, pool=True).view(F_, n * n, C)

    def average_coordinate_in_patch(self, world_coords, patch_size=27):
        """[V,384,384,3] -> [V,14,14,3]  (llava_arch.py:213-223)."""
        avg, _, _ = ops.coord_pool_voxel(world_coords, patch_size, self.config.min_xyz_range, self.config.max_xyz_range,
                                         self.config.voxel_size, want_vox=False, want_ids=False)
        return avg

    def discrete_coords(self, world_coords, xyz_min=None):
        """clamp / shift / divide / round, integer-valued floats of the input dtype (llava_arch.py:259-272)."""
        return ops.discrete_coords(world_coords, self.config.min_xyz_range, self.config.max_xyz_range, self.config.voxel_size)

    def add_token_per_grid(self, image_feature):
        """[F, h*h, C] -> [F*h*(h+1), C] with image_newline after each row (llava_arch.py:307-328)."""
        h = int(math.isqrt(image_feature.shape[1]))
        return ops.visual_tokens(image_feature, newline=self.get_model().image_newline, n=h, pool=False)

    def fused_visual_tokens(self, image_feature, world_coords, table, out=None):
        """The whole 3-D branch for one video sample in two launches (llava_arch.py:395-420, 469, 506-517, 536):
        coords [F,384,384,3] -> voxel ids;  feat [F,729,C] -> pooled + PE(ids) + newline rows [F*210, C]."""
        c = self.config
        _, _, ids = ops.coord_pool_voxel(world_coords, 27, c.min_xyz_range, c.max_xyz_range, c.voxel_size,
                                         want_avg=False, want_vox=False)
        side = self.get_vision_tower().num_patches_per_side
        return ops.visual_tokens(image_feature, ids, table, self.get_model().image_newline, side=side,
                                 n=math.ceil(side / 2), pool=True, out=out)
