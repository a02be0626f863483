"""llava.model.llava_arch on the HIP engine (reference: llava/model/llava_arch.py).

`LlavaMetaModel` / `LlavaMetaForCausalLM` keep the reference's names, method signatures and return conventions for the
video + `avg-discrete-sin3d` + bilinear-pool + `grid`-newline branch - the one every 3-D eval driver and the shipped
training script select (scripts/3d/train/train_multi.sh:78-87) - and launch kernels of libv3d_hip.so through
v3d.engine.Engine.  Branches those scripts never reach (anyres / unpad image grids, faster-video, mrope, llava3d,
sample9 / minmax / mlp PE, multi-image prompts, batch > 1) raise NotImplementedError instead of computing something else.
"""
import math

import torch

from v3d import ops
from v3d.token_ids import IGNORE_INDEX, IMAGE_TOKEN_INDEX


class _Projector:
    """model.mm_projector handle: mlp2x_gelu on the engine's GEMMs (multimodal_projector/builder.py:41-48)."""

    def __init__(self, engine):
        self._e = engine

    def __call__(self, x):
        e = self._e
        flat = x.reshape(-1, x.shape[-1]).to(device=e.device, dtype=e.dtype)
        if flat.shape[1] != e.v_Hk:                                   # the engine's K-padded layout
            pad = torch.zeros((flat.shape[0], e.v_Hk), dtype=e.dtype, device=e.device)
            pad[:, : flat.shape[1]] = flat
            flat = pad
        h = ops.gemm(flat.contiguous(), e.p_w0, bias=e.p_b0, epilogue=ops.EPI_BIAS_GELU_ERF)
        return ops.gemm(h, e.p_w2, bias=e.p_b2, epilogue=ops.EPI_BIAS).view(*x.shape[:-1], -1)


class LlavaMetaModel:
    """What `model.get_model()` returns: the handles the reference's LlavaMetaModel owns (llava_arch.py:34-70) -
    vision tower, projector, image_newline, embed_tokens, world_position_embedding - all views of one Engine."""

    def __init__(self, config, engine, vision_tower):
        from .position_encoding import PositionEmbeddingSine3D
        self.config = config
        self._engine = engine
        self.vision_tower = vision_tower
        self.mm_projector = _Projector(engine)
        self.image_newline = engine.newline
        self.world_position_embedding = PositionEmbeddingSine3D(config.hidden_size)

    def get_vision_tower(self):
        return self.vision_tower

    def embed_tokens(self, input_ids):
        flat = input_ids.reshape(-1)
        return ops.embed_gather(self._engine.embed, flat.to(self._engine.device)).view(*input_ids.shape, -1)


class LlavaMetaForCausalLM:
    """Mixin for the model class.  Host needs: self.config, self.engine (v3d.engine.Engine), self.get_model()."""

    def get_vision_tower(self):
        return self.get_model().get_vision_tower()

    # ---- llava_arch.py:191-210
    def get_2dPool(self, image_feature, stride=2):
        """[F, 729, C] -> [F, 196, C]  (bilinear mode)."""
        if getattr(self.config, "mm_spatial_pool_mode", "bilinear") != "bilinear":
            raise NotImplementedError("only mm_spatial_pool_mode == 'bilinear' is on the accelerated path")
        side = self.get_vision_tower().num_patches_per_side
        n = math.ceil(side / stride)
        F_, _, C = image_feature.shape
        return ops.visual_tokens(image_feature, side=side, n=n, pool=True).view(F_, n * n, C)

    # ---- llava_arch.py:213-223
    def average_coordinate_in_patch(self, world_coords, patch_size=27):
        """[V,384,384,3] -> [V,14,14,3]."""
        avg, _, _ = ops.coord_pool_voxel(world_coords, patch_size, self.config.min_xyz_range, self.config.max_xyz_range,
                                         self.config.voxel_size, want_vox=False, want_ids=False)
        return avg

    # ---- llava_arch.py:259-272
    def discrete_coords(self, world_coords, xyz_min=None):
        """clamp / shift / divide / round, integer-valued floats of the input dtype."""
        return ops.discrete_coords(world_coords, self.config.min_xyz_range, self.config.max_xyz_range, self.config.voxel_size)

    # ---- llava_arch.py:275-280
    def encode_images(self, images):
        """[F,3,384,384] -> [F,729,hidden]: SigLIP tower (hidden_states[-1] of the truncated encoder) + mm_projector."""
        return self.engine.encode_images(images.to(self.engine.device))

    # ---- llava_arch.py:307-328
    def add_token_per_grid(self, image_feature):
        """[F, h*h, C] -> [F*h*(h+1), C] with image_newline after each row."""
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

    # ---- llava_arch.py:336-836
    def prepare_inputs_labels_for_multimodal(self, input_ids, position_ids, attention_mask, past_key_values, labels, images,
                                             modalities=["image"], image_sizes=None, video_dict=None, use_object_proposals=False):
        """Same arguments and 8-tuple as the reference:
            (None, position_ids, attention_mask, past_key_values, inputs_embeds [1,S,H], labels [1,S] | None,
             object_features [n,H] | None, object_boxes [n,6] | None)
        for ONE video sample with one <image> placeholder.  Decode re-entry (input_ids [1,1], llava_arch.py:434-435) returns the
        arguments unchanged with inputs_embeds None, as the reference does."""
        eng = self.engine
        if self.get_vision_tower() is None or images is None or input_ids.shape[1] == 1:
            return input_ids, position_ids, attention_mask, past_key_values, None, labels, None, None
        if isinstance(modalities, str):
            modalities = [modalities]
        if input_ids.shape[0] != 1 or list(modalities) != ["video"]:
            raise NotImplementedError("the accelerated path takes one video sample per call (batch 1, modalities=['video']), "
                                      "as every 3-D eval driver passes")
        if video_dict is None or "world_coords" not in video_dict:
            raise ValueError("video_dict['world_coords'] is required (world_position_embedding_type 'avg-discrete-sin3d')")
        if past_key_values is not None:
            raise NotImplementedError("an external past_key_values cache is not supported: the engine owns the KV cache")
        if attention_mask is not None and not bool(attention_mask.bool().all()):
            raise NotImplementedError("padded prompts (attention_mask with zeros) are not on the accelerated path")
        frames = images[0] if isinstance(images, (list, tuple)) else (images[0] if images.dim() == 5 else images)
        coords = video_dict["world_coords"][0].to(device=eng.device, dtype=eng.dtype)
        ids = input_ids[0].cpu()
        feats = eng.encode_images(frames.to(eng.device))
        vox = eng.voxel_ids(coords)
        object_features = object_boxes = None
        if use_object_proposals:
            oft = getattr(self.config, "object_feature_type", "patch14-pe")
            if "patch14" not in oft and "patch27" not in oft:                                   # llava_arch.py:367-376
                raise NotImplementedError(f"object_feature_type {oft!r}: the reference defines 'patch14*' and 'patch27*'")
            object_boxes = video_dict["objects"][0]
            object_features = eng.object_features(feats, coords, object_boxes.to(device=eng.device, dtype=eng.dtype).contiguous())
        box_input = video_dict.get("box_input")
        coord_ids = getattr(self.config, "coord_token_ids", None)
        x = eng.build_inputs_embeds(ids, feats, vox, image_token=IMAGE_TOKEN_INDEX,
                                    box_input=box_input if box_input is not None and len(box_input) else None,
                                    coord_token_id=coord_ids[0] if coord_ids else None)
        limit = getattr(self.config, "tokenizer_model_max_length", None)
        S = x.shape[0] if limit is None else min(x.shape[0], int(limit))                    # llava_arch.py:766-770
        at = ids.tolist().index(IMAGE_TOKEN_INDEX)
        n_vis = x.shape[0] - (len(ids) - 1)
        new_labels = None
        if labels is not None:
            lab = labels[0].cpu()
            new_labels = torch.cat([lab[:at], torch.full((n_vis,), IGNORE_INDEX, dtype=lab.dtype), lab[at + 1:]])[:S][None].to(labels.device)
        if attention_mask is not None:
            attention_mask = torch.ones((1, S), dtype=attention_mask.dtype, device=attention_mask.device)
        if position_ids is not None:
            position_ids = torch.arange(S, dtype=position_ids.dtype, device=position_ids.device)[None]
        # the caller owns the returned tensor (the engine's own buffer is reused by the next call)
        return None, position_ids, attention_mask, past_key_values, x[:S].clone()[None], new_labels, object_features, object_boxes
