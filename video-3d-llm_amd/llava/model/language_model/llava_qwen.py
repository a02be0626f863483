"""llava.model.language_model.llava_qwen.LlavaQwenForCausalLM on the HIP engine.

Mirrors the call contract the 3-D eval drivers use (llava_qwen.py:208-236, model_scanqa.py:173-185):
    model.generate(input_ids, images=[1,F,3,384,384], modalities="video", video_dict={"world_coords": [1,F,384,384,3], ...},
                   do_sample=False, num_beams=1, max_new_tokens=N, use_cache=True)  -> LongTensor [1, n_new]
and the grounding call of ScanRefer / Multi3DRefer (model_scanrefer.py:165-173):
    _, scores = model(input_ids, images=..., modalities="video", video_dict=..., labels=labels,
                      use_object_proposals=True, box_labels=None)            -> scores [n_obj + 1]
Weights come in under the reference's state-dict keys (Engine docstring; ground_head_{obj,query}.*,
ground_head_zero_target for the infonce head).  Sampling and beams are not on the accelerated path.
"""
import types

import torch
import torch.nn as nn

from v3d.engine import Engine, EngineConfig


class LlavaQwenForCausalLM(nn.Module):
    def __init__(self, engine_config: EngineConfig, state_dict, dtype=torch.float16, device="cuda", eos_token_id=None):
        super().__init__()
        self.engine = Engine(engine_config, state_dict, dtype=dtype, device=device)
        l = engine_config.llm
        self.config = types.SimpleNamespace(
            hidden_size=l.hidden, vocab_size=l.vocab, mm_use_im_start_end=False, mm_spatial_pool_mode="bilinear",
            mm_spatial_pool_stride=2, mm_newline_position="grid", world_position_embedding_type="avg-discrete-sin3d",
            voxel_size=engine_config.voxel_size, min_xyz_range=list(engine_config.min_xyz),
            max_xyz_range=list(engine_config.max_xyz), eos_token_id=eos_token_id)
        self._device = torch.device(device)

    @property
    def device(self):
        return self._device

    @torch.no_grad()
    def forward(self, input_ids=None, images=None, modalities=("image",), video_dict=None, labels=None,
                use_object_proposals=False, box_labels=None, **kw):
        """Only the inference grounding call is provided: returns (None, scores) like predict_box with
        box_labels=None (llava_qwen.py:176-205, 239-331)."""
        if not use_object_proposals or box_labels is not None:
            raise NotImplementedError("training / plain-LM forward is outside the accelerated path; use generate()")
        gt = getattr(self.config, "ground_token_ids", None)
        if gt is None:
            raise ValueError("config.ground_token_ids is needed to locate the <ground> label")
        loc = ((labels[0] >= gt[0]) & (labels[0] <= gt[-1])).nonzero().flatten()
        if loc.numel() != 1:
            raise ValueError("exactly one <ground> label token expected")
        frames = images[0] if images.dim() == 5 else images
        scores = self.engine.ground_scores(input_ids[0].cpu(), int(loc[0]), frames.to(self._device),
                                           video_dict["world_coords"][0].to(self._device), video_dict["objects"][0])
        return None, scores

    @torch.no_grad()
    def generate(self, inputs=None, images=None, image_sizes=None, modalities=("image",), video_dict=None,
                 max_new_tokens=512, do_sample=False, num_beams=1, temperature=0.0, top_p=None, use_cache=True, **kw):
        if do_sample or num_beams != 1:
            raise NotImplementedError("greedy decoding only (model_scanqa.py runs temperature 0, num_beams 1)")
        if inputs.shape[0] != 1:
            raise NotImplementedError("batch size 1, as every 3-D eval driver uses")
        frames = images[0] if images.dim() == 5 else images
        coords = video_dict["world_coords"][0]
        out = self.engine.generate(inputs[0].cpu(), frames.to(self._device), coords.to(self._device),
                                   max_new_tokens=max_new_tokens, eos_token_id=self.config.eos_token_id)
        return out.view(1, -1)
