"""llava.model.language_model.llava_qwen on the HIP engine (reference: llava/model/language_model/llava_qwen.py).

`LlavaQwenConfig` / `LlavaQwenForCausalLM` keep the reference's names, are registered with the HF Auto classes
(llava_qwen.py:347-348) and are what `llava.model.builder.load_pretrained_model` (builder.py:206-228) instantiates:

    LlavaQwenForCausalLM.from_pretrained(model_path, low_cpu_mem_usage=True, attn_implementation=..., config=llava_cfg,
                                         device_map="auto", torch_dtype=torch.float16)

reads the checkpoint's safetensors shards (state-dict keys of the reference: model.vision_tower.*, model.mm_projector.{0,2}.*,
model.image_newline, model.embed_tokens, model.layers.N.*, model.norm, lm_head, ground_head_*) straight into
v3d.engine.Engine's HBM layouts.  Call contracts mirrored (3-D eval drivers):
    model.generate(input_ids, images=[1,F,3,384,384], modalities="video", video_dict={...}, do_sample=False, num_beams=1,
                   max_new_tokens=N, use_cache=True)                                     -> LongTensor [1, n_new]
    model(input_ids, images=..., modalities="video", video_dict=..., labels=labels, use_object_proposals=True)
                                                                                         -> (None, scores [n_obj + 1])
    model(input_ids, images=..., modalities="video", video_dict=..., [labels=...])       -> CausalLMOutputWithPast(loss, logits [1,S,V] f32)
Greedy decoding only, no autograd (the loss is a forward value); sampling / beams / box_labels raise NotImplementedError, unknown keyword arguments raise
TypeError - nothing is silently dropped.
"""
import json
import os
import warnings

import torch
import torch.nn as nn
from transformers import AutoConfig, AutoModelForCausalLM, Qwen2Config
from transformers.modeling_outputs import CausalLMOutputWithPast

from llava.model.llava_arch import LlavaMetaForCausalLM, LlavaMetaModel
from llava.model.multimodal_encoder.siglip_encoder import SigLipVisionTower
from v3d import loader
from v3d._native import V3DError
from v3d.engine import Engine
from v3d.token_ids import IMAGE_TOKEN_INDEX


class LlavaQwenConfig(Qwen2Config):
    model_type = "llava_qwen"


class LlavaQwenModel(LlavaMetaModel):
    config_class = LlavaQwenConfig


def _device_of(device_map):
    if not torch.cuda.is_available():
        raise V3DError("LlavaQwenForCausalLM needs an MI355X: the accelerated path has no CPU fallback")
    if device_map in (None, "auto", "cuda"):
        return torch.device("cuda", torch.cuda.current_device())
    if isinstance(device_map, dict):
        vals = set(device_map.values())
        if len(vals) != 1:
            raise NotImplementedError("device_map must place the whole model on one GPU (scenes shard data-parallel, one process per GPU)")
        device_map = vals.pop()
    return torch.device("cuda", device_map) if isinstance(device_map, int) else torch.device(device_map)


class LlavaQwenForCausalLM(nn.Module, LlavaMetaForCausalLM):
    config_class = LlavaQwenConfig
    _GENERATE_KW = {"max_new_tokens", "do_sample", "num_beams", "temperature", "top_p", "top_k", "use_cache", "attention_mask",
                    "position_ids", "stopping_criteria", "eos_token_id", "pad_token_id", "video_dict", "max_length"}

    def __init__(self, config, state_dict=None, dtype=torch.float16, device=None, llm_fp8=False, max_frames=32):
        super().__init__()
        if state_dict is None:
            raise V3DError("LlavaQwenForCausalLM(config) without weights: construct it with from_pretrained(checkpoint_dir) "
                           "(training from a fresh init is outside the accelerated path)")
        config.model_type = "llava_qwen"
        config.rope_scaling = None
        device = _device_of(device)
        self.config = config
        self.engine = Engine(loader.engine_config(config, state_dict), state_dict, dtype=dtype, device=device, max_frames=max_frames,
                             llm_fp8=llm_fp8)
        self._device, self._dtype = device, dtype
        self.ground_head_type = getattr(config, "ground_head_type", None)
        if self.ground_head_type not in (None, "infonce", "mlp", "score"):                      # llava_qwen.py:57-104
            raise NotImplementedError(f"ground_head_type {self.ground_head_type!r}: the reference defines 'infonce', 'mlp' and 'score'")
        tower = SigLipVisionTower(getattr(config, "mm_vision_tower", "siglip"), vision_tower_cfg=config).bind(self.engine)
        self.model = LlavaQwenModel(config, self.engine, tower)
        self.generation_eos = None

    # ------------------------------------------------------------------ loading (builder.py:206-228)
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, *model_args, config=None, torch_dtype=None, dtype=None, device_map=None,
                        low_cpu_mem_usage=None, attn_implementation=None, llm_fp8=False, max_frames=32, **kwargs):
        """low_cpu_mem_usage / attn_implementation are accepted and have no effect here (weights go straight to HBM; attention
        is the engine's flash kernel whatever the name says).  Quantised / sharded-device loading is refused."""
        for k in ("load_in_8bit", "load_in_4bit", "quantization_config"):
            if kwargs.pop(k, None):
                raise NotImplementedError(f"{k}: bitsandbytes quantisation is outside the accelerated path (use llm_fp8=True for e4m3 weights)")
        if model_args or kwargs:
            raise TypeError(f"LlavaQwenForCausalLM.from_pretrained: unsupported arguments {list(model_args) + sorted(kwargs)}")
        path = os.fspath(pretrained_model_name_or_path)
        if config is None:
            config = LlavaQwenConfig.from_pretrained(path)
        dt = dtype if dtype is not None else torch_dtype
        if dt is None or dt == "auto":
            dt = getattr(config, "dtype", None) or getattr(config, "torch_dtype", None) or torch.float16
        if isinstance(dt, str):
            dt = getattr(torch, dt.replace("torch.", ""))
        if dt not in (torch.float16, torch.bfloat16):
            raise NotImplementedError(f"dtype {dt}: the engine runs f16 (the eval default, builder.py:27) or bf16")
        sd = loader.read_weights(path)
        if not any(k.startswith(loader.VIT_PREFIX) for k in sd):         # tower not saved with the model: SigLIP directory
            tower_dir = getattr(config, "mm_vision_tower", None)
            if tower_dir == "google/siglip-so400m-patch14-384":
                tower_dir = os.getenv("VISION_TOWER", tower_dir)        # multimodal_encoder/builder.py:20-21
            if not tower_dir or not os.path.isdir(tower_dir):
                raise V3DError(f"the checkpoint holds no vision tower and config.mm_vision_tower = {tower_dir!r} is not a local directory")
            sd.update(loader.drop_deleted_tower_layer(loader.read_tower_weights(tower_dir), True))
        if "lm_head.weight" not in sd:                 # tie_word_embeddings checkpoints store the table once
            if not getattr(config, "tie_word_embeddings", False):
                raise V3DError("the checkpoint holds no lm_head.weight and config.tie_word_embeddings is not set")
            sd["lm_head.weight"] = sd["model.embed_tokens.weight"]
        vocab = getattr(config, "vocab_size", None)
        if vocab is not None and vocab != sd["lm_head.weight"].shape[0]:
            # overwrite_config={"vocab_size": ...} (model_scanqa.py:98) on a checkpoint whose tables are larger / smaller
            if vocab > sd["lm_head.weight"].shape[0]:
                raise V3DError(f"config.vocab_size {vocab} exceeds the checkpoint's {sd['lm_head.weight'].shape[0]} rows")
            sd["lm_head.weight"] = sd["lm_head.weight"][:vocab]
            sd["model.embed_tokens.weight"] = sd["model.embed_tokens.weight"][:vocab]
        model = cls(config, sd, dtype=dt, device=device_map, llm_fp8=llm_fp8, max_frames=max_frames)
        gen = os.path.join(path, "generation_config.json")
        if os.path.exists(gen):
            with open(gen) as f:
                model.generation_eos = json.load(f).get("eos_token_id")
        return model.eval()

    def resize_token_embeddings(self, new_num_tokens=None, **kw):
        """builder.py:287 always calls this with len(tokenizer).  Same size: nothing to do; fewer: the tables are cut (HF keeps the
        first rows); more (builder.py:282-286 adds <im_patch> / <im_start> / <im_end> when the config asks for them): the embedding and
        LM-head tables grow by rows set to the MEAN of the existing rows - transformers' own initialisation of resized embeddings
        (mean_resizing) without its covariance noise, so the load is deterministic; an inference prompt never contains the new ids and a
        mean row's logit cannot win an argmax."""
        eng = self.engine
        l = eng.cfg.llm
        if new_num_tokens is None or new_num_tokens == l.vocab:
            return self
        if new_num_tokens > l.vocab:
            eng.grow_vocab(int(new_num_tokens))
        l.vocab = int(new_num_tokens)
        self.config.vocab_size = int(new_num_tokens)
        return self

    # ------------------------------------------------------------------ module surface
    def get_model(self):
        return self.model

    @property
    def device(self):
        return self._device

    @property
    def dtype(self):
        return self._dtype

    def _eos(self, override=None):
        e = override if override is not None else (self.generation_eos if self.generation_eos is not None
                                                   else getattr(self.config, "eos_token_id", None))
        return e

    def _sample(self, input_ids, images, video_dict):
        if input_ids is None or input_ids.shape[0] != 1:
            raise NotImplementedError("batch size 1, as every 3-D eval driver uses")
        if images is None or video_dict is None:
            raise ValueError("images and video_dict are required on the accelerated (video) path")
        frames = images[0] if isinstance(images, (list, tuple)) else (images[0] if images.dim() == 5 else images)
        return input_ids[0].cpu(), frames.to(self._device), video_dict["world_coords"][0].to(self._device)

    # ------------------------------------------------------------------ forward (llava_qwen.py:121-205)
    @torch.no_grad()
    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None, labels=None,
                use_cache=None, output_attentions=None, output_hidden_states=None, images=None, image_sizes=None, return_dict=None,
                modalities=("image",), dpo_forward=False, cache_position=None, video_dict=None, use_object_proposals=False,
                box_labels=None):
        """Inference forward.  use_object_proposals=True: (None, scores) of predict_box with box_labels=None
        (llava_qwen.py:239-331, model_scanrefer.py:165-173).  Otherwise CausalLMOutputWithPast with the f32 logits of every
        position (modeling_qwen2.py:1188-1192)."""
        if output_attentions or output_hidden_states or dpo_forward:
            raise NotImplementedError("output_attentions / output_hidden_states / dpo_forward are not on the accelerated path")
        if box_labels is not None:
            raise NotImplementedError("the grounding loss (box_labels) belongs to training, outside the accelerated inference path")
        if past_key_values is not None:
            raise NotImplementedError("external KV caches are not supported; use generate() (the engine owns the cache)")
        eng = self.engine
        if inputs_embeds is None:
            (_, position_ids, attention_mask, _, inputs_embeds, new_labels, object_features, object_boxes) = \
                self.prepare_inputs_labels_for_multimodal(input_ids, position_ids, attention_mask, None, labels, images,
                                                          modalities, image_sizes, video_dict, use_object_proposals=use_object_proposals)
            if inputs_embeds is None:
                raise NotImplementedError("text-only / decode re-entry forward is served by generate()")
        elif use_object_proposals:
            raise ValueError("use_object_proposals needs input_ids + images (the object features come from the ViT output)")
        if inputs_embeds.dim() != 3 or inputs_embeds.shape[0] != 1:
            raise NotImplementedError("batch size 1")
        S = inputs_embeds.shape[1]
        if S > eng.cfg.llm.max_pos:
            raise V3DError(f"sequence {S} exceeds engine capacity {eng.cfg.llm.max_pos}")
        x = eng.l_x[:S]
        x.copy_(inputs_embeds[0].to(device=eng.device, dtype=eng.dtype))
        if use_object_proposals:
            gt = getattr(self.config, "ground_token_ids", None)
            if gt is None:
                raise ValueError("config.ground_token_ids is needed to locate the <ground> label")
            loc = ((new_labels[0] >= gt[0]) & (new_labels[0] <= gt[-1])).nonzero().flatten()
            if loc.numel() != 1:
                raise ValueError("exactly one <ground> label token expected")
            eng.llm_forward(x, 0, head=False, last_rows=[int(loc[0])])      # predict_box reads the <ground> row only (llava_qwen.py:280-281)
            return None, eng.predict_box(x, int(loc[0]), object_features)
        eng.llm_forward(x, 0, head=False)
        logits = eng.logits_all_rows(x)
        loss = None
        if labels is not None:        # the shifted cross-entropy of modeling_qwen2.py:1195-1205 over the re-aligned labels (forward value only)
            from v3d import ops
            loss, _ = ops.cross_entropy(logits, new_labels[0].to(eng.device))
        return CausalLMOutputWithPast(loss=loss, logits=logits[None], past_key_values=None)

    # ------------------------------------------------------------------ generate (llava_qwen.py:208-226)
    @torch.no_grad()
    def generate(self, inputs=None, images=None, image_sizes=None, modalities=("image",), **kwargs):
        unknown = set(kwargs) - self._GENERATE_KW
        if unknown:
            raise TypeError(f"LlavaQwenForCausalLM.generate: unsupported arguments {sorted(unknown)} (accepted: {sorted(self._GENERATE_KW)})")
        if kwargs.get("do_sample") or (kwargs.get("num_beams") or 1) != 1:
            raise NotImplementedError("greedy decoding only (the 3-D eval drivers run temperature 0, num_beams 1)")
        if kwargs.get("use_cache") is False:
            raise NotImplementedError("use_cache=False: the engine always decodes through its KV cache")
        am = kwargs.get("attention_mask")
        if am is not None and not bool(am.bool().all()):
            raise NotImplementedError("padded prompts (attention_mask with zeros) are not on the accelerated path")
        if kwargs.get("position_ids") is not None:
            raise NotImplementedError("explicit position_ids are not on the accelerated path (positions are 0..S-1)")
        video_dict = kwargs.get("video_dict")
        ids, frames, coords = self._sample(inputs, images, video_dict)
        n_vis = frames.shape[0] * self.engine.cfg.pool_out * (self.engine.cfg.pool_out + 1)
        S = len(ids) - 1 + n_vis
        max_new = kwargs.get("max_new_tokens")
        if max_new is None:
            max_new = (kwargs["max_length"] - 0) if kwargs.get("max_length") else 20          # HF: max_length counts new tokens when only inputs_embeds are given
        room = self.engine.cfg.llm.max_pos - S + 1
        if int(max_new) > room:
            warnings.warn(f"max_new_tokens {max_new} cut to {room}: prompt of {S} rows in an engine of {self.engine.cfg.llm.max_pos} "
                          "positions (config.v3d_max_positions raises the capacity)")
        max_new = min(int(max_new), room)
        crit = kwargs.get("stopping_criteria")
        stopping = None
        if crit:
            stopping = lambda toks: any(bool(torch.as_tensor(c(toks[None], None)).any()) for c in crit)      # noqa: E731
        box_input = video_dict.get("box_input") if hasattr(video_dict, "get") else None
        coord_ids = getattr(self.config, "coord_token_ids", None)
        out = self.engine.generate(ids, frames, coords, max_new_tokens=max_new, eos_token_id=self._eos(kwargs.get("eos_token_id")),
                                   stopping=stopping, box_input=box_input if box_input is not None and len(box_input) else None,
                                   coord_token_id=coord_ids[0] if coord_ids else None)
        return out.view(1, -1)

    def prepare_inputs_for_generation(self, *a, **kw):
        raise NotImplementedError("HF's generation loop is replaced by Engine.generate (greedy, device-side token loop)")


AutoConfig.register("llava_qwen", LlavaQwenConfig, exist_ok=True)
AutoModelForCausalLM.register(LlavaQwenConfig, LlavaQwenForCausalLM, exist_ok=True)
