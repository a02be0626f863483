"""Checkpoint side of `LlavaQwenForCausalLM.from_pretrained` (reference: llava/model/builder.py:206-228 ->
transformers' from_pretrained): reads the weight shards of a LLaVA-Qwen checkpoint directory into one state dict
under the reference's keys and derives the engine's widths from the HF config + the tensors' shapes.

Only loaders that execute nothing from the file are used: safetensors, or torch.load(weights_only=True) for
pytorch_model*.bin.  Host-side plumbing; nothing here computes on the path."""
import glob
import json
import os

import torch

from ._native import V3DError
from .engine import EngineConfig, LlmConfig, VitConfig

VIT_PREFIX = "model.vision_tower.vision_tower.vision_model."


def _load_file(path):
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path, device="cpu")
    return torch.load(path, map_location="cpu", weights_only=True)


def read_weights(model_dir):
    """All tensors of a checkpoint directory (model.safetensors[.index.json] shards or pytorch_model*.bin)."""
    if not os.path.isdir(model_dir):
        raise V3DError(f"{model_dir} is not a checkpoint directory (hub ids cannot be fetched: there is no network path here)")
    for index in ("model.safetensors.index.json", "pytorch_model.bin.index.json"):
        ip = os.path.join(model_dir, index)
        if os.path.exists(ip):
            with open(ip) as f:
                files = sorted(set(json.load(f)["weight_map"].values()))
            break
    else:
        files = sorted(os.path.basename(p) for p in glob.glob(os.path.join(model_dir, "*.safetensors")))
        if not files:
            files = sorted(os.path.basename(p) for p in glob.glob(os.path.join(model_dir, "pytorch_model*.bin")))
    if not files:
        raise V3DError(f"no *.safetensors / pytorch_model*.bin weights under {model_dir}")
    sd = {}
    for f in files:
        sd.update(_load_file(os.path.join(model_dir, f)))
    return sd


def read_tower_weights(tower_dir):
    """SigLipVisionModel.from_pretrained(vision_tower_name) (siglip_encoder.py:568) for checkpoints that do not carry
    the tower: keys `vision_model.*` of the SigLIP directory, re-prefixed to where the LLaVA state dict keeps them."""
    sd = read_weights(tower_dir)
    out = {}
    for k, v in sd.items():
        if k.startswith("vision_model."):
            out["model.vision_tower.vision_tower." + k] = v
    if not out:
        raise V3DError(f"{tower_dir} holds no vision_model.* tensors")
    return out


def engine_config(hf_config, sd, max_positions=None):
    """EngineConfig from the HF config (Qwen2 fields, llava_qwen.py:37-38 + the 3-D keys train_3d.py:1429-1459 persists)
    and the SigLIP tensors' shapes.  The tower keeps every layer the checkpoint has except the one the reference deletes
    (siglip_encoder.py:570): a checkpoint saved AFTER that deletion (the tower was trained / saved with the model) has it
    gone already, a raw SigLIP directory still has all 27."""
    c = hf_config
    wtype = getattr(c, "world_position_embedding_type", "avg-discrete-sin3d")
    if not all(t in wtype for t in ("avg", "discrete", "sin3d")) or "mrope" in wtype or "mlp" in wtype:
        raise V3DError(f"world_position_embedding_type {wtype!r}: only the shipped 'avg-discrete-sin3d' variant is on the accelerated path")
    for key, want in (("mm_spatial_pool_mode", "bilinear"), ("mm_newline_position", "grid"), ("mm_projector_type", "mlp2x_gelu")):
        got = getattr(c, key, want)
        if got != want:
            raise V3DError(f"config.{key} = {got!r}: only {want!r} is on the accelerated path")
    if int(getattr(c, "mm_spatial_pool_stride", 2)) != 2:
        raise V3DError("only mm_spatial_pool_stride 2 (27x27 -> 14x14) is on the accelerated path")
    pe_w = sd.get(VIT_PREFIX + "embeddings.patch_embedding.weight")
    if pe_w is None:
        raise V3DError("the state dict has no SigLIP tower (model.vision_tower.vision_tower.vision_model.*)")
    layers = 1 + max(int(k[len(VIT_PREFIX + "encoder.layers."):].split(".")[0]) for k in sd if k.startswith(VIT_PREFIX + "encoder.layers."))
    v_hidden, patch = pe_w.shape[0], pe_w.shape[-1]
    n_pos = sd[VIT_PREFIX + "embeddings.position_embedding.weight"].shape[0]
    side = int(round(n_pos ** 0.5))
    vit = VitConfig(hidden=v_hidden, inter=sd[VIT_PREFIX + "encoder.layers.0.mlp.fc1.weight"].shape[0], layers=layers,
                    heads=v_hidden // 72, image=side * patch, patch=patch, eps=1e-6)
    cap = max_positions or getattr(c, "v3d_max_positions", None) or min(int(getattr(c, "max_position_embeddings", 8192)), 8192)
    llm = LlmConfig(hidden=c.hidden_size, inter=c.intermediate_size, layers=c.num_hidden_layers, heads=c.num_attention_heads,
                    kv_heads=c.num_key_value_heads, vocab=sd["lm_head.weight"].shape[0], eps=float(c.rms_norm_eps),
                    rope_theta=float(getattr(c, "rope_theta", None) or (getattr(c, "rope_parameters", None) or {}).get("rope_theta", 1e6)),
                    max_pos=int(cap))
    return EngineConfig(vit=vit, llm=llm, min_xyz=tuple(getattr(c, "min_xyz_range", (-15, -15, -5))),
                        max_xyz=tuple(getattr(c, "max_xyz_range", (15, 15, 5))), voxel_size=float(getattr(c, "voxel_size", 0.1)),
                        ground_head_type=getattr(c, "ground_head_type", None) or "infonce",
                        object_feature_type=getattr(c, "object_feature_type", None) or "patch14-pe")


def drop_deleted_tower_layer(sd, tower_from_raw_siglip):
    """siglip_encoder.py:570 `del encoder.layers[-1:]` for weights that come from a raw SigLIP directory."""
    if not tower_from_raw_siglip:
        return sd
    pfx = VIT_PREFIX + "encoder.layers."
    last = max(int(k[len(pfx):].split(".")[0]) for k in sd if k.startswith(pfx))
    return {k: v for k, v in sd.items() if not k.startswith(f"{pfx}{last}.")}


def load_pretrained_model(model_path, model_base=None, model_name=None, device_map="auto", torch_dtype="float16",
                          attn_implementation="flash_attention_2", overwrite_config=None, **kwargs):
    """The LLaVA-Qwen branch of llava.model.builder.load_pretrained_model (builder.py:26-36, 206-228, 266-292) for callers that
    run without the reference checkout on the path: tokenizer, LlavaQwenForCausalLM.from_pretrained with the same keyword
    arguments, resize_token_embeddings, the tower's image processor, the context length.  Same return tuple."""
    from transformers import AutoTokenizer

    from llava.model import LlavaQwenConfig, LlavaQwenForCausalLM
    if model_base is not None or kwargs:
        raise NotImplementedError(f"only a full LLaVA-Qwen checkpoint directory is supported here (model_base / {sorted(kwargs)} given)")
    dt = {"float16": torch.float16, "bfloat16": torch.bfloat16}[torch_dtype] if isinstance(torch_dtype, str) else torch_dtype
    tokenizer = AutoTokenizer.from_pretrained(model_path)
    cfg = None
    if overwrite_config:
        cfg = LlavaQwenConfig.from_pretrained(model_path)
        for k, v in overwrite_config.items():
            setattr(cfg, k, v)
    model = LlavaQwenForCausalLM.from_pretrained(model_path, low_cpu_mem_usage=True, attn_implementation=attn_implementation, config=cfg,
                                                 device_map=device_map, torch_dtype=dt)
    # builder.py:266-275: the image patch / start / end tokens join the tokenizer before the tables are resized to len(tokenizer)
    # (mm_use_im_patch_token defaults to True upstream) - mirrored so that both loaders see the same vocabulary size
    if getattr(model.config, "mm_use_im_patch_token", True):
        tokenizer.add_tokens(["<im_patch>"], special_tokens=True)
    if getattr(model.config, "mm_use_im_start_end", False):
        tokenizer.add_tokens(["<im_start>", "<im_end>"], special_tokens=True)
    model.resize_token_embeddings(len(tokenizer))
    tower = model.get_vision_tower()
    if not tower.is_loaded:
        tower.load_model(device_map=device_map)
    for key in ("max_sequence_length", "max_position_embeddings", "tokenizer_model_max_length"):
        if hasattr(model.config, key):
            return tokenizer, model, tower.image_processor, getattr(model.config, key)
    return tokenizer, model, tower.image_processor, 2048
