"""Loader for tests/golden/tiny_model.npz (oracle/gen_golden.py g_tiny_model): the reference's own
LlavaQwenForCausalLM, random-init at tiny widths, run through prepare_inputs_labels_for_multimodal + forward on CPU.
Weights are stored as raw bf16 bits (every value bf16-representable), images / coordinates at 48 x 48 to be repeated
8x along both image axes (the generator applied the same expansion before calling the reference)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DT = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}
ORACLE_CFG = dict(layers=2, heads=2, kv_heads=1, rope_theta=1e6, eps=1e-6, vit_layers=2, vit_heads=2)
GROUND_TOKEN = 318


def bits_to_f32(a):
    return torch.from_numpy((a.astype(np.uint32) << 16).view(np.float32).copy())


def load():
    return np.load(os.path.join(GOLDEN, "tiny_model.npz"))


def state_dict(g):
    return {k[2:]: bits_to_f32(g[k]) for k in g.files if k.startswith("w.")}


def engine_config():
    from v3d.engine import EngineConfig, LlmConfig, VitConfig
    return EngineConfig(vit=VitConfig(hidden=144, inter=272, layers=2, heads=2),
                        llm=LlmConfig(hidden=256, inter=384, layers=2, heads=2, kv_heads=1, vocab=320, max_pos=2048))


def case_inputs(g, case):
    img = torch.from_numpy(g[case + "_img_lo"]).repeat_interleave(8, 2).repeat_interleave(8, 3)
    wc = torch.from_numpy(g[case + "_wc_lo"]).repeat_interleave(8, 1).repeat_interleave(8, 2)
    return dict(images=img, world_coords=wc, boxes=torch.from_numpy(g[case + "_boxes"]),
                ids=torch.from_numpy(g[case + "_ids"]), gids=torch.from_numpy(g[case + "_gids"]),
                glabels=torch.from_numpy(g[case + "_glabels"]))


def expected(g, case, kind):
    raw = (lambda a: bits_to_f32(a)) if kind == "bf16" else (lambda a: torch.from_numpy(a).float())
    return dict(embeds=raw(g[f"{case}_embeds_{kind}"]), logits=torch.from_numpy(g[f"{case}_logits_{kind}"]),
                tokens=g[f"{case}_tokens_{kind}"].tolist(), step_logits=torch.from_numpy(g[f"{case}_step_logits_{kind}"]),
                scores=raw(g[f"{case}_scores_{kind}"]))


HF_CONFIG = dict(
    architectures=["LlavaQwenForCausalLM"], model_type="llava_qwen", vocab_size=320, hidden_size=256, intermediate_size=384,
    num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=4096, rms_norm_eps=1e-6,
    rope_theta=1000000.0, use_sliding_window=False, attention_dropout=0.0, tie_word_embeddings=False, eos_token_id=310,
    mm_vision_tower="google/siglip-so400m-patch14-384", mm_projector_type="mlp2x_gelu", mm_hidden_size=144,
    mm_use_im_start_end=False, mm_use_im_patch_token=False, mm_patch_merge_type="spatial_unpad", mm_newline_position="grid",
    mm_spatial_pool_mode="bilinear", mm_spatial_pool_stride=2, world_position_embedding_type="avg-discrete-sin3d",
    voxel_size=0.1, min_xyz_range=[-15, -15, -5], max_xyz_range=[15, 15, 5], ground_head_type="infonce",
    ground_head_temperature=0.07, object_feature_type="patch14-pe", ground_token_ids=[GROUND_TOKEN], coord_token_ids=[317],
    tokenizer_model_max_length=32768, v3d_max_positions=2048)


def write_checkpoint(path, g, shards=2):
    """A LLaVA-Qwen checkpoint directory as `load_pretrained_model` expects it (builder.py:206-228): config.json
    (model_type llava_qwen + the 3-D keys train_3d.py:1429-1459 persists), safetensors shards with an index, a tokenizer."""
    import json

    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(HF_CONFIG, f)
    sd = {k: v.to(torch.bfloat16).contiguous() for k, v in state_dict(g).items()}      # every value is bf16-representable: exact
    keys = sorted(sd)
    weight_map = {}
    for i in range(shards):
        name = f"model-{i + 1:05d}-of-{shards:05d}.safetensors"
        part = {k: sd[k] for k in keys[i::shards]}
        save_file(part, os.path.join(path, name), metadata={"format": "pt"})
        weight_map.update({k: name for k in part})
    with open(os.path.join(path, "model.safetensors.index.json"), "w") as f:
        json.dump({"metadata": {}, "weight_map": weight_map}, f)
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    # 320 ids: words t0..t299, the ChatML words and specials the eval harness needs, the rest of the table as t3xx
    names = {300: "system", 301: "user", 302: "assistant", 303: "\n", 304: "You", 305: "are", 306: "a", 307: "helpful",
             308: "assistant.", 309: "<|im_start|>", 310: "<|im_end|>"}
    vocab = {names.get(i, f"t{i}"): i for i in range(320)}
    tok = Tokenizer(models.WordLevel(vocab, unk_token="t0"))
    tok.pre_tokenizer = pre_tokenizers.Sequence([pre_tokenizers.Split("\n", "isolated"), pre_tokenizers.Split(" ", "removed")])
    PreTrainedTokenizerFast(tokenizer_object=tok, eos_token="<|im_end|>", pad_token="t0", unk_token="t0",
                            additional_special_tokens=["<|im_start|>", "<|im_end|>"]).save_pretrained(path)
    return path
