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
