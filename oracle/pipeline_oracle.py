"""CPU oracle for one whole (scene, question) pass: SigLIP tower -> projector -> coordinate pool /
voxelise -> bilinear pool + 3-D PE + newline -> splice into text embeddings -> Qwen2 prefill -> greedy
decode.  Composition of oracle/llm_oracle.py (torch) and oracle/v3d_oracle.py (numpy) in the order
LlavaMetaForCausalLM.prepare_inputs_labels_for_multimodal (llava/model/llava_arch.py:336-836, eval
branch) and LlavaQwenForCausalLM.generate (llava/model/language_model/llava_qwen.py:208-236) use.

TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py cpu_baseline)."""
import numpy as np
import torch

from . import llm_oracle as L
from . import v3d_oracle as O

KIND = {torch.float32: "f32", torch.float16: "f16", torch.bfloat16: "bf16"}
IMAGE_TOKEN_INDEX = -200   # llava/constants.py


def _np(t):
    return t.float().numpy()


def visual_sequence(sd, world_coords, feats, dtype):
    """feats [F,729,C] (dtype) + coords [F,384,384,3] (dtype) -> (ids int32 [F,14,14,3], tokens [F*210, C])."""
    kind = KIND[dtype]
    C = feats.shape[-1]
    d = torch.arange(C // 3, dtype=torch.float32)
    dim_t = (10000 ** (2 * (d // 2) / (C // 3))).numpy()      # position_encoding.py:24-25, torch pow on the host
    ids, seq = O.fused_visual_tokens(_np(world_coords), _np(feats), _np(sd["model.image_newline"].to(dtype)), kind, dim_t=dim_t)
    return ids, torch.from_numpy(seq).to(dtype)


def inputs_embeds(sd, input_ids, vis, dtype, box_input=None, coord_token_id=None):
    """llava_arch.py:684-745 for one sample with one <image> token.  box_input [1,3] + coord_token_id: the 3-D PE of the
    discretised box centre is added to the embedding of every <coord> text token (llava_arch.py:416-417, 697-700)."""
    ids = input_ids.tolist()
    at = ids.index(IMAGE_TOKEN_INDEX)
    emb = sd["model.embed_tokens.weight"].to(dtype)
    text = torch.cat([input_ids[:at], input_ids[at + 1:]])
    e = emb[text]
    if box_input is not None and coord_token_id is not None and bool((text == coord_token_id).any()):
        kind = KIND[dtype]
        C = emb.shape[1]
        centre = torch.from_numpy(O.discrete_coords(_np(box_input.to(dtype).reshape(-1, 3)[:1]), kind)).to(dtype)
        d = torch.arange(C // 3, dtype=torch.float32)
        dim_t = (10000 ** (2 * (d // 2) / (C // 3))).numpy()
        pe = torch.from_numpy(O.sin3d_pe(_np(centre)[None], C, kind, dim_t=dim_t)[0]).to(dtype)        # [1, C]
        e = e.clone()
        e[text == coord_token_id] += pe[0]
    return torch.cat([e[:at], vis, e[at:]], 0)


def scene_forward(sd, cfg, input_ids, images, world_coords, dtype, max_new_tokens=4):
    """Returns a dict of the intermediates the GPU tests compare against."""
    w = {k: v.to(dtype) for k, v in sd.items()}
    tower = L.siglip_tower(images.to(dtype), w, cfg["vit_layers"], cfg["vit_heads"])
    feats = L.projector(tower, w)
    ids, vis = visual_sequence(w, world_coords.to(dtype), feats, dtype)
    x = inputs_embeds(w, input_ids, vis, dtype)[None]
    logits, hidden, kv = L.qwen2_model(x, w, cfg)
    out = dict(tower=tower, feats=feats, ids=ids, vis=vis, embeds=x[0], logits_last=logits[0, -1], hidden_last=hidden[0, -1])
    toks = []
    tok = int(torch.argmax(logits[0, -1]))
    pos = x.shape[1]
    step_logits = []
    for _ in range(max_new_tokens):
        toks.append(tok)
        if len(toks) == max_new_tokens:
            break
        xe = w["model.embed_tokens.weight"][tok][None, None]
        lg, _, kv = L.qwen2_model(xe, w, cfg, past=kv, pos0=pos)
        step_logits.append(lg[0, -1])
        pos += 1
        tok = int(torch.argmax(lg[0, -1]))
    out["tokens"] = toks
    out["step_logits"] = step_logits
    return out


def scene_ground(sd, cfg, input_ids, ground_index, images, world_coords, objects, dtype):
    """ScanRefer-style forward: prefill + object-proposal features + infonce scores
    (llava_arch.py:351-376, 479-501; llava_qwen.py:280-300)."""
    w = {k: v.to(dtype) for k, v in sd.items()}
    coords = world_coords.to(dtype)
    boxes = objects.to(dtype)
    tower = L.siglip_tower(images.to(dtype), w, cfg["vit_layers"], cfg["vit_heads"])
    feats = L.projector(tower, w)
    ids, vis = visual_sequence(w, coords, feats, dtype)
    x = inputs_embeds(w, input_ids, vis, dtype)[None]
    _, hidden, _ = L.qwen2_model(x, w, cfg)
    at = input_ids.tolist().index(IMAGE_TOKEN_INDEX)
    gpos = ground_index if ground_index < at else ground_index + vis.shape[0] - 1
    masks = L.object_patch_mask(coords, boxes)
    kind = KIND[dtype]
    centres = torch.from_numpy(O.discrete_coords(_np(boxes[:, :3]), kind)).to(dtype)
    C = feats.shape[-1]
    d = torch.arange(C // 3, dtype=torch.float32)
    dim_t = (10000 ** (2 * (d // 2) / (C // 3))).numpy()
    pe = torch.from_numpy(O.sin3d_pe(_np(centres)[None], C, kind, dim_t=dim_t)[0]).to(dtype)
    objf = L.object_features(feats, masks, pe)
    scores = L.infonce_scores(objf, w["ground_head_zero_target"], hidden[0, gpos][None], w)
    return dict(masks=masks, objf=objf, scores=scores, query=hidden[0, gpos])
