"""CPU oracle (plain torch, any float dtype) for the dense part of the hot path: SigLIP tower,
mlp2x_gelu projector, Qwen2 decoder (RMSNorm, rotary, eager causal GQA attention, SwiGLU), LM head
and the multimodal splice.

TEST INFRASTRUCTURE ONLY - imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg, never by the product path.  Each function restates one reference module and cites it
(paths relative to the reference checkout).  Running these functions on tensors of dtype
bf16 / f16 / f32 reproduces the reference's own rounding points, because the reference *is* these
torch ops.  Pinned against golden vectors produced by the reference modules themselves
(oracle/gen_golden.py: g_llm / g_vit), see tests/test_oracle_llm_golden.py.

Weights are passed as plain dicts keyed like the reference's state_dict.
"""
import math

import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------- Qwen2


def rmsnorm(x, weight, eps=1e-6):
    """llava/model/language_model/qwen2/modeling_qwen2.py:85-90."""
    dt = x.dtype
    h = x.to(torch.float32)
    var = h.pow(2).mean(-1, keepdim=True)
    h = h * torch.rsqrt(var + eps)
    return weight * h.to(dt)


def inv_freq(head_dim, base):
    """modeling_qwen2.py:100."""
    return 1.0 / (base ** (torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim))


def rotary_cos_sin(positions, head_dim, base, dtype):
    """modeling_qwen2.py:106-129 with the three position rows equal (:1003-1004): [S, head_dim]."""
    f = inv_freq(head_dim, base)[None, :].float() * positions[:, None].float()     # outer product, one f32 rounding
    emb = torch.cat((f, f), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def rotate_half(x):
    """modeling_qwen2.py:133-137."""
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def apply_rope(q, k, cos, sin):
    """modeling_qwen2.py:141-173 (mrope sections re-assemble the same row when the 3 id rows are equal).
    q [B,H,S,D], k [B,Hkv,S,D], cos/sin [S,D]."""
    c, s = cos[None, None], sin[None, None]
    return (q * c) + (rotate_half(q) * s), (k * c) + (rotate_half(k) * s)


def repeat_kv(x, n_rep):
    """modeling_qwen2.py:193-202."""
    b, h, s, d = x.shape
    if n_rep == 1:
        return x
    return x[:, :, None].expand(b, h, n_rep, s, d).reshape(b, h * n_rep, s, d)


def causal_mask(q_len, kv_len, dtype):
    """transformers' _prepare_4d_causal_attention_mask: 0 where key <= query position, dtype-min elsewhere."""
    past = kv_len - q_len
    i = torch.arange(q_len)[:, None] + past
    j = torch.arange(kv_len)[None, :]
    m = torch.zeros(q_len, kv_len, dtype=dtype)
    return m.masked_fill(j > i, torch.finfo(dtype).min)[None, None]


def eager_attention(q, k, v, n_rep, mask):
    """modeling_qwen2.py:289-311: softmax(QK^T/sqrt(d) + mask) in f32, cast back, @ V.  [B,H,S,D] layout."""
    k = repeat_kv(k, n_rep)
    v = repeat_kv(v, n_rep)
    w = torch.matmul(q, k.transpose(2, 3)) / math.sqrt(q.shape[-1])
    if mask is not None:
        w = w + mask
    w = F.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
    return torch.matmul(w, v)


def qwen2_attention(x, w, pfx, n_heads, n_kv, positions, rope_base, past_kv=None):
    """Qwen2Attention.forward, modeling_qwen2.py:248-327.  x [B,S,H]; returns (out, (k, v))."""
    B, S, H = x.shape
    hd = H // n_heads
    q = F.linear(x, w[pfx + "q_proj.weight"], w[pfx + "q_proj.bias"]).view(B, S, n_heads, hd).transpose(1, 2)
    k = F.linear(x, w[pfx + "k_proj.weight"], w[pfx + "k_proj.bias"]).view(B, S, n_kv, hd).transpose(1, 2)
    v = F.linear(x, w[pfx + "v_proj.weight"], w[pfx + "v_proj.bias"]).view(B, S, n_kv, hd).transpose(1, 2)
    cos, sin = rotary_cos_sin(positions, hd, rope_base, x.dtype)
    q, k = apply_rope(q, k, cos, sin)
    if past_kv is not None:
        k = torch.cat([past_kv[0], k], dim=2)
        v = torch.cat([past_kv[1], v], dim=2)
    mask = causal_mask(S, k.shape[2], x.dtype)
    o = eager_attention(q, k, v, n_heads // n_kv, mask)
    o = o.transpose(1, 2).contiguous().reshape(B, S, H)
    return F.linear(o, w[pfx + "o_proj.weight"]), (k, v)


def qwen2_mlp(x, w, pfx):
    """Qwen2MLP.forward, modeling_qwen2.py:188-189."""
    return F.linear(F.silu(F.linear(x, w[pfx + "gate_proj.weight"])) * F.linear(x, w[pfx + "up_proj.weight"]),
                    w[pfx + "down_proj.weight"])


def qwen2_layer(x, w, pfx, n_heads, n_kv, positions, rope_base, eps, past_kv=None):
    """Qwen2DecoderLayer.forward, modeling_qwen2.py:770-801."""
    h = rmsnorm(x, w[pfx + "input_layernorm.weight"], eps)
    a, kv = qwen2_attention(h, w, pfx + "self_attn.", n_heads, n_kv, positions, rope_base, past_kv)
    x = x + a
    h = rmsnorm(x, w[pfx + "post_attention_layernorm.weight"], eps)
    return x + qwen2_mlp(h, w, pfx + "mlp."), kv


def qwen2_model(inputs_embeds, w, cfg, past=None, pos0=0):
    """Qwen2Model.forward + lm_head (modeling_qwen2.py:952-1097, 1188-1192).  Returns (logits f32, hidden, kv list)."""
    x = inputs_embeds
    S = x.shape[1]
    positions = torch.arange(pos0, pos0 + S)
    kvs = []
    for i in range(cfg["layers"]):
        x, kv = qwen2_layer(x, w, f"model.layers.{i}.", cfg["heads"], cfg["kv_heads"], positions, cfg["rope_theta"],
                            cfg["eps"], None if past is None else past[i])
        kvs.append(kv)
    x = rmsnorm(x, w["model.norm.weight"], cfg["eps"])
    logits = F.linear(x, w["lm_head.weight"]).float()
    return logits, x, kvs


# ----------------------------------------------------------------------------- SigLIP + projector


def siglip_embeddings(pixel_values, w, pfx, patch=14):
    """SigLipVisionEmbeddings.forward, siglip_encoder.py:168-174."""
    p = F.conv2d(pixel_values, w[pfx + "patch_embedding.weight"], w[pfx + "patch_embedding.bias"], stride=patch)
    e = p.flatten(2).transpose(1, 2)
    return e + w[pfx + "position_embedding.weight"][None]


def siglip_attention(x, w, pfx, n_heads):
    """SigLipAttention.forward, siglip_encoder.py:197-239."""
    B, S, E = x.shape
    hd = E // n_heads
    q = F.linear(x, w[pfx + "q_proj.weight"], w[pfx + "q_proj.bias"]).view(B, S, n_heads, hd).transpose(1, 2)
    k = F.linear(x, w[pfx + "k_proj.weight"], w[pfx + "k_proj.bias"]).view(B, S, n_heads, hd).transpose(1, 2)
    v = F.linear(x, w[pfx + "v_proj.weight"], w[pfx + "v_proj.bias"]).view(B, S, n_heads, hd).transpose(1, 2)
    a = torch.matmul(q, k.transpose(2, 3)) * (hd ** -0.5)
    a = F.softmax(a, dim=-1, dtype=torch.float32).to(q.dtype)
    o = torch.matmul(a, v).transpose(1, 2).contiguous().reshape(B, S, E)
    return F.linear(o, w[pfx + "out_proj.weight"], w[pfx + "out_proj.bias"])


def siglip_layer(x, w, pfx, n_heads, eps=1e-6):
    """SigLipEncoderLayer.forward, siglip_encoder.py:276-305 (hidden_act gelu_pytorch_tanh)."""
    E = x.shape[-1]
    h = F.layer_norm(x, (E,), w[pfx + "layer_norm1.weight"], w[pfx + "layer_norm1.bias"], eps)
    x = x + siglip_attention(h, w, pfx + "self_attn.", n_heads)
    h = F.layer_norm(x, (E,), w[pfx + "layer_norm2.weight"], w[pfx + "layer_norm2.bias"], eps)
    h = F.linear(h, w[pfx + "mlp.fc1.weight"], w[pfx + "mlp.fc1.bias"])
    h = F.gelu(h, approximate="tanh")
    h = F.linear(h, w[pfx + "mlp.fc2.weight"], w[pfx + "mlp.fc2.bias"])
    return x + h


def siglip_tower(pixel_values, w, n_layers, n_heads, pfx="model.vision_tower.vision_tower.vision_model."):
    """SigLipVisionTower.forward, siglip_encoder.py:576-589: hidden_states[-1] of the truncated encoder
    (last layer deleted at :570, post_layernorm/head not applied to the returned features)."""
    x = siglip_embeddings(pixel_values, w, pfx + "embeddings.")
    for i in range(n_layers):
        x = siglip_layer(x, w, pfx + f"encoder.layers.{i}.", n_heads)
    return x


def projector(x, w, pfx="model.mm_projector."):
    """mlp2x_gelu, multimodal_projector/builder.py:41-48: Linear -> nn.GELU() (erf) -> Linear."""
    h = F.gelu(F.linear(x, w[pfx + "0.weight"], w[pfx + "0.bias"]))
    return F.linear(h, w[pfx + "2.weight"], w[pfx + "2.bias"])


# ----------------------------------------------------------------------------- object proposals / grounding


def object_patch_mask(world_coords, boxes, cell=14, thresh_frac=0.5):
    """llava/model/llava_arch.py:351-376, object_feature_type 'patch14': per proposal box, the ViT patches
    (27 x 27 cells of 14 x 14 pixels over the first 378 rows/cols) with >= int(14*14*0.5) pixels inside the box.
    ('patch27', :367-371: cell=27, thresh_frac=0.25 - the 14 x 14 grid of the pooled tokens.)
    world_coords [F,384,384,3], boxes [n,6] (centre, size), both in the model dtype -> bool [n,F,g,g], g = 378 // cell."""
    F_ = world_coords.shape[0]
    g = 378 // cell
    wc = world_coords[:, :378, :378, :].reshape(-1, g, cell, g, cell, 3).transpose(2, 3).flatten(3, 4)
    out = []
    for box in boxes:
        lo = box[:3] - box[3:] / 2
        hi = box[:3] + box[3:] / 2
        inside = torch.all((lo <= wc) & (wc <= hi), dim=-1)
        out.append(inside.sum(dim=3) >= int(cell * cell * thresh_frac))
    return torch.stack(out) if out else torch.zeros((0, F_, g, g), dtype=torch.bool)


def object_features(encoded, masks, centre_pe=None):
    """llava_arch.py:479-501: mean of the projector rows of the selected patches (zeros if none), plus the
    3-D PE of the (discretised) box centre.  encoded [F,729,C]; masks bool [n,F,27,27]; centre_pe [n,C] or None."""
    C = encoded.shape[-1]
    feats = []
    for m in masks:
        sel = encoded[m.view(-1, encoded.shape[1])]          # 729 ViT rows ('patch14') or 196 pooled rows ('patch27', :485-486)
        feats.append(sel.mean(dim=0) if len(sel) else torch.zeros(C, dtype=encoded.dtype))
    f = torch.stack(feats)
    if centre_pe is not None:
        f = f + centre_pe
    return f


def ground_head(x, w, pfx):
    """nn.Sequential(Linear, ReLU, LayerNorm, Linear), llava_qwen.py:93-104."""
    h = F.relu(F.linear(x, w[pfx + "0.weight"], w[pfx + "0.bias"]))
    h = F.layer_norm(h, (h.shape[-1],), w[pfx + "2.weight"], w[pfx + "2.bias"], 1e-5)
    return F.linear(h, w[pfx + "3.weight"], w[pfx + "3.bias"])


def mlp_scores(object_feats, query_hidden, w, pfx="ground_head."):
    """predict_box, ground_head_type 'mlp', llava_qwen.py:59-71, 283-285: the MLP on the <ground> hidden state, then the
    elementwise product with the object features summed over the channels -> [n]."""
    return (ground_head(query_hidden, w, pfx).squeeze(0) * object_feats).sum(dim=-1)


def _ln_relu_head(x, w, pfx, last=True):
    """nn.Sequential(Linear, LayerNorm, ReLU[, Linear]), llava_qwen.py:74-91."""
    h = F.linear(x, w[pfx + "0.weight"], w[pfx + "0.bias"])
    h = F.relu(F.layer_norm(h, (h.shape[-1],), w[pfx + "1.weight"], w[pfx + "1.bias"], 1e-5))
    return F.linear(h, w[pfx + "3.weight"], w[pfx + "3.bias"]) if last else h


def score_scores(object_feats, query_hidden, w):
    """predict_box, ground_head_type 'score', llava_qwen.py:72-91, 286-292: obj and query MLPs (width 1024), their product,
    the scoring MLP -> [n]."""
    obj = _ln_relu_head(object_feats.to(query_hidden.dtype), w, "ground_head_obj.")
    q = _ln_relu_head(query_hidden, w, "ground_head_query.")
    h = _ln_relu_head(obj * q, w, "ground_head_score.", last=False)
    return F.linear(h, w["ground_head_score.3.weight"], w["ground_head_score.3.bias"]).squeeze(1)


def seeded_ground_head(kind, hidden, seed):
    """Weights of an 'mlp' / 'score' grounding head from a seed (the 'score' head's three 1024-wide MLPs are too large to store in
    a fixture): tests/golden/ground_variants.npz holds the seeds, a checksum of what they produced and the REFERENCE's scores."""
    g = torch.Generator().manual_seed(seed)
    w = {}

    def lin(name, n_out, n_in, s):
        w[name + ".weight"] = (torch.randn(n_out, n_in, generator=g) * s).to(torch.bfloat16).float()
        w[name + ".bias"] = (torch.randn(n_out, generator=g) * 0.1).to(torch.bfloat16).float()

    def ln(name, n):
        w[name + ".weight"] = (1 + 0.1 * torch.randn(n, generator=g)).to(torch.bfloat16).float()
        w[name + ".bias"] = (0.1 * torch.randn(n, generator=g)).to(torch.bfloat16).float()

    if kind == "mlp":
        lin("ground_head.0", hidden, hidden, 0.08); ln("ground_head.2", hidden); lin("ground_head.3", hidden, hidden, 0.08)
    elif kind == "score":
        for pfx, n_in in (("ground_head_obj", hidden), ("ground_head_query", hidden), ("ground_head_score", 1024)):
            lin(pfx + ".0", 1024, n_in, 0.08 if n_in == hidden else 0.03)
            ln(pfx + ".1", 1024)
            lin(pfx + ".3", 1 if pfx == "ground_head_score" else 1024, 1024, 0.03)
    else:
        raise ValueError(kind)
    return w


def infonce_scores(object_feats, zero_target, query_hidden, w, obj_pfx="ground_head_obj.", q_pfx="ground_head_query."):
    """predict_box, ground_head_type 'infonce', llava_qwen.py:294-300: cosine similarity of the projected
    object features (+ the learned zero-target row) with the projected <ground> hidden state -> [n+1]."""
    of = torch.cat([object_feats, zero_target[None]], 0)
    o = F.normalize(ground_head(of.to(query_hidden.dtype), w, obj_pfx))
    q = F.normalize(ground_head(query_hidden, w, q_pfx))
    return (o * q).sum(dim=-1)
