"""The loader-produced model object on the GPU: a tiny random checkpoint the test writes itself (safetensors shards + index,
config.json with model_type llava_qwen, a tokenizer) is loaded with exactly the calls `load_pretrained_model` makes
(llava/model/builder.py:206-228, 266-292: AutoTokenizer, LlavaQwenConfig.from_pretrained + overwrite_config,
LlavaQwenForCausalLM.from_pretrained(path, low_cpu_mem_usage=True, attn_implementation=..., config=..., device_map="auto",
torch_dtype=float16), resize_token_embeddings, get_vision_tower().is_loaded / image_processor) - the reference checkout
itself does not exist on the GPU box, so the call sequence is restated here; tests/test_overlay_loader.py runs the real
builder.py over the overlay in the build container.  The weights are those of tests/golden/tiny_model.npz, so generate /
forward are compared with the REFERENCE model's own outputs."""
import os

import pytest
import torch

import tiny_model_fixture as TM

pytestmark = pytest.mark.gpu


def rel_err(got, want):
    got, want = got.float().cpu(), want.float()
    return ((got - want).norm() / want.norm().clamp_min(1e-9)).item()


def load_like_builder(model_path, overwrite_config=None, torch_dtype=torch.float16, attn_implementation="flash_attention_2"):
    from llava.model import LlavaQwenForCausalLM            # `from llava.model import *`, builder.py:22
    from llava.model.language_model.llava_qwen import LlavaQwenConfig          # builder.py:220
    from transformers import AutoTokenizer
    kwargs = {"device_map": "auto", "torch_dtype": torch_dtype}                 # builder.py:27-36
    tokenizer = AutoTokenizer.from_pretrained(model_path)                      # :207
    if overwrite_config is not None:                                           # :221-226
        llava_cfg = LlavaQwenConfig.from_pretrained(model_path)
        for k, v in overwrite_config.items():
            setattr(llava_cfg, k, v)
        model = LlavaQwenForCausalLM.from_pretrained(model_path, low_cpu_mem_usage=True, attn_implementation=attn_implementation,
                                                     config=llava_cfg, **kwargs)
    else:                                                                      # :228
        model = LlavaQwenForCausalLM.from_pretrained(model_path, low_cpu_mem_usage=True, attn_implementation=attn_implementation, **kwargs)
    model.resize_token_embeddings(len(tokenizer))                              # :275
    vision_tower = model.get_vision_tower()                                    # :277-283
    if not vision_tower.is_loaded:
        vision_tower.load_model(device_map="auto")
    image_processor = vision_tower.image_processor
    context_len = model.config.max_position_embeddings                         # :285-292
    return tokenizer, model, image_processor, context_len


@pytest.fixture(scope="module")
def g():
    return TM.load()


@pytest.fixture(scope="module")
def loaded(g, tmp_path_factory):
    path = TM.write_checkpoint(str(tmp_path_factory.mktemp("ckpt") / "llava_qwen_tiny"), g)
    return load_like_builder(path, overwrite_config={"tie_word_embeddings": False, "use_cache": True, "vocab_size": 320})


def _call_args(g, case, dt=torch.float16):
    inp = TM.case_inputs(g, case)
    images = inp["images"][None].to(dt).cuda()                                  # model_scanqa.py:163-165: .half().to(device)
    video_dict = {"world_coords": inp["world_coords"][None].to(dt).cuda(), "box_input": torch.Tensor([]),
                  "objects": inp["boxes"][None].to(dt).cuda()}
    return inp, images, video_dict


def test_loader_object_has_the_reference_surface(loaded):
    tokenizer, model, image_processor, context_len = loaded
    assert type(model).__name__ == "LlavaQwenForCausalLM" and model.config.model_type == "llava_qwen"
    assert model.device.type == "cuda" and model.dtype == torch.float16 and context_len == 4096
    assert model.get_vision_tower().num_patches_per_side == 27 and model.get_vision_tower().is_loaded
    assert type(image_processor).__name__ == "SigLipImageProcessor" and image_processor.crop_size["width"] == 384
    assert model.get_model().image_newline.shape == (256,)
    assert model.config.ground_token_ids == [TM.GROUND_TOKEN]


def test_generate_through_the_loaded_model_matches_the_reference(loaded, g):
    """model_scanqa.py:173-185's call; tokens against the reference model's own greedy continuation (f16)."""
    _, model, _, _ = loaded
    inp, images, video_dict = _call_args(g, "F2")
    want = TM.expected(g, "F2", "f16")
    out = model.generate(inp["ids"][None].cuda(), images=images, modalities="video", do_sample=False, temperature=0, top_p=None,
                         num_beams=1, max_new_tokens=4, use_cache=True, video_dict=video_dict)
    assert out.shape[0] == 1 and out.dtype == torch.int64
    ref_steps = [want["logits"]] + list(want["step_logits"])
    for i, (a, b) in enumerate(zip(out[0].tolist(), want["tokens"])):
        if a != b:
            top2 = torch.topk(ref_steps[i], 2).values
            assert (top2[0] - top2[1]).item() < 0.05 * ref_steps[i].abs().max().item(), f"token {i}: {a} != {b}"
            break
    # EOS from the config stops the answer (and is returned as its last token, as HF does)
    first = int(out[0, 0])
    model.config.eos_token_id = first
    try:
        cut = model.generate(inp["ids"][None].cuda(), images=images, modalities="video", max_new_tokens=4, video_dict=video_dict)
    finally:
        model.config.eos_token_id = 319
    assert cut[0].tolist() == [first]


def test_stopping_criteria_and_refused_arguments(loaded, g):
    from llava.mm_utils import KeywordsStoppingCriteria
    tokenizer, model, _, _ = loaded
    inp, images, video_dict = _call_args(g, "F2")
    ids = inp["ids"][None].cuda()
    free = model.generate(ids, images=images, modalities="video", max_new_tokens=4, video_dict=video_dict)[0].tolist()
    word = tokenizer.decode([free[1]])
    crit = KeywordsStoppingCriteria([word], tokenizer, ids[:, :0])
    out = model.generate(ids, images=images, modalities="video", max_new_tokens=4, video_dict=video_dict, stopping_criteria=[crit])
    assert out[0].tolist() == free[: free.index(free[1]) + 1]                    # stopped right after the first keyword token
    with pytest.raises(TypeError):
        model.generate(ids, images=images, modalities="video", video_dict=video_dict, no_repeat_ngram_size=3)
    with pytest.raises(NotImplementedError):
        model.generate(ids, images=images, modalities="video", video_dict=video_dict, do_sample=True, temperature=0.7)
    with pytest.raises(NotImplementedError):
        model.generate(ids, images=images, modalities="video", video_dict=video_dict, num_beams=4)
    with pytest.raises(TypeError):
        model(ids, images=images, modalities="video", video_dict=video_dict, some_flag=True)
    from v3d._native import V3DError
    with pytest.raises(V3DError):                                                # no write past the KV cache / residual stream
        model.engine.generate(inp["ids"], images[0], video_dict["world_coords"][0], max_new_tokens=4096)


def test_plain_and_grounding_forward_match_the_reference(loaded, g):
    _, model, _, _ = loaded
    inp, images, video_dict = _call_args(g, "F2")
    want = TM.expected(g, "F2", "f16")
    out = model(inp["ids"][None].cuda(), images=images, modalities="video", video_dict=video_dict)
    assert tuple(out.logits.shape) == (1, 431, 320) and out.logits.dtype == torch.float32
    assert rel_err(out.logits[0, -1], want["logits"]) < 9e-3
    # labels -> the shifted cross-entropy of Qwen2ForCausalLM.forward over the re-aligned labels, against torch on the returned logits
    lab = torch.full_like(inp["ids"], -100)
    lab[-4:] = inp["ids"][-4:]
    out2 = model(inp["ids"][None].cuda(), images=images, modalities="video", video_dict=video_dict, labels=lab[None].cuda())
    full = torch.cat([lab[:5], torch.full((420,), -100), lab[6:]])
    ref_loss = torch.nn.functional.cross_entropy(out2.logits[0, :-1].float().cpu(), full[1:], ignore_index=-100)
    assert abs(out2.loss.item() - ref_loss.item()) < 1e-5 * max(1.0, ref_loss.item())
    tup = model.prepare_inputs_labels_for_multimodal(inp["ids"][None].cuda(), None, None, None, None, images, ["video"], None, video_dict)
    assert tup[0] is None and tup[1] is None and tup[2] is None and tup[5] is None and tup[6] is None
    assert tuple(tup[4].shape) == (1, 431, 256) and rel_err(tup[4][0], want["embeds"]) < 3e-3
    assert torch.equal(model.encode_images(images[0]), model.engine.feat[: 2 * 729].view(2, 729, 256))
    # ScanRefer / Multi3DRefer call (model_scanrefer.py:165-173)
    loss, scores = model(inp["gids"][None].cuda(), images=images, modalities="video", video_dict=video_dict,
                         labels=inp["glabels"][None].cuda(), use_object_proposals=True, box_labels=None)
    assert loss is None and scores.shape == (6,)
    assert float((scores.float().cpu() - want["scores"]).abs().max()) < 6e-3
    # labels come back re-aligned to the spliced sequence
    tup = model.prepare_inputs_labels_for_multimodal(inp["gids"][None].cuda(), None, None, None, inp["glabels"][None].cuda(), images,
                                                     ["video"], None, video_dict, use_object_proposals=True)
    lab = tup[5][0].cpu()
    assert lab.shape[0] == tup[4].shape[1] == 432 and int((lab == TM.GROUND_TOKEN).nonzero()[0]) == 10 + 420 - 1
    assert tup[6].shape == (5, 256) and tup[7].shape == (5, 6)


def test_vocabulary_growth_and_tied_head(g, tmp_path):
    """(1) builder.py:282-287 adds <im_patch> (mm_use_im_patch_token defaults to True) before resize_token_embeddings(len(tokenizer)):
    growing the tables must work - new rows = mean of the old ones, never the argmax, the answers unchanged; the fallback loader
    (v3d.loader.load_pretrained_model) mirrors the add_tokens step; (2) a checkpoint saved with tied embeddings has no lm_head.weight."""
    import json
    from safetensors.torch import load_file, save_file
    from v3d import loader as L
    path = TM.write_checkpoint(str(tmp_path / "llava_qwen_grow"), g)
    cfg = json.load(open(os.path.join(path, "config.json")))
    cfg["mm_use_im_patch_token"] = True
    json.dump(cfg, open(os.path.join(path, "config.json"), "w"))
    tok, model, _, _ = L.load_pretrained_model(path, None, "llava_qwen_grow")
    assert len(tok) == 321 and model.engine.cfg.llm.vocab == 321 and model.engine.embed.shape[0] == 321
    base = load_like_builder(TM.write_checkpoint(str(tmp_path / "llava_qwen_base"), g))[1]
    inp, images, video_dict = _call_args(g, "F2")
    kw = dict(images=images, modalities="video", do_sample=False, num_beams=1, max_new_tokens=5, use_cache=True, video_dict=video_dict)
    a = model.generate(inp["ids"][None].cuda(), **kw)
    b = base.generate(inp["ids"][None].cuda(), **kw)
    assert torch.equal(a, b) and int(a.max()) < 320
    assert torch.equal(model.engine.embed[:320], base.engine.embed[:320])
    assert torch.allclose(model.engine.embed[320].float(), base.engine.embed[:320].float().mean(0), atol=2e-3)
    # tied checkpoint: drop lm_head.weight from the shards, make it equal to the embedding table
    tied = TM.write_checkpoint(str(tmp_path / "llava_qwen_tied"), g)
    idx = json.load(open(os.path.join(tied, "model.safetensors.index.json")))
    shard = idx["weight_map"].pop("lm_head.weight")
    sd = load_file(os.path.join(tied, shard))
    del sd["lm_head.weight"]
    save_file(sd, os.path.join(tied, shard), metadata={"format": "pt"})
    json.dump(idx, open(os.path.join(tied, "model.safetensors.index.json"), "w"))
    cfg = json.load(open(os.path.join(tied, "config.json")))
    cfg["tie_word_embeddings"] = True
    json.dump(cfg, open(os.path.join(tied, "config.json"), "w"))
    m2 = load_like_builder(tied)[1]
    assert torch.equal(m2.engine.l_head[:320], m2.engine.embed[:320])
