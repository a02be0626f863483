"""The overlay as a drop-in for the reference's loader path (CPU side; the GPU half is tests/test_gpu_loader.py).

1. With the overlay AHEAD of the reference checkout on PYTHONPATH (the documented setup) the
   reference's own `llava.model.builder` imports, finds LlavaQwenForCausalLM / LlavaQwenConfig where builder.py:22,220
   look for them, and `load_pretrained_model(<tiny checkpoint>)` travels through the reference's code into the overlay's
   from_pretrained - which, without a GPU, fails loudly (no CPU fallback).  Needs the reference checkout: skipped where
   /root/reference does not exist (the GPU box).
2. Without the reference: the overlay package alone exports the names, registers them with the HF Auto classes and
   refuses unknown arguments instead of swallowing them."""
import os
import subprocess
import sys
import textwrap

import pytest

import tiny_model_fixture as TM

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OVERLAY = os.path.join(ROOT, "video-3d-llm_amd")
REF = os.environ.get("V3D_REFERENCE", "/root/reference")


def _run(code, pythonpath):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join(pythonpath), PYTHONDONTWRITEBYTECODE="1", HF_HUB_OFFLINE="1")
    return subprocess.run([sys.executable, "-c", textwrap.dedent(code)], env=env, cwd="/tmp", capture_output=True, text=True, timeout=600)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "llava")), reason="needs the reference checkout (build container only)")
def test_reference_builder_resolves_the_overlay_and_reaches_from_pretrained(tmp_path):
    ckpt = TM.write_checkpoint(str(tmp_path / "llava_qwen_tiny"), TM.load())
    r = _run(f"""
        import llava, llava.model.builder as B, llava.model.llava_arch as A, llava.video_utils as V
        from llava.model.builder import load_pretrained_model
        from llava.model.language_model.llava_qwen import LlavaQwenConfig, LlavaQwenForCausalLM        # builder.py:220
        from llava.mm_utils import tokenizer_image_token, get_model_name_from_path, KeywordsStoppingCriteria   # model_scanqa.py:17
        from llava.video_utils import VideoProcessor, merge_video_dict                                     # model_scanqa.py:18
        from llava.constants import IMAGE_TOKEN_INDEX                                                      # falls through to the reference
        assert B.__file__.startswith({REF!r}), B.__file__                     # the reference's own loader ...
        assert A.__file__.startswith({OVERLAY!r}) and V.__file__.startswith({OVERLAY!r})   # ... over the overlay's modules
        assert B.LlavaQwenForCausalLM is LlavaQwenForCausalLM                 # `from llava.model import *` (builder.py:22)
        for name in ("LlavaMetaModel", "LlavaMetaForCausalLM"):
            assert hasattr(A, name)
        for name in ("encode_images", "prepare_inputs_labels_for_multimodal", "get_2dPool", "average_coordinate_in_patch",
                     "discrete_coords", "add_token_per_grid"):
            assert hasattr(A.LlavaMetaForCausalLM, name), name
        from transformers import AutoConfig
        cfg = AutoConfig.from_pretrained({ckpt!r})
        assert type(cfg) is LlavaQwenConfig and cfg.world_position_embedding_type == "avg-discrete-sin3d"
        try:
            load_pretrained_model({ckpt!r}, None, get_model_name_from_path({ckpt!r}), overwrite_config={{"use_cache": True}})
        except Exception as e:
            print("RAISED", type(e).__name__, e)
        else:
            print("LOADED")
        """, [OVERLAY, REF])
    assert r.returncode == 0, r.stderr[-3000:]
    # no GPU here: the call must have gone builder.py -> overlay from_pretrained -> "needs an MI355X" (no CPU fallback)
    assert "RAISED V3DError" in r.stdout and "MI355X" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_overlay_alone_exports_registers_and_refuses_unknown_arguments(tmp_path):
    ckpt = TM.write_checkpoint(str(tmp_path / "llava_qwen_tiny"), TM.load())
    r = _run(f"""
        from llava.model import LlavaQwenForCausalLM, LlavaQwenConfig
        import llava.model as M
        assert M.__all__ == ["LlavaQwenConfig", "LlavaQwenForCausalLM"]
        from transformers import AutoConfig, AutoModelForCausalLM
        cfg = AutoConfig.from_pretrained({ckpt!r})
        assert type(cfg) is LlavaQwenConfig
        assert AutoModelForCausalLM._model_mapping[LlavaQwenConfig] is LlavaQwenForCausalLM
        for kw in (dict(trust_remote_code=True), dict(load_in_8bit=True), dict(some_new_flag=1)):
            try:
                LlavaQwenForCausalLM.from_pretrained({ckpt!r}, **kw)
            except (TypeError, NotImplementedError) as e:
                print("REFUSED", sorted(kw), type(e).__name__)
        from v3d import loader
        sd = loader.read_weights({ckpt!r})
        ec = loader.engine_config(cfg, sd)
        print("CFG", ec.vit.hidden, ec.vit.layers, ec.vit.heads, ec.vit.inter, ec.llm.hidden, ec.llm.layers, ec.llm.heads, ec.llm.kv_heads,
              ec.llm.inter, ec.llm.vocab, ec.llm.max_pos)
        print("GROUND", ec.ground_head_type, ec.object_feature_type)
        cfg.ground_head_type, cfg.object_feature_type = "score", "patch27-pe"        # the variants no shipped script selects
        ec = loader.engine_config(cfg, sd)
        print("GROUND", ec.ground_head_type, ec.object_feature_type)
        """, [OVERLAY])
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout.count("REFUSED") == 3, r.stdout
    assert "CFG 144 2 2 272 256 2 2 1 384 320 2048" in r.stdout, r.stdout
    assert "GROUND infonce patch14-pe" in r.stdout and "GROUND score patch27-pe" in r.stdout, r.stdout
