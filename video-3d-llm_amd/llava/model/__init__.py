"""`llava.model` under the overlay: the accelerated classes are exported here, as the reference's package does for its own
(llava/model/__init__.py:1-16: `from llava.model import *` in builder.py:22 and train_3d.py:46 must find them); every other
`llava.model.*` module (builder, the multimodal_projector / multimodal_resampler packages, ...) falls through to the reference
checkout via extend_path.  Only the Qwen2 backbone exists here - the one the 3-D scripts train and evaluate
(scripts/3d/train/train_multi.sh:17,25); the Llama / Mistral / Mixtral / Gemma wrappers are not served by the overlay."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)

from .language_model.llava_qwen import LlavaQwenConfig, LlavaQwenForCausalLM  # noqa: E402,F401

__all__ = ["LlavaQwenConfig", "LlavaQwenForCausalLM"]
