"""Drop-in overlay of the reference's `llava` package for the position-aware video->LLM hot path.

Put this directory's parent AHEAD of the reference checkout on PYTHONPATH: the modules defined here
(video_utils, utils_3d, mm_utils, model.position_encoding, model.llava_arch,
model.language_model.llava_qwen) shadow the reference's, every other `llava.*` import falls through
to the reference (pkgutil.extend_path).  See INTEGRATION.md.
"""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
