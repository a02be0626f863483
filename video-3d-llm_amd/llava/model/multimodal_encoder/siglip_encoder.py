"""MI355X mirror of the image-processor half of llava/model/multimodal_encoder/siglip_encoder.py (a7).

`SigLipImageProcessor.preprocess` (reference :47-67) keeps its name, constructor arguments and return contract
(`{"pixel_values": [F,3,384,384] float32}`); the rescale / normalize / HWC->CHW arithmetic runs in
v3d_preprocess_rgb_u8 on the device.  Frames that are not yet 384 x 384 are first resized by the same PIL call the
reference's transform chain makes (host I/O-side plumbing, as decoding the JPEG is); VideoProcessor already hands over
384 x 384 crops (video_utils.py:292-308), so that branch is idle on the eval path.  `SigLipVisionTower` keeps the
reference's constructor and the attributes its callers read (builder.py:277-283: is_loaded / load_model / to / image_processor;
llava_arch.py: num_patches_per_side, hidden_size, forward); the encoder itself runs inside v3d.engine.Engine (patch-embed GEMM,
26 x [LayerNorm, fused-QKV GEMM, non-causal attention, out-proj, LayerNorm, fc1 + gelu_tanh, fc2]).
"""
import numpy as np
import torch
from PIL import Image

from v3d import ops
from v3d._native import V3DError


class SigLipImageProcessor:
    def __init__(self, image_mean=(0.5, 0.5, 0.5), image_std=(0.5, 0.5, 0.5), size=(384, 384), crop_size=None,
                 resample=Image.BICUBIC, rescale_factor=1 / 255, data_format="channels_first"):
        self.image_mean, self.image_std = tuple(image_mean), tuple(image_std)
        self.size, self.resample, self.rescale_factor = tuple(size), resample, rescale_factor
        self.data_format = data_format
        self.crop_size = crop_size if crop_size is not None else {"height": 384, "width": 384}

    def _frame(self, image):
        if isinstance(image, torch.Tensor):
            image = image.cpu().numpy()
        if isinstance(image, np.ndarray):
            if image.dtype != np.uint8 or image.ndim != 3 or image.shape[-1] != 3:
                raise V3DError("SigLipImageProcessor wants RGB uint8 frames [H,W,3] or PIL images")
            if image.shape[:2] == (self.size[0], self.size[1]):
                return image
            image = Image.fromarray(image)
        image = image.convert("RGB")
        if (image.height, image.width) != (self.size[0], self.size[1]):
            image = image.resize((self.size[1], self.size[0]), resample=self.resample)
        return np.asarray(image)

    def preprocess(self, images, return_tensors="pt", device="cuda"):
        if isinstance(images, Image.Image):
            images = [images]
        if isinstance(images, torch.Tensor) and images.dim() == 4 and images.dtype == torch.uint8 and \
                tuple(images.shape[1:3]) == (self.size[0], self.size[1]):
            frames = images.to(device)          # VideoProcessor.preprocess's crops, already on the device at the tower's size
        else:
            frames = torch.from_numpy(np.stack([self._frame(im) for im in images])).to(device)
        pixel_values = ops.preprocess_rgb(frames, torch.float32, self.image_mean, self.image_std, self.rescale_factor)
        return {"pixel_values": pixel_values if return_tensors == "pt" else pixel_values.cpu().numpy()}


class SigLipVisionConfig:
    """siglip_encoder.py:70-113 defaults (google/siglip-so400m-patch14-384); the widths actually used come from the
    checkpoint's tensors (v3d.loader.engine_config)."""
    model_type = "siglip_vision_model"

    def __init__(self, hidden_size=1152, image_mean=(0.5, 0.5, 0.5), intermediate_size=4304, num_hidden_layers=27,
                 num_attention_heads=16, num_channels=3, image_size=384, patch_size=14, hidden_act="gelu_pytorch_tanh",
                 layer_norm_eps=1e-6, attention_dropout=0.0, **kwargs):
        self.hidden_size, self.intermediate_size, self.num_hidden_layers = hidden_size, intermediate_size, num_hidden_layers
        self.num_attention_heads, self.num_channels, self.patch_size, self.image_size = num_attention_heads, num_channels, patch_size, image_size
        self.attention_dropout, self.layer_norm_eps, self.hidden_act, self.image_mean = attention_dropout, layer_norm_eps, hidden_act, image_mean


class SigLipVisionTower:
    """Handle with the reference class's surface (siglip_encoder.py:538-620).  The weights live in the Engine the owning
    LlavaQwenForCausalLM built from the checkpoint; `bind(engine)` attaches it.  forward(images [F,3,384,384]) returns
    hidden_states[-1] of the truncated encoder, [F, 729, 1152]."""

    def __init__(self, vision_tower, vision_tower_cfg=None, delay_load=False):
        self.vision_tower_name = vision_tower
        self.config = SigLipVisionConfig()
        self.image_processor = SigLipImageProcessor()
        self.is_loaded = False
        self._engine = None

    def bind(self, engine):
        v = engine.cfg.vit
        self._engine = engine
        self.config = SigLipVisionConfig(hidden_size=v.hidden, intermediate_size=v.inter, num_hidden_layers=v.layers + 1,
                                         num_attention_heads=v.heads, image_size=v.image, patch_size=v.patch)
        self.is_loaded = True
        return self

    def load_model(self, device_map=None):
        if not self.is_loaded:
            raise V3DError("SigLipVisionTower: the tower is loaded with the model's checkpoint (LlavaQwenForCausalLM.from_pretrained); "
                           "a stand-alone load is not on the accelerated path")

    def to(self, *args, **kwargs):        # builder.py:281: the engine already holds the weights on its device / dtype
        return self

    def requires_grad_(self, flag=False):
        if flag:
            raise NotImplementedError("the accelerated tower is inference-only")
        return self

    def forward(self, images):
        e = self._engine
        if e is None:
            raise V3DError("SigLipVisionTower is not bound to an engine")
        if isinstance(images, (list, tuple)):
            images = torch.stack([im for im in images])
        e.encode_images(images.to(e.device))
        return e.vit_hidden(images.shape[0]).to(images.dtype)

    __call__ = forward

    @property
    def dtype(self):
        return self._engine.dtype

    @property
    def device(self):
        return torch.device(self._engine.device)

    @property
    def hidden_size(self):
        return self.config.hidden_size

    @property
    def num_patches(self):
        return (self.config.image_size // self.config.patch_size) ** 2

    @property
    def num_patches_per_side(self):
        return self.config.image_size // self.config.patch_size

    @property
    def image_size(self):
        return self.config.image_size
