"""MI355X mirror of the image-processor half of llava/model/multimodal_encoder/siglip_encoder.py (a7).

`SigLipImageProcessor.preprocess` (reference :47-67) keeps its name, constructor arguments and return contract
(`{"pixel_values": [F,3,384,384] float32}`); the rescale / normalize / HWC->CHW arithmetic runs in
v3d_preprocess_rgb_u8 on the device.  Frames that are not yet 384 x 384 are first resized by the same PIL call the
reference's transform chain makes (host I/O-side plumbing, as decoding the JPEG is); VideoProcessor already hands over
384 x 384 crops (video_utils.py:292-308), so that branch is idle on the eval path.  The vision tower itself is
v3d.engine.Engine.encode_images.
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
        frames = torch.from_numpy(np.stack([self._frame(im) for im in images])).to(device)
        pixel_values = ops.preprocess_rgb(frames, torch.float32, self.image_mean, self.image_std, self.rescale_factor)
        return {"pixel_values": pixel_values if return_tensors == "pt" else pixel_values.cpu().numpy()}
