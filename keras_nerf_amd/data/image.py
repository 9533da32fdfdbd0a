"""ImageLoader -- counterpart of reference keras_nerf/data/image.py:4-35 (host side, PIL instead of tf.image).

PNG -> float32 RGBA in [0,1], antialiased resize, RGB composited over a white or black background with the image's
alpha, alpha kept as 4th channel, clip.  Note the reference passes (image_width, image_height) as the resize SIZE
(image.py:22-23, i.e. height := image_width) -- identical for the square images it is used with; kept here.
The resampling filter cannot match TensorFlow's antialiased bilinear kernel bit for bit (PIL's reducing BILINEAR is the
closest counterpart)."""
from __future__ import annotations

import numpy as np
from PIL import Image


class ImageLoader:
    def __init__(self, image_width: int, image_height: int, white_background: bool = False, **kwargs):
        self.image_width, self.image_height, self.white_background = image_width, image_height, white_background

    def __call__(self, image_path) -> np.ndarray:
        img = Image.open(image_path).convert("RGBA")
        rows, cols = self.image_width, self.image_height          # tf.image.resize(image, (image_width, image_height))
        if img.size != (cols, rows):
            img = img.resize((cols, rows), resample=Image.BILINEAR, reducing_gap=None)
        a = np.asarray(img, dtype=np.float32) / np.float32(255.0)
        alpha = a[..., 3:4]
        bg = np.ones_like(a[..., :3]) if self.white_background else np.zeros_like(a[..., :3])
        rgb = alpha * a[..., :3] + (np.float32(1.0) - alpha) * bg
        return np.clip(np.concatenate([rgb, alpha], axis=-1), 0.0, 1.0).astype(np.float32)
