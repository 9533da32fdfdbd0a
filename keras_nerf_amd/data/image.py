"""ImageLoader -- counterpart of reference keras_nerf/data/image.py:4-35 (host side, no tf.image).

PNG -> float32 RGBA in [0,1], antialiased bilinear resize IN FLOAT32, RGB composited over a white or black background with the
image's alpha, alpha kept as 4th channel, clip.  Note the reference passes (image_width, image_height) as the resize SIZE
(image.py:22-23, i.e. height := image_width) -- identical for the square images it is used with; kept here.

The resize follows what `tf.image.resize(image, size, antialias=True)` (method bilinear, image.py:22-23) computes -- TensorFlow's
scale-and-translate resampler with the triangle kernel: for output index x the sample point (x + 0.5) * in / out, a kernel widened by
max(in / out, 1), taps on the input pixel centres within its radius, weights normalised per output pixel, rows first and then
columns (the order of its span gather as recalled -- the two orders differ by float32 rounding only, ~1e-7), all in float32 on the UNQUANTISED values (rounds 1-4 resized the 8-bit image with PIL: the same kernel, but each output
rounded to 1/255 -- up to 0.002 per target value).  TensorFlow itself is not available here to compare against
(DESIGN.md section 3), so tests/test_host_logic.py pins the algorithm by known answers: identity at equal size, the
(1, 3, 3, 1) / 8 taps of a 2:1 reduction, partition of unity at 800 -> 128, agreement with PIL's reducing BILINEAR to its rounding."""
from __future__ import annotations

import numpy as np
from PIL import Image


def triangle_resize_weights(n_in: int, n_out: int) -> np.ndarray:
    """[n_out, n_in] float32: row x holds the normalised triangle-kernel taps of output pixel x (antialiased when n_out < n_in)"""
    scale = np.float32(n_out) / np.float32(n_in)
    inv_scale = np.float32(1.0) / scale
    kernel_scale = max(float(inv_scale), 1.0)              # antialias=True: the kernel follows the reduction factor
    radius = 1.0 * kernel_scale                            # the triangle kernel's radius is 1
    W = np.zeros((n_out, n_in), np.float32)
    for x in range(n_out):
        sample = (x + 0.5) * float(inv_scale)
        lo = max(int(np.ceil(sample - radius - 0.5)), 0)
        hi = min(int(np.floor(sample + radius - 0.5)), n_in - 1)
        src = np.arange(lo, hi + 1)
        w = np.maximum(0.0, 1.0 - np.abs((src + 0.5 - sample) / kernel_scale)).astype(np.float32)
        total = w.sum(dtype=np.float32)
        if total > 0:
            w = w / total
        W[x, lo:hi + 1] = w
    return W


def resize_antialiased(img: np.ndarray, rows: int, cols: int) -> np.ndarray:
    """img [H, W, C] float32 -> [rows, cols, C] float32"""
    img = np.asarray(img, np.float32)
    if img.shape[0] == rows and img.shape[1] == cols:
        return img
    Wc = triangle_resize_weights(img.shape[1], cols)
    Wr = triangle_resize_weights(img.shape[0], rows)
    H, Wd, Cn = img.shape
    # rows first, then columns: two dense float32 products (the weight rows hold at most 2 * ceil(in / out) + 1 non-zero taps; the
    # zeros add exactly nothing; ~20 ms for 800 x 800 -> 128 x 128, once per image and epoch-cached by the loader); NumPy only
    tmp = Wr @ np.ascontiguousarray(img).reshape(H, Wd * Cn)                                               # [rows, W * C]
    tmp = np.ascontiguousarray(tmp.reshape(rows, Wd, Cn).transpose(1, 0, 2)).reshape(Wd, rows * Cn)        # [W, rows * C]
    out = (Wc @ tmp).reshape(cols, rows, Cn).transpose(1, 0, 2)                                            # [rows, cols, C]
    return np.ascontiguousarray(out, dtype=np.float32)


class ImageLoader:
    def __init__(self, image_width: int, image_height: int, white_background: bool = False, **kwargs):
        self.image_width, self.image_height, self.white_background = image_width, image_height, white_background

    def __call__(self, image_path) -> np.ndarray:
        img = Image.open(image_path).convert("RGBA")
        a = np.asarray(img, dtype=np.float32) / np.float32(255.0)                  # convert_image_dtype(uint8 -> float32)
        rows, cols = self.image_width, self.image_height          # tf.image.resize(image, (image_width, image_height))
        a = resize_antialiased(a, rows, cols)
        alpha = a[..., 3:4]
        bg = np.ones_like(a[..., :3]) if self.white_background else np.zeros_like(a[..., :3])
        rgb = alpha * a[..., :3] + (np.float32(1.0) - alpha) * bg
        return np.clip(np.concatenate([rgb, alpha], axis=-1), 0.0, 1.0).astype(np.float32)
