"""alias of keras_nerf_amd.data.image (reference keras_nerf/data/image.py)"""
from keras_nerf_amd.data.image import ImageLoader  # noqa: F401
