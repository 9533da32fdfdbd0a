"""alias of keras_nerf_amd.data.loader (reference keras_nerf/data/loader.py)"""
from keras_nerf_amd.data.loader import DatasetLoader, RayImageDataset  # noqa: F401
