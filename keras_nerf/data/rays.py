"""alias of keras_nerf_amd.data.rays (reference keras_nerf/data/rays.py)"""
from keras_nerf_amd.data.rays import RaysGenerator  # noqa: F401
