"""alias of keras_nerf_amd.model.nerf.utils (reference keras_nerf/model/nerf/utils.py)"""
from keras_nerf_amd.model.nerf.utils import NeRFUtils  # noqa: F401
