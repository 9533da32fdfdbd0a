"""alias of keras_nerf_amd.model.nerf.nerf (reference keras_nerf/model/nerf/nerf.py)"""
from keras_nerf_amd.model.nerf.nerf import *  # noqa: F401,F403
from keras_nerf_amd.model.nerf.nerf import NeRF, NonFiniteGradientError  # noqa: F401
