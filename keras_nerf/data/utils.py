"""alias of keras_nerf_amd.data.utils (reference keras_nerf/data/utils.py)"""
from keras_nerf_amd.data.utils import *  # noqa: F401,F403
