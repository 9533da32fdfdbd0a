"""alias of keras_nerf_amd.model.nerf.callback (reference keras_nerf/model/nerf/callback.py)"""
from keras_nerf_amd.model.nerf.callback import NeRFTrainMonitor  # noqa: F401
