"""alias of keras_nerf_amd.model.nerf.mlp (reference keras_nerf/model/nerf/mlp.py)"""
from keras_nerf_amd.model.nerf.mlp import NeRFMLP, layer_shapes  # noqa: F401
