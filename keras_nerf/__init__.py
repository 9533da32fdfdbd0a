"""Import-name alias: `keras_nerf.*` resolves to the MI355X implementation in `keras_nerf_amd.*`, so code written against the
reference package (`from keras_nerf.model.nerf.nerf import NeRF`, reference train_single.py:8-12) runs on this path without
an import swap.  Nothing lives here but re-exports; see INTEGRATION.md."""
