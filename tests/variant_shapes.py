"""The extra trunk shapes of the `xshape` build variant: the list lives with the build script (keras_nerf_amd/build.py XSHAPES), so
that product build code does not import the test tree; tests import it from here."""
from keras_nerf_amd.build import XSHAPES  # noqa: F401
