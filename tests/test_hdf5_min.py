"""Keras HDF5 checkpoints without h5py (keras_nerf_amd/io/hdf5_min.py; reference nerf.py:63-64, 132-136).

Fixtures: tests/golden/keras_layout_small_{earliest,latest}.h5 were written by the REAL HDF5 library (libhdf5 1.10.x from
this image's /opt/conda, driven by oracle/hdf5_fixture/make_keras_h5.c) in the layout of Keras' save_weights for a
subclassed model of Dense layers -- once with default ("earliest") library bounds as h5py uses, once with "latest".
They are data: a 3 x 8 MLP with xyz_dim 9 / dir_dim 5 whose values come from a fixed LCG that this file regenerates.
When the HDF5 command-line tools are present the writer's output is also read back by libhdf5 itself (h5diff)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from keras_nerf_amd.io import hdf5_min as H

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = [f"layer_{i}" for i in range(3)] + ["sigma", "features", "rgb_features", "rgb"]
SHAPES = [(9, 8), (8, 8), (8, 8), (17, 1), (17, 8), (13, 4), (4, 3)]          # n_layers 3, units 8, skip 2, xyz 9, dir 5


def lcg_weights():
    seed, out, M = 12345, [], (1 << 64) - 1
    for fi, fo in SHAPES:
        for n, shape in ((fi * fo, (fi, fo)), (fo, (fo,))):
            v = np.empty(n, np.float32)
            for i in range(n):
                seed = (seed * 6364136223846793005 + 1442695040888963407) & M
                v[i] = np.float32(((seed >> 40) & 0xFFFF) / 65536.0) - np.float32(0.5)
            out.append(v.reshape(shape))
    return out


@pytest.mark.parametrize("fixture,superblock", [("keras_layout_small_earliest.h5", 0), ("keras_layout_small_latest.h5", 3)])
def test_reader_on_files_written_by_libhdf5(fixture, superblock):
    path = os.path.join(G, fixture)
    assert H.is_hdf5(path)
    r = H.Hdf5Reader(path)
    assert r.buf[8] == superblock
    assert sorted(r.root) == sorted(NAMES + ["top_level_model_weights"])
    ds = r.datasets()
    assert len(ds) == 14 and all(k.split("/")[-1] in ("kernel:0", "bias:0") for k in ds)
    model = "coarse_nerf" if superblock == 0 else "fine_nerf"
    assert f"layer_0/{model}/layer_0/kernel:0" in ds                 # Keras: /<layer>/<variable name incl. its name scope>
    got = H.read_keras_weights(path, NAMES)
    for a, b in zip(got, lcg_weights()):
        assert a.dtype == np.float32 and a.shape == b.shape
        np.testing.assert_array_equal(a, b)
    with pytest.raises(KeyError):
        H.read_keras_weights(path, NAMES + ["layer_9"])


def test_writer_roundtrip_and_structure(tmp_path):
    ws = lcg_weights()
    p = str(tmp_path / "coarse.h5")
    H.write_keras_weights(p, "coarse_nerf", NAMES, ws)
    assert H.is_hdf5(p)
    for a, b in zip(H.read_keras_weights(p, NAMES), ws):
        np.testing.assert_array_equal(a, b)
    r = H.Hdf5Reader(p)
    assert r.buf[8] == 0 and "top_level_model_weights" in r.root and r.root["top_level_model_weights"] == {}
    assert sorted(r.datasets()) == sorted(H.Hdf5Reader(os.path.join(G, "keras_layout_small_earliest.h5")).datasets())
    # zero-sized and 1-element arrays, deep nesting
    w = H.Hdf5Writer(); w.dataset("a/b/c/d/x", np.zeros((0, 3))); w.dataset("a/y", np.array([1.5])); w.write(str(tmp_path / "t.h5"))
    d = H.Hdf5Reader(str(tmp_path / "t.h5")).datasets()
    assert d["a/b/c/d/x"].shape == (0, 3) and d["a/y"].tolist() == [1.5]


def test_not_hdf5_and_unsupported_features_fail_clearly(tmp_path):
    p = tmp_path / "x.h5"; p.write_bytes(b"PK\x03\x04 not hdf5")
    assert not H.is_hdf5(str(p))
    with pytest.raises(H.Hdf5FormatError, match="not an HDF5 file"):
        H.Hdf5Reader(str(p))
    raw = bytearray(open(os.path.join(G, "keras_layout_small_earliest.h5"), "rb").read())
    raw[8] = 9                                                        # unknown superblock version
    q = tmp_path / "y.h5"; q.write_bytes(bytes(raw))
    with pytest.raises(H.Hdf5FormatError, match="superblock version 9"):
        H.Hdf5Reader(str(q))


def _tool(name):
    for d in ("/opt/conda/bin", "/usr/bin", "/usr/local/bin"):
        if os.path.exists(os.path.join(d, name)):
            return os.path.join(d, name)
    return shutil.which(name)


@pytest.mark.skipif(_tool("h5diff") is None or _tool("h5ls") is None, reason="HDF5 command-line tools not installed")
def test_libhdf5_reads_what_the_writer_wrote(tmp_path):
    """h5ls lists the Keras layout; h5diff (libhdf5's own comparison) finds the datasets and the string-array attributes of
    our file identical to the fixture the library wrote itself"""
    p = str(tmp_path / "ours.h5")
    H.write_keras_weights(p, "coarse_nerf", NAMES, lcg_weights())
    ls = subprocess.run([_tool("h5ls"), "-r", p], capture_output=True, text=True)
    assert ls.returncode == 0, ls.stderr
    assert "/rgb_features/coarse_nerf/rgb_features/kernel:0 Dataset {13, 4}" in ls.stdout
    assert "/top_level_model_weights Group" in ls.stdout
    ref = os.path.join(G, "keras_layout_small_earliest.h5")
    for obj in [f"/{n}/coarse_nerf/{n}/{w}" for n in NAMES for w in ("kernel:0", "bias:0")]:
        r = subprocess.run([_tool("h5diff"), ref, p, obj], capture_output=True, text=True)
        assert r.returncode == 0, (obj, r.stdout, r.stderr)
    r = subprocess.run([_tool("h5diff"), "-v", ref, p], capture_output=True, text=True)
    assert "attribute: <layer_names of </>> and <layer_names of </>>\n0 differences found" in r.stdout
    assert "attribute: <weight_names of </sigma>> and <weight_names of </sigma>>\n0 differences found" in r.stdout
    assert " differences found" in r.stdout and not [ln for ln in r.stdout.splitlines() if ln.endswith("differences found") and not ln.startswith("0 ")]


def test_nerf_mlp_checkpoints_are_keras_hdf5_and_old_npz_still_loads(tmp_path):
    from keras_nerf_amd.model.nerf.mlp import NeRFMLP
    m = NeRFMLP(8, 256, 4, name="coarse_nerf", seed=3); m.build()
    p = str(tmp_path / "coarse.h5")
    m.save_weights(p)
    assert H.is_hdf5(p)                                               # the reference's file name now holds the reference's format
    ds = H.Hdf5Reader(p).datasets()
    assert ds["layer_5/coarse_nerf/layer_5/kernel:0"].shape == (319, 256) and ds["rgb/coarse_nerf/rgb/bias:0"].shape == (3,)
    m2 = NeRFMLP(8, 256, 4, seed=4); m2.load_weights(p)
    np.testing.assert_array_equal(m.get_flat_weights(), m2.get_flat_weights())
    # a round-1 checkpoint: a NumPy archive under the name coarse.h5 -- told apart by its magic bytes
    old = str(tmp_path / "old" ); os.makedirs(old); old = os.path.join(old, "coarse.h5")
    m.save_weights(old, save_format="npz")
    assert not H.is_hdf5(old)
    m3 = NeRFMLP(8, 256, 4, seed=5); m3.load_weights(old)
    np.testing.assert_array_equal(m.get_flat_weights(), m3.get_flat_weights())
    # explicit .npz stays available under an honest name
    m.save_weights(str(tmp_path / "coarse.npz")); assert not H.is_hdf5(str(tmp_path / "coarse.npz"))
    # a Keras file of another architecture is refused with the offending layer named
    small = NeRFMLP(3, 8, 2, xyz_dim=9, dir_dim=5, seed=1)
    small.load_weights(os.path.join(G, "keras_layout_small_earliest.h5"))
    for a, b in zip(small.get_weights(), lcg_weights()):
        np.testing.assert_array_equal(a, b)
    with pytest.raises((ValueError, KeyError)):
        m2.load_weights(os.path.join(G, "keras_layout_small_earliest.h5"))
    bad = tmp_path / "junk.h5"; bad.write_bytes(b"\x00" * 64)
    with pytest.raises(ValueError, match="neither an HDF5 file"):
        m2.load_weights(str(bad))


def test_reader_refuses_damaged_files_with_a_format_error(tmp_path):
    """the reader takes untrusted files: a group that contains itself, truncation and random byte damage must end in
    Hdf5FormatError (or load, if the damage missed everything that matters) -- never a hang, a RecursionError or a bare IndexError"""
    src = open(os.path.join(G, "keras_layout_small_earliest.h5"), "rb").read()
    # (1) a cycle: point the first symbol-table entry of the root group back at the root's own object header
    r = H.Hdf5Reader(os.path.join(G, "keras_layout_small_earliest.h5"))
    root = r.root_addr
    snod = src.index(b"SNOD")
    cyc = bytearray(src)
    cyc[snod + 8 + 8:snod + 8 + 16] = int(root).to_bytes(8, "little")
    p = tmp_path / "cycle.h5"; p.write_bytes(bytes(cyc))
    with pytest.raises(H.Hdf5FormatError, match="cyclic"):
        H.Hdf5Reader(str(p))
    # (2) truncations
    refused = 0
    for cut in (9, 100, 600, len(src) // 3, len(src) // 2, len(src) - 7):
        p = tmp_path / f"cut{cut}.h5"; p.write_bytes(src[:cut])
        try:
            H.Hdf5Reader(str(p)).datasets()           # a cut behind the last object may still load
        except H.Hdf5FormatError:
            refused += 1
    assert refused >= 3
    # (3) random damage in both fixtures (superblock 0 / symbol tables and superblock 2 / compact groups)
    rng = np.random.default_rng(0)
    outcomes = {"ok": 0, "refused": 0}
    for name in ("keras_layout_small_earliest.h5", "keras_layout_small_latest.h5"):
        data = open(os.path.join(G, name), "rb").read()
        for trial in range(150):
            b = bytearray(data)
            for pos in rng.integers(8, len(b), size=int(rng.integers(1, 6))):
                b[pos] = int(rng.integers(0, 256))
            p = tmp_path / "fuzz.h5"; p.write_bytes(bytes(b))
            try:
                H.Hdf5Reader(str(p)).datasets()
                outcomes["ok"] += 1
            except H.Hdf5FormatError:
                outcomes["refused"] += 1
    assert outcomes["ok"] + outcomes["refused"] == 300 and outcomes["refused"] > 0
