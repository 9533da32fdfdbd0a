"""CPU check of the device layouts: the real library's packing tables (knerf_debug_table) are replayed through a
lane-level NumPy model of the MFMA instruction and must reproduce the oracle MLP (bf16-emulating mode)."""
import numpy as np
import pytest

from keras_nerf_amd import _lib
from keras_nerf_amd import debug as D
from oracle import nerf_oracle as O
from tests import mfma_sim as M


@pytest.fixture(scope="module")
def setup():
    cfg = O.NerfConfig()
    rng = np.random.default_rng(3)
    params = O.init_params(cfg, 11)
    for b in params[1::2]:
        b += rng.normal(0, 0.05, b.shape).astype(np.float32)
    flat = M.extended_weights(params, cfg)       # parameters + the composed head (csrc/layout.h)
    o = rng.normal(0, 1.5, (32, 3)).astype(np.float32)
    d = rng.normal(0, 1, (32, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    t = rng.uniform(2, 6, (32, 1)).astype(np.float32)
    p = (o + d * t).astype(np.float32)
    return cfg, params, flat, p, d


def test_tables_cover_every_parameter_exactly_where_expected():
    fwd, bias = D.debug_table(0), D.debug_table(1)
    n = _lib.load().knerf_param_count()
    used = np.zeros(n, bool)
    used[fwd[(fwd >= 0) & (fwd < n)]] = True
    used[bias[(bias >= 0) & (bias < n)]] = True
    # the trunk (layer_0..7) is streamed tensor by tensor; the four tensors behind it (sigma, features, rgb_features, rgb)
    # reach the kernels through the composed head matrix [283 real rows][4] + bias [4] stored behind the parameters
    n_trunk = 63 * 256 + 256 + 4 * (256 * 256 + 256) + (319 * 256 + 256) + 2 * (256 * 256 + 256)
    assert used[:n_trunk].all() and not used[n_trunk:].any()
    head = np.concatenate([fwd[fwd >= n], bias[bias >= n]]) - n
    assert sorted(head) == sorted([r * 4 + c for r in range(283) for c in range(4)] + [288 * 4 + c for c in range(4)])
    vals, counts = np.unique(fwd[fwd >= 0], return_counts=True)
    assert counts.max() == 1                            # and appears once
    assert fwd.size == 978 * 512 and bias.size == 65 * 32
    bwd = D.debug_table(2)
    assert bwd.size == 904 * 512
    vals, counts = np.unique(bwd[bwd >= 0], return_counts=True)
    assert counts.max() == 1
    assert sorted(bwd[bwd >= n] - n) == [r * 4 + c for r in range(256) for c in range(4)]   # dh7 = H[:256] dz_head


def test_forward_chain_matches_oracle(setup):
    cfg, params, flat, p, d = setup
    rgb, sigma, _ = M.forward_chain(D.debug_table(0), D.debug_table(1), flat, p, d)
    xyz = O.positional_encoding(p, 10)[None]
    dire = O.positional_encoding(d, 4)[None]
    rgb_o, sigma_o = O.mlp_forward(params, xyz, dire, cfg, emulate_bf16=O.FUSED)
    np.testing.assert_allclose(rgb, rgb_o[0], atol=1e-3)   # a bf16 rounding flip of one activation moves an output by ~1e-4
    np.testing.assert_allclose(sigma, sigma_o[0, :, 0], atol=1e-3, rtol=1e-2)
    rgb32, sigma32 = O.mlp_forward(params, xyz, dire, cfg)
    assert np.abs(rgb - rgb32[0]).max() < 2e-2          # bf16 operands vs fp32: the kernel's stated tolerance class


def test_backward_chain_and_wgrad_match_oracle(setup):
    """dgrad stream table, saved-block swizzle, transposed-read addressing and the wgrad destination tables, replayed
    on the CPU for two sample tiles, reproduce the oracle's 24 gradient tensors (bf16-emulating mode)."""
    cfg, params, flat, p, d = setup
    fwd, bias, bwd = D.debug_table(0), D.debug_table(1), D.debug_table(2)
    dst_tab, job_off = D.debug_table(3), D.debug_table(4)
    rng = np.random.default_rng(8)
    acts, dzs, xs, ds, drgbs, dsigs, m7 = [], [], [], [], [], [], []
    for tile in range(2):
        pt = (p + rng.normal(0, 0.3, p.shape)).astype(np.float32)
        rgb, sigma, saved = M.forward_chain(fwd, bias, flat, pt, d)
        drgb = rng.normal(0, 1, (32, 3)).astype(np.float32)
        dsig = rng.normal(0, 1, (32,)).astype(np.float32)
        acts.append(M.act_run(saved)); m7.append(M.mask_block_words(saved["masks"][7]))
        dzs.append(M.backward_chain(bwd, flat, rgb, sigma, drgb, dsig, saved["masks"]))
        xs.append(pt); ds.append(d); drgbs.append(drgb); dsigs.append(dsig)
    n_par = O.param_count(cfg)
    grad = M.wgrad(acts, dzs, dst_tab, job_off, n_par, flat, fwd, bias, bwd, m7)
    xyz = O.positional_encoding(np.concatenate(xs), 10)[None]
    dire = O.positional_encoding(np.concatenate(ds), 4)[None]
    _, _, cache = O.mlp_forward(params, xyz, dire, cfg, emulate_bf16=O.FUSED, want_cache=True)
    g_o = O.flatten_params(O.mlp_backward(params, cache, np.concatenate(drgbs), np.concatenate(dsigs), cfg))
    scale = np.abs(g_o).max()
    err = np.abs(grad - g_o).max() / scale
    assert err < 2e-2, err
    # per tensor, so that a mis-routed small tensor (biases, sigma, rgb heads) cannot hide behind a large one
    off = 0
    for name, fi, fo in O.layer_shapes(cfg):
        for n in (fi * fo, fo):
            a, b = grad[off:off + n], g_o[off:off + n]
            assert np.abs(a - b).max() <= 3e-2 * max(np.abs(b).max(), 1e-6), (name, n)
            off += n
