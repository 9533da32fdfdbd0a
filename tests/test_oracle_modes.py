"""An independent check of the oracle's `emulate_bf16=FUSED` arithmetic.

Every tight GPU tolerance is taken against FUSED -- the collapsed 283x4 head the kernels evaluate (oracle/nerf_oracle.py head_compose,
DESIGN.md 2.0) -- so an error that FUSED and the kernels SHARED would be invisible to those tests; tests/test_oracle_grad.py proves the
collapse in fp64 only.  Here the network AS WRITTEN (mlp.py:29-50: twelve Dense layers, one after the other) is evaluated in the same
precision (`emulate_bf16=True`: bf16 operands, fp32 accumulate, layer by layer) and compared with FUSED and with the GPU, on 4,096
rows (64 rays x 64 samples) of the default shape and of a width the kernels run zero-padded (96 -> 128): outputs and every one of the
24 gradient tensors.  The gap between the two bf16 evaluations is the rounding of three small intermediate tensors (features,
rgb_features, the composed matrix): ~2e-3 of a tensor's max -- an order of magnitude below the bf16-vs-fp32 gap (6e-2 here), so a
wrong index, a missing bias term or a transposed product in the composed head (each O(1)) cannot hide in it."""
import numpy as np
import pytest

from oracle import nerf_oracle as O
from tests.problem import make_problem

# measured on the seeded problems below (weight scale 1.0): 2.2e-3 (8 x 256), 4.2e-3 (8 x 96); asserted with a factor ~3
GAP_TRUE_VS_FUSED = {256: 8e-3, 96: 1.2e-2}


def _problem(units):
    cfg = O.NerfConfig(dense_units=units)
    P = make_problem(n_images=1, wh=8, weight_scale=1.0, bias_std=0.05, cfg=cfg)
    N = P["N"]
    return P, cfg, (P["o"].reshape(N, 3), P["d"].reshape(N, 3), P["t"].reshape(N, -1), P["img"].reshape(N, 3), P["u"].reshape(N, -1))


def _worst(ga, gb):
    errs = [float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)) for a, b in zip(ga, gb)]
    return max(errs), int(np.argmax(errs))


@pytest.mark.parametrize("units", [256, 96])
def test_fused_head_arithmetic_against_the_twelve_layers_as_written_in_the_same_precision(units):
    P, cfg, (o, d, t, img, _) = _problem(units)
    assert o.shape[0] * cfg.n_coarse == 4096
    res = {emu: O.chunk_loss_and_grads(P["cp"], o, d, t, img, cfg, True, emulate_bf16=emu) for emu in (False, True, O.FUSED)}
    gap, where = _worst(res[True][2], res[O.FUSED][2])
    gap32, _ = _worst(res[True][2], res[False][2])
    print(f"width {units}: layer-by-layer bf16 vs fused head {gap:.2e} (tensor {where}); vs fp32 {gap32:.2e}")
    assert 0 < gap < GAP_TRUE_VS_FUSED[units], (gap, where)
    assert gap < 0.25 * gap32                                        # far inside the precision gap: a logic error would not be
    for k in ("image", "depth", "weights"):
        assert np.abs(res[True][0][k] - res[O.FUSED][0][k]).max() < 2e-4, k
    assert abs(float(res[True][1]) - float(res[O.FUSED][1])) < 2e-5
    # every head tensor on its own (they are what the collapse touches): sigma, features, rgb_features, rgb kernels and biases
    n = cfg.n_layers
    for j in range(2 * n, 2 * n + 8):
        a, b = res[True][2][j], res[O.FUSED][2][j]
        assert a.shape == b.shape and np.abs(a - b).max() <= GAP_TRUE_VS_FUSED[units] * np.abs(b).max(), j


@pytest.mark.gpu
@pytest.mark.parametrize("units", [256, 96])
def test_gpu_against_both_bf16_evaluations(units):
    """the kernels against the network as written (mode True) as well as against their own arithmetic (FUSED): coarse and fine pass,
    every gradient tensor; width 96 runs zero-padded at 128 on the fused kernels (runtime.py _set_up_padding)"""
    import torch
    from keras_nerf_amd.debug import debug_buffer
    from keras_nerf_amd.runtime import KnerfContext
    from tests.test_gpu_forward import log_stats
    P, cfg, (o, d, t, img, u) = _problem(units)
    ctx = KnerfContext(dense_units=units, white_background=True)
    assert not ctx.get_option("general_shape_path")
    ctx.set_weights(0, O.flatten_params(P["cp"])); ctx.set_weights(1, O.flatten_params(P["fp"]))
    loss = torch.zeros(2, device="cuda")
    ctx.train_chunk(o, d, t, img, u, inv_chunks=1.0, loss=loss)
    torch.cuda.synchronize()
    g = np.concatenate([ctx.grads(0).cpu().numpy(), ctx.grads(1).cpu().numpy()])       # the REAL layout (un-padded where the width is padded)
    n = O.param_count(cfg)
    assert g.size == 2 * n
    N = P["N"]
    t_fine = debug_buffer(ctx, 5).view(torch.float32).cpu().numpy()[:N * 192].reshape(N, 192)
    for emu, tol in ((O.FUSED, 1.5e-2), (True, 2e-2)):
        _, lc, gc = O.chunk_loss_and_grads(P["cp"], o, d, t, img, cfg, True, emulate_bf16=emu)
        _, lf, gf = O.chunk_loss_and_grads(P["fp"], o, d, t_fine, img, cfg, True, emulate_bf16=emu)
        ec, _ = _worst(O.unflatten_params(g[:n], cfg), gc)
        ef, _ = _worst(O.unflatten_params(g[n:], cfg), gf)
        log_stats(f"gpu_vs_oracle_mode_{emu}_w{units}", coarse_worst=ec, fine_worst=ef)
        assert ec < tol and ef < tol, (emu, ec, ef)
        assert abs(float(loss[0]) - float(lc)) < 2e-3 and abs(float(loss[1]) - float(lf)) < 2e-3
    ctx.close()
