"""dense_units between the fused widths (50, 96, 192, 200, ...): the host shim runs the network on the FUSED kernels of the next of
64 / 128 / 256 with zero-padded weights (keras_nerf_amd/runtime.py `_set_up_padding`, round 5).  CPU: the index map and the
exactness claim on the oracle (the padded network IS the real one: same outputs, same gradients at the real parameters, exactly zero
gradients everywhere else).  GPU: the padded context against the oracle of the REAL shape and against the general-shape kernels,
padding still zero after Adam steps, the real layout at set_weights / get_weights / grads / checkpoints."""
import numpy as np
import pytest

from oracle import nerf_oracle as O
from tests.problem import make_problem

SHAPES = [(8, 4, 192), (8, 4, 96), (4, 2, 50), (6, 3, 200), (4, 2, 51),      # (4, 2, 51): odd (rgb_features has 51 // 2 = 25 outputs)
          (3, 2, 288)]                                                       # above 256: the general-shape kernels at the next multiple of 128


def _pad_params(params, cfg, wide):
    from keras_nerf_amd.runtime import width_pad_index
    cfg_p = O.NerfConfig(n_layers=cfg.n_layers, dense_units=wide, skip_layer=cfg.skip_layer)
    idx = width_pad_index(cfg.n_layers, cfg.dense_units, wide, cfg.skip_layer, cfg.xyz_dim, cfg.dir_dim)
    flat = np.zeros(O.param_count(cfg_p), np.float32)
    flat[idx] = O.flatten_params(params)
    return cfg_p, idx, flat


@pytest.mark.parametrize("nl,sk,units", SHAPES)
def test_zero_padded_network_is_the_real_one_on_the_oracle(nl, sk, units):
    from keras_nerf_amd.runtime import padded_width
    cfg = O.NerfConfig(n_layers=nl, dense_units=units, skip_layer=sk)
    wide = padded_width(units)
    assert wide in (64, 128, 256, 384) and wide > units
    P = make_problem(n_images=1, wh=6, weight_scale=1.5, bias_std=0.05, cfg=cfg)
    N = P["N"]
    o, d, t, img = P["o"].reshape(N, 3), P["d"].reshape(N, 3), P["t"].reshape(N, -1), P["img"].reshape(N, 3)
    cfg_p, idx, flat_p = _pad_params(P["cp"], cfg, wide)
    assert len(set(idx.tolist())) == idx.size == O.param_count(cfg)
    for kernel_arith in (False, O.FUSED):                     # fp32 and the kernels' bf16 arithmetic
        res, loss, grads = O.chunk_loss_and_grads(P["cp"], o, d, t, img, cfg, True, emulate_bf16=kernel_arith)
        res_p, loss_p, grads_p = O.chunk_loss_and_grads(O.unflatten_params(flat_p, cfg_p), o, d, t, img, cfg_p, True, emulate_bf16=kernel_arith)
        gp = O.flatten_params(grads_p)
        # exact zeros only ever join the sums: the padded network's outputs and real-parameter gradients are the real network's up
        # to the order BLAS adds the (now longer) dot products in -- and bit-identical where that order is fixed
        np.testing.assert_allclose(res_p["image"], res["image"], rtol=0, atol=1e-6)
        assert abs(float(loss_p) - float(loss)) < 1e-7
        np.testing.assert_allclose(gp[idx], O.flatten_params(grads), rtol=0, atol=2e-7 * max(1.0, float(np.abs(O.flatten_params(grads)).max())))
        rest = np.ones(gp.size, bool); rest[idx] = False
        assert rest.sum() == O.param_count(cfg_p) - O.param_count(cfg) and (gp[rest] == 0).all()      # padded parameters: EXACTLY zero gradient


def test_widths_that_are_not_padded():
    from keras_nerf_amd.runtime import padded_width
    assert [padded_width(u) for u in (64, 128, 256, 384, 512, 1024, 1)] == [None] * 7         # a fused width / a multiple of 128 / no rgb_features outputs
    assert [padded_width(u) for u in (2, 50, 63, 66, 130, 255)] == [64, 64, 64, 128, 256, 256]
    assert [padded_width(u) for u in (257, 288, 352, 385, 416, 520)] == [384, 384, 384, 512, 512, 640]


@pytest.mark.gpu
@pytest.mark.parametrize("nl,sk,units", SHAPES[:3] + SHAPES[4:])      # incl. 288 -> 384 on the general-shape kernels
def test_padded_context_runs_fused_and_meets_the_oracle_of_the_real_shape(nl, sk, units):
    import torch
    from keras_nerf_amd.runtime import KnerfContext
    from tests.test_gpu_train import flat, per_tensor_err
    cfg = O.NerfConfig(n_layers=nl, dense_units=units, skip_layer=sk)
    P = make_problem(n_images=1, wh=16, weight_scale=1.5, bias_std=0.05, cfg=cfg)
    o, d, t, u, img = flat(P)
    res = {}
    for pad in (True, False):
        ctx = KnerfContext(n_layers=nl, dense_units=units, skip_layer=sk, white_background=True, pad_width=pad)
        assert ctx.get_option("general_shape_path") == (0.0 if pad and units < 256 else 1.0)      # below 256 the padded shape is a fused one
        assert ctx.param_count == O.param_count(cfg)
        ctx.set_weights(0, O.flatten_params(P["cp"])); ctx.set_weights(1, O.flatten_params(P["fp"]))
        np.testing.assert_array_equal(ctx.get_weights(0), O.flatten_params(P["cp"]))          # the real layout, round trip
        loss = torch.zeros(2, device="cuda"); ci = torch.empty((P["N"], 3), device="cuda"); fi = torch.empty_like(ci)
        ctx.train_chunk(o, d, t, img, u, inv_chunks=1.0, loss=loss, c_image=ci, f_image=fi)
        torch.cuda.synchronize()
        res[pad] = (ctx.grads(0).cpu().numpy(), ctx.grads(1).cpu().numpy(), loss.cpu().numpy().copy(), ci.cpu().numpy())
        if pad:
            n_p = ctx.padded_param_count
            gv = ctx.grads_view().cpu().numpy()
            rest = np.ones(n_p, bool); rest[ctx._pad_index_host] = False
            assert (gv[:n_p][rest] == 0).all() and (gv[n_p:][rest] == 0).all()                # padded parameters: exactly zero gradient
            for _ in range(3):
                ctx.apply_adam()
                ctx.train_chunk(o, d, t, img, u, inv_chunks=1.0, loss=loss)
            ctx.apply_adam(); torch.cuda.synchronize()
            for net in (0, 1):
                w = ctx.weights_view(net).cpu().numpy()
                assert (w[:n_p][rest] == 0).all()                                              # ... and still exactly zero weights after four Adam steps
            assert np.abs(ctx.get_weights(0) - O.flatten_params(P["cp"])).max() > 1e-4           # (the real ones moved)
        ctx.close()
    rc, lc, gc = O.chunk_loss_and_grads(P["cp"], o, d, t, img, cfg, True, emulate_bf16=O.FUSED)
    gp, gg = res[True], res[False]
    e_or = per_tensor_err(gp[0], O.flatten_params(gc), cfg)[0]
    e_gen = per_tensor_err(gg[0], O.flatten_params(gc), cfg)[0]
    assert e_or < max(1.5e-2, 1.5 * e_gen), (e_or, e_gen)                                       # the built-in shapes' tolerance (test_gpu_fused_shapes.py)
    assert abs(float(gp[2][0]) - float(lc)) < 2e-3 and np.abs(gp[3] - rc["image"]).max() < 1e-2
    assert per_tensor_err(gp[0], gg[0], cfg)[0] < 4e-2 and np.abs(gp[2] - gg[2]).max() < 2e-3  # fused-padded vs general-shape kernels


@pytest.mark.gpu
def test_a_pair_the_library_lacks_at_the_next_width_is_taken_one_width_up():
    """6 layers / skip 3 is built in at width 256 only: dense_units = 100 (next fused width 128: not in the library) runs zero-padded on
    the 256-wide fused kernels -- still faster than the general-shape kernels at 128 --; dense_units = 40 (next: 64, one up: 128, both
    missing) stays on the general-shape kernels: two widths up would cost more than it saves"""
    from keras_nerf_amd.runtime import KnerfContext
    ctx = KnerfContext(n_layers=6, dense_units=100, skip_layer=3, white_background=True)
    assert ctx.get_option("general_shape_path") == 0.0 and ctx.real_dense_units == 100 and ctx.cfg.dense_units == 256
    cfg = O.NerfConfig(n_layers=6, dense_units=100, skip_layer=3)
    assert ctx.param_count == O.param_count(cfg)
    w = O.flatten_params(O.init_params(cfg, 0))
    ctx.set_weights(0, w)
    np.testing.assert_array_equal(ctx.get_weights(0), w)
    ctx.close()
    ctx = KnerfContext(n_layers=6, dense_units=40, skip_layer=3, white_background=True)
    assert ctx.get_option("general_shape_path") == 1.0 and ctx.cfg.dense_units == 40
    ctx.close()


@pytest.mark.gpu
def test_nerf_with_a_padded_width_trains_and_checkpoints_in_the_real_layout(tmp_path):
    import torch
    from keras_nerf_amd.model.nerf.nerf import NeRF
    P = make_problem(n_images=2, wh=16)
    nerf = NeRF(dense_units=192)
    nerf.compile("adam", "mse", batch_size=2, image_height=16, image_width=16, ray_chunks=128, white_background=True)
    assert nerf._ctx.get_option("general_shape_path") == 0.0 and nerf._ctx.real_dense_units == 192
    ws = nerf.coarse.get_weights()
    assert ws[2].shape == (192, 192) and ws[0].shape == (63, 192) and ws[-2].shape == (96, 3) and nerf.coarse.count_params() == sum(w.size for w in ws)
    data = (torch.as_tensor(P["img"], device="cuda"), tuple(torch.as_tensor(P[k], device="cuda") for k in ("o", "d", "t")))
    l0 = float(dict(nerf.train_step(data))["fine_loss"])
    for _ in range(30):
        logs = nerf.train_step(data)
    assert float(dict(logs)["fine_loss"]) < l0
    nerf.save_model(str(tmp_path / "m"))
    other = NeRF(dense_units=192, model_path=str(tmp_path / "m"))
    other.compile("adam", "mse", batch_size=2, image_height=16, image_width=16, ray_chunks=128, white_background=True, is_training=False)
    for a, b in zip(nerf.fine.get_weights(), other.fine.get_weights()):
        np.testing.assert_array_equal(a, b)
    u = torch.as_tensor(P["u"], device="cuda")                          # the same fine-sampling numbers for both (each model has its own Philox stream)
    c1, f1 = nerf.predict_and_render_images(data[1], u=u); c2, f2 = other.predict_and_render_images(data[1], u=u)
    assert torch.equal(f1["image"], f2["image"]) and torch.equal(c1["image"], c2["image"])
