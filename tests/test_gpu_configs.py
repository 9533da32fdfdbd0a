"""BASELINE.json's single-GPU configurations at their real sizes, against the oracle:

cfg5  inference.py-shaped render, 256x256, ray_chunks 4096 (16 chunks): RaysGenerator -> predict_and_render_images;
      256 sampled rays against the oracle (coarse image / weights, merged t-values bit for bit, fine image / weights on
      the GPU's own t-values).
cfg3  400x400, batch 1: ray_chunks 16384 violates the divisibility assert (nerf.py:100), 16000 trains; one train_batch
      equals ten train_chunk calls; the gradients of a sampled sub-chunk against the oracle.
cfg1  the coarse-only configuration (n_fine = 0) TRAINED, not only rendered: gradients of both nets against the oracle -- at 16 x 16 in
      one chunk, and AS WRITTEN (64 x 64, four chunks of 1,024 through knerf_train_batch) through an exact zero-gradient property.
cfg4 needs eight GPUs and is the driver's to run; its per-GPU workload runs as a TWO-rank data-parallel job (gloo, one GPU) in tests/test_gpu_api.py::test_two_rank_data_parallel_step_at_cfg4_size.
"""
import numpy as np
import pytest
from keras_nerf_amd.debug import debug_buffer
import torch

from oracle import nerf_oracle as O
from tests.test_gpu_forward import log_stats
from tests.test_gpu_train import per_tensor_err

pytestmark = pytest.mark.gpu

FOV = 0.6911112070083618          # inference.py:27
# of each tensor's max |g|.  Against the oracle in the kernels' arithmetic (the check of the KERNELS: measured 1.3e-3 ... 1.1e-2, round 2)
# / against the fp32 oracle (bounded by the precision choice: the oracle's own bf16-vs-fp32 gap on these problems is 5.8e-2 ... 6.9e-2)
GRAD_TOL_EMU, GRAD_TOL_FP32 = 1.5e-2, 8e-2


def problem_weights(cfg=None):
    """the weights of the small parity tests (tests/problem.py: glorot x 1.5 + N(0, 0.05) biases): sigma is positive on a
    good part of the samples, so the gradients are carried by many samples and the bf16-vs-fp32 gap is the 3-5 % that
    DESIGN.md section 4 states (with near-empty density fields the trunk gradients are rounding noise in BOTH arithmetics)"""
    from tests.problem import make_problem
    P = make_problem(n_images=1, wh=16, weight_scale=1.5, bias_std=0.05, cfg=cfg)     # wh fixes the bias draw
    return P["cp"], P["fp"]


def check_grads_against_oracle(tag, g, n, loss, cfg, white, coarse_args, fine_args):
    """per-tensor gradient error (of the tensor's max |g|) of both nets: against the oracle in the kernels' arithmetic to
    GRAD_TOL_EMU; against the fp32 oracle (the reference's arithmetic) to GRAD_TOL_FP32, where the problem itself must be
    well conditioned (the oracle's own bf16-vs-fp32 gap is asserted to stay under that tolerance too)"""
    ref = {}
    for emu in (O.FUSED, False):
        for k, a in (("c", coarse_args), ("f", fine_args)):
            _, l, gr = O.chunk_loss_and_grads(a[0], a[1], a[2], a[3], a[4], cfg, white, emulate_bf16=emu)
            ref[k, emu] = (float(l), O.flatten_params(gr))
    for emu, tol in ((O.FUSED, GRAD_TOL_EMU), (False, GRAD_TOL_FP32)):
        ec, ef = per_tensor_err(g[:n], ref["c", emu][1], cfg), per_tensor_err(g[n:], ref["f", emu][1], cfg)
        log_stats(f"{tag}_grads_emulate_{emu}", coarse_worst=ec[0], fine_worst=ef[0])
        assert ec[0] < tol and ef[0] < tol, (emu, ec, ef)
        assert abs(float(loss[0]) - ref["c", emu][0]) < 2e-3 and abs(float(loss[1]) - ref["f", emu][0]) < 2e-3
    gap = max(per_tensor_err(ref[k, O.FUSED][1], ref[k, False][1], cfg)[0] for k in "cf")
    log_stats(f"{tag}_oracle_bf16_vs_fp32_gap", gap=gap)
    assert gap < GRAD_TOL_FP32, gap


def test_cfg5_render_256_in_sixteen_chunks_against_oracle():
    from keras_nerf_amd.data.rays import RaysGenerator
    from keras_nerf_amd.data.utils import get_focal_from_fov, pose_spherical
    from keras_nerf_amd.model.nerf.nerf import NeRF
    wh, R = 256, 4096
    cfg = O.NerfConfig()
    cp, fp = problem_weights()
    nerf = NeRF(seed=0)
    nerf.compile("adam", "mse", batch_size=1, image_height=wh, image_width=wh, ray_chunks=R, white_background=True, is_training=False)
    assert nerf.sequential_chunks == 16
    nerf.coarse.set_flat_weights(O.flatten_params(cp)); nerf.fine.set_flat_weights(O.flatten_params(fp))
    rg = RaysGenerator(get_focal_from_fov(FOV, wh), wh, wh, 2.0, 6.0, cfg.n_coarse, seed=3)
    o, d, t = rg(pose_spherical(70.0, -30.0, 4.0))                      # inference.py:84-87
    N = wh * wh
    u = torch.rand((N, cfg.n_fine), device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    coarse, fine = nerf.predict_and_render_images((o[None], d[None], t[None]), u=u)
    assert fine["image"].shape == (1, wh, wh, 3) and fine["depth"].shape == (1, wh, wh) and fine["weights"].shape == (1, wh, wh, 192)
    # the same library call with the merged t-values exposed
    of, df, tf_ = o.reshape(N, 3).contiguous(), d.reshape(N, 3).contiguous(), t.reshape(N, -1).contiguous()
    buf = dict(c_image=torch.empty((N, 3), device="cuda"), c_weights=torch.empty((N, 64), device="cuda"),
               f_image=torch.empty((N, 3), device="cuda"), f_weights=torch.empty((N, 192), device="cuda"),
               t_fine=torch.empty((N, 192), device="cuda"))
    nerf._ctx.render_batch(of, df, tf_, u, 0, R, out=buf)
    assert torch.equal(buf["f_image"].reshape(1, wh, wh, 3), fine["image"]) and torch.equal(buf["c_image"].reshape(1, wh, wh, 3), coarse["image"])
    # 256 rays, a few from every chunk
    idx = np.sort(np.random.default_rng(1).choice(N, 256, replace=False))
    assert len(set(idx // R)) == 16
    g = lambda x: x.cpu().numpy()[idx]
    so, sd, st, su = g(of), g(df), g(tf_), g(u)
    rc = O.predict_and_render_chunk_single(cp, so, sd, st, cfg, True, emulate_bf16=O.FUSED)
    ci, cw, tfine = g(buf["c_image"]), g(buf["c_weights"]), g(buf["t_fine"])
    np.testing.assert_array_equal(tfine, O.fine_points(st, cw, su, "zero"))          # sampler + 192-way merge: bit exact
    rf = O.predict_and_render_chunk_single(fp, so, sd, tfine, cfg, True, emulate_bf16=O.FUSED)
    rf32 = O.predict_and_render_chunk_single(fp, so, sd, tfine, cfg, True)
    fi, fw = g(buf["f_image"]), g(buf["f_weights"])
    log_stats("cfg5_render_256", c_img=np.abs(ci - rc["image"]).max(), c_w=np.abs(cw - rc["weights"]).max(),
              f_img=np.abs(fi - rf["image"]).max(), f_w=np.abs(fw - rf["weights"]).max(), f_img_fp32=np.abs(fi - rf32["image"]).max())
    np.testing.assert_allclose(ci, rc["image"], atol=1e-2); np.testing.assert_allclose(cw, rc["weights"], atol=1e-2)
    np.testing.assert_allclose(fi, rf["image"], atol=1e-2); np.testing.assert_allclose(fw, rf["weights"], atol=1e-2)
    assert np.abs(fi - rf32["image"]).max() < 2e-2                                   # stated tolerance
    assert cw.std() > 1e-3 and fw.std() > 1e-3


def test_grouped_wgrad_path_meets_the_oracle():
    """the grouped coarse weight-gradient launch (knerf_train_batch, G = 2: two chunks' coarse passes in two workspace regions, one
    launch) against the ORACLE, not only against the ungrouped kernels: 2 chunks of 128 rays, gradients of both nets per tensor"""
    from keras_nerf_amd.runtime import KnerfContext
    from keras_nerf_amd.debug import debug_buffer
    from tests.problem import make_problem
    P = make_problem(n_images=1, wh=16, weight_scale=1.5, bias_std=0.05)
    cfg, N, R = P["cfg"], 256, 128
    o, d, t, u, img = (P[k].reshape(N, -1) for k in ("o", "d", "t", "u", "img"))
    ctx = KnerfContext(white_background=True, options=dict(wgrad_group_max=2, merge_chunk_rays=0))     # the caller's own chunks: this test is about their grouped launch
    ctx.set_weights(0, O.flatten_params(P["cp"])); ctx.set_weights(1, O.flatten_params(P["fp"]))
    loss = torch.zeros(2, device="cuda")
    tf_all = torch.empty((N, 192), device="cuda")
    ctx.zero_grads()
    ctx.train_batch(o, d, t, img, u, ray_chunks=R, loss=loss)
    torch.cuda.synchronize()
    assert ctx.get_option("wgrad_group") == 2.0
    g = ctx.grads_view().cpu().numpy(); n = ctx.param_count
    # reference: mean over the two chunks of the per-chunk gradients (nerf.py:383-384), fine pass on the oracle's own merged
    # t-values (bit-equal to the kernels' given the same coarse weights; the sampler tests pin that)
    gc_ref = np.zeros(n, np.float32); gf_ref = np.zeros(n, np.float32); lc = lf = 0.0
    for c in range(2):
        sl = slice(c * R, (c + 1) * R)
        rc, l0, gc = O.chunk_loss_and_grads(P["cp"], o[sl], d[sl], t[sl], img[sl], cfg, True, emulate_bf16=O.FUSED)
        tf_ = O.fine_points(t[sl], rc["weights"], u[sl], "zero")
        _, l1, gf = O.chunk_loss_and_grads(P["fp"], o[sl], d[sl], tf_, img[sl], cfg, True, emulate_bf16=O.FUSED)
        gc_ref += O.flatten_params(gc) / 2; gf_ref += O.flatten_params(gf) / 2; lc += float(l0) / 2; lf += float(l1) / 2
    ec, ef = per_tensor_err(g[:n], gc_ref, cfg), per_tensor_err(g[n:], gf_ref, cfg)
    log_stats("grouped_wgrad_vs_oracle", coarse_worst=ec[0], fine_worst=ef[0])
    assert ec[0] < GRAD_TOL_EMU, ec
    # the fine pass is not grouped; it runs on the GPU's own importance samples, and a percent-level difference in a coarse weight
    # moves individual samples across bins (oob = zero is discontinuous), so its sparse gradients differ more than the kernels err
    # (the per-chunk tests compare the fine pass on the GPU's own t-values): a sanity bound only
    assert ef[0] < 0.3, ef
    assert abs(float(loss[0]) - lc) < 2e-3 and abs(float(loss[1]) - lf) < 2e-3
    ctx.close()


def test_cfg2_train_batch_at_bench_size_against_oracle():
    """BASELINE configs[1] (the benched workload): 2 x 128 x 128 rays in 8 chunks of 4096 through knerf_train_batch.  The images
    the training pass returns, for 256 rays sampled from all chunks, against the oracle (coarse on the given t-values, fine on
    the pass's own merged t-values, recovered bit-exactly from its coarse weights); the step's losses are the MSE of those images;
    the gradients are finite and every tensor of both nets received one."""
    from keras_nerf_amd.data.rays import RaysGenerator
    from keras_nerf_amd.data.utils import get_focal_from_fov, pose_spherical
    from keras_nerf_amd.model.nerf.nerf import NeRF
    wh, B, R = 128, 2, 4096
    cfg = O.NerfConfig()
    cp, fp = problem_weights()
    nerf = NeRF(seed=0)
    nerf.compile("adam", "mse", batch_size=B, image_height=wh, image_width=wh, ray_chunks=R, white_background=True)
    assert nerf.sequential_chunks == 8
    nerf.coarse.set_flat_weights(O.flatten_params(cp)); nerf.fine.set_flat_weights(O.flatten_params(fp))
    ctx = nerf._ctx
    rg = RaysGenerator(get_focal_from_fov(FOV, wh), wh, wh, 2.0, 6.0, cfg.n_coarse, seed=21)
    o, d, t = rg(np.stack([pose_spherical(15.0, -30.0, 4.0), pose_spherical(190.0, -30.0, 4.0)]))
    N = B * wh * wh
    o, d, t = o.reshape(N, 3).contiguous(), d.reshape(N, 3).contiguous(), t.reshape(N, -1).contiguous()
    gen = torch.Generator(device="cuda").manual_seed(11)
    tgt = torch.rand((N, 3), device="cuda", generator=gen); u = torch.rand((N, cfg.n_fine), device="cuda", generator=gen)
    loss = torch.zeros(2, device="cuda"); ci = torch.empty((N, 3), device="cuda"); fi = torch.empty((N, 3), device="cuda")
    ctx.zero_grads()
    ctx.train_batch(o, d, t, tgt, u, seed=0, ray_chunks=R, loss=loss, c_image=ci, f_image=fi)
    g = ctx.grads_view().cpu().numpy()
    assert np.isfinite(g).all()
    off = 0
    for net in range(2):
        for name, fin, fout in O.layer_shapes(cfg):
            for n_el in (fin * fout, fout):
                assert np.abs(g[off:off + n_el]).max() > 0, (net, name)
                off += n_el
    assert abs(float(((ci - tgt) ** 2).mean()) - float(loss[0])) < 1e-5 and abs(float(((fi - tgt) ** 2).mean()) - float(loss[1])) < 1e-5
    # the forward-only path on the same rays and u gives the same images (training and rendering share the kernels' arithmetic)
    buf = dict(c_image=torch.empty((N, 3), device="cuda"), c_weights=torch.empty((N, 64), device="cuda"),
               f_image=torch.empty((N, 3), device="cuda"), t_fine=torch.empty((N, 192), device="cuda"))
    ctx.render_batch(o, d, t, u, 0, R, out=buf)
    assert torch.equal(buf["c_image"], ci) and torch.equal(buf["f_image"], fi)
    idx = np.sort(np.random.default_rng(4).choice(N, 256, replace=False))
    assert len(set(idx // R)) == 8
    s = lambda x: x.cpu().numpy()[idx]
    so, sd, st, su = s(o), s(d), s(t), s(u)
    rc = O.predict_and_render_chunk_single(cp, so, sd, st, cfg, True, emulate_bf16=O.FUSED)
    tfine = s(buf["t_fine"])
    np.testing.assert_array_equal(tfine, O.fine_points(st, s(buf["c_weights"]), su, "zero"))
    rf = O.predict_and_render_chunk_single(fp, so, sd, tfine, cfg, True, emulate_bf16=O.FUSED)
    rf32 = O.predict_and_render_chunk_single(fp, so, sd, tfine, cfg, True)
    log_stats("cfg2_train_batch_images", c_img=np.abs(s(ci) - rc["image"]).max(), f_img=np.abs(s(fi) - rf["image"]).max(),
              f_img_fp32=np.abs(s(fi) - rf32["image"]).max())
    np.testing.assert_allclose(s(ci), rc["image"], atol=1e-2); np.testing.assert_allclose(s(fi), rf["image"], atol=1e-2)
    assert np.abs(s(fi) - rf32["image"]).max() < 2e-2
    ctx.apply_adam()
    assert ctx.step == 1


def test_cfg3_400x400_ten_chunks_of_16000():
    from keras_nerf_amd.data.rays import RaysGenerator
    from keras_nerf_amd.data.utils import get_focal_from_fov, pose_spherical
    from keras_nerf_amd.model.nerf.nerf import NeRF
    wh, R = 400, 16000
    cfg = O.NerfConfig()
    cp, fp = problem_weights()
    nerf = NeRF(seed=0)
    with pytest.raises(AssertionError):                                              # nerf.py:100: 160000 % 16384 = 12544
        nerf.compile("adam", "mse", batch_size=1, image_height=wh, image_width=wh, ray_chunks=16384, white_background=True)
    nerf.compile("adam", "mse", batch_size=1, image_height=wh, image_width=wh, ray_chunks=R, white_background=True)
    assert nerf.sequential_chunks == 10 and nerf.num_rays == 160000
    nerf.coarse.set_flat_weights(O.flatten_params(cp)); nerf.fine.set_flat_weights(O.flatten_params(fp))
    ctx = nerf._ctx
    N = wh * wh
    o, d, t = RaysGenerator(get_focal_from_fov(FOV, wh), wh, wh, 2.0, 6.0, cfg.n_coarse, seed=9)(pose_spherical(200.0, -30.0, 4.0))
    o, d, t = o.reshape(N, 3).contiguous(), d.reshape(N, 3).contiguous(), t.reshape(N, -1).contiguous()
    gen = torch.Generator(device="cuda").manual_seed(7)
    tgt = torch.rand((N, 3), device="cuda", generator=gen); u = torch.rand((N, cfg.n_fine), device="cuda", generator=gen)
    res = []
    for whole in (True, False):
        loss = torch.zeros(2, device="cuda"); ci = torch.empty((N, 3), device="cuda"); fi = torch.empty((N, 3), device="cuda")
        ctx.zero_grads()
        if whole:
            ctx.train_batch(o, d, t, tgt, u, seed=0, ray_chunks=R, loss=loss, c_image=ci, f_image=fi)
        else:
            for c in range(N // R):
                sl = slice(c * R, (c + 1) * R)
                ctx.train_chunk(o[sl], d[sl], t[sl], tgt[sl], u[sl], ray_offset=c * R, inv_chunks=R / N, loss=loss, c_image=ci[sl], f_image=fi[sl])
        torch.cuda.synchronize()
        res.append((ci, fi, loss.clone(), ctx.grads_view().clone()))
    (c0, f0, l0, g0), (c1, f1, l1, g1) = res
    assert torch.equal(c0, c1) and torch.equal(f0, f1)
    assert float((l0 - l1).abs().max()) < 1e-6
    assert float((g0 - g1).abs().max()) <= 2e-5 * float(g0.abs().max()) and float(g0.abs().max()) > 0   # fp32 summation order
    assert abs(float(((f0 - tgt) ** 2).mean()) - float(l0[1])) < 1e-5                # the loss is the MSE of the returned image
    # a sampled sub-chunk (rays from all over the image) against the oracle
    idx = np.sort(np.random.default_rng(2).choice(N, 192, replace=False))
    ti = torch.as_tensor(idx, device="cuda")
    so, sd, st, sg, su = (x[ti].contiguous() for x in (o, d, t, tgt, u))
    ctx.zero_grads()
    loss = torch.zeros(2, device="cuda")
    ctx.train_chunk(so, sd, st, sg, su, loss=loss)
    g = ctx.grads_view().cpu().numpy(); n = ctx.param_count
    t_fine = debug_buffer(ctx, 5).view(torch.float32).cpu().numpy()[:192 * 192].reshape(192, 192)
    so, sd, st, sg = (x.cpu().numpy() for x in (so, sd, st, sg))
    check_grads_against_oracle("cfg3_subchunk", g, n, loss, cfg, True, (cp, so, sd, st, sg), (fp, so, sd, t_fine, sg))
    # the class-level step at this size (nerf.py:332-473)
    ctx.zero_grads()
    before = nerf.fine.get_flat_weights()
    logs = nerf.train_step((tgt.reshape(1, wh, wh, 3), (o.reshape(1, wh, wh, 3), d.reshape(1, wh, wh, 3), t.reshape(1, wh, wh, -1))),
                           u=u, with_metrics=False)
    assert abs(float(logs["fine_loss"]) - float(l0[1])) < 1e-5
    assert 0 < np.abs(nerf.fine.get_flat_weights() - before).max() <= 1.01e-3       # one Adam step of lr 1e-3


def test_coarse_only_configuration_trains_both_nets_against_oracle():
    """BASELINE configs[0] (coarse-only, 64 samples): with n_fine = 0 the reference's fine branch degenerates to
    sort(concat(t, [])) = t (nerf.py:182-191), so both networks train on the coarse t-values."""
    from keras_nerf_amd.runtime import KnerfContext
    cfg = O.NerfConfig(n_coarse=64, n_fine=0)
    cp, fp = problem_weights(cfg)
    rng = np.random.default_rng(0)
    wh = 16
    o, d, t = O.generate_rays(O.pose_spherical(33.0, -30.0, 4.0), O.get_focal_from_fov(FOV, wh), wh, wh, 2.0, 6.0, 64, rng.random((wh, wh, 64)))
    N = wh * wh
    o, d, t = o.reshape(N, 3), d.reshape(N, 3), t.reshape(N, 64)
    img = rng.random((N, 3), dtype=np.float32)
    ctx = KnerfContext(n_coarse=64, n_fine=0, white_background=False)
    ctx.set_weights(0, O.flatten_params(cp)); ctx.set_weights(1, O.flatten_params(fp))
    loss = torch.zeros(2, device="cuda")
    ctx.train_chunk(o, d, t, img, None, loss=loss)
    g = ctx.grads_view().cpu().numpy(); n = ctx.param_count
    t_fine = debug_buffer(ctx, 5).view(torch.float32).cpu().numpy()[:N * 64].reshape(N, 64)
    np.testing.assert_array_equal(t_fine, t)
    check_grads_against_oracle("coarse_only_train", g, n, loss, cfg, False, (cp, o, d, t, img), (fp, o, d, t, img))
    ctx.apply_adam()
    assert ctx.step == 1 and np.abs(ctx.get_weights(0) - O.flatten_params(cp)).max() > 0
    ctx.close()


@pytest.mark.parametrize("merge", [0, 4096])
def test_cfg1_as_written_64x64_in_four_chunks_of_1024_against_oracle(merge):
    """merge = 0: the four chunks as four sets of launches; merge = 4096 (the library's default, option merge_chunk_rays): the same
    call runs them as ONE 4,096-ray set of launches -- the same oracle comparison must hold.
    BASELINE configs[0] AS WRITTEN on the GPU: lego-shaped 64 x 64 image, batch 1, ray_chunks 1024 -> 4 chunks,
    coarse-only (n_fine = 0: the second network runs on the coarse t-values, nerf.py:182-191), through knerf_train_batch -- full-size
    launches, the coarse weight gradients of the four chunks in one grouped launch.  The NumPy oracle cannot do 4,096 rays in
    seconds, so the comparison uses an EXACT property instead of a smaller problem: a ray whose target equals the rendered pixel has
    dL/dimage = 0 and contributes exactly nothing to any gradient (utils.py:36-58, train_single.py:127).  The step runs with the
    net's own render as the target everywhere except on 256 sampled rays (64 from every chunk), which keep random targets: the
    accumulated gradient of all 4,096 rays is then the oracle's gradient of those 256 rays x 256/4096, per tensor, and the step's
    loss the oracle's x 256/4096.  Once per net (the two nets render different images), the coarse check with dead-tile skipping
    on (lists with holes: 7/8 of the tiles are dead), the second net's with skipping off (contiguous launches)."""
    from keras_nerf_amd.data.rays import RaysGenerator
    from keras_nerf_amd.data.utils import get_focal_from_fov, pose_spherical
    from keras_nerf_amd.runtime import KnerfContext
    wh, R, C = 64, 1024, 4
    N = wh * wh
    cfg = O.NerfConfig(n_coarse=64, n_fine=0)
    cp, fp = problem_weights(cfg)
    o, d, t = RaysGenerator(get_focal_from_fov(FOV, wh), wh, wh, 2.0, 6.0, 64, seed=4)(pose_spherical(55.0, -30.0, 4.0))
    o, d, t = o.reshape(N, 3).contiguous(), d.reshape(N, 3).contiguous(), t.reshape(N, 64).contiguous()
    rng = np.random.default_rng(12)
    idx = np.sort(np.concatenate([c * R + rng.choice(R, 64, replace=False) for c in range(C)]))
    ti = torch.as_tensor(idx, device="cuda")
    rnd = torch.rand((N, 3), device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    so, sd, st, sg = (x[ti].cpu().numpy() for x in (o, d, t, rnd))
    for net, params, skip in ((0, cp, 1), (1, fp, 0)):
        ctx = KnerfContext(n_coarse=64, n_fine=0, white_background=False, options=dict(skip_dead_tiles=skip, merge_chunk_rays=merge))
        assert ctx.get_option("merge_chunk_rays") == merge
        ctx.set_weights(0, O.flatten_params(cp)); ctx.set_weights(1, O.flatten_params(fp))
        ren = ctx.render_batch(o, d, t, None, ray_chunks=R)
        tgt = ren["c_image" if net == 0 else "f_image"].clone()
        tgt[ti] = rnd[ti]
        loss = torch.zeros(2, device="cuda"); ci = torch.empty((N, 3), device="cuda"); fi = torch.empty((N, 3), device="cuda")
        ctx.zero_grads(); ctx.tile_stats(reset=True)
        ctx.train_batch(o, d, t, tgt, None, ray_chunks=R, loss=loss, c_image=ci, f_image=fi)
        torch.cuda.synchronize()
        assert torch.equal(ci, ren["c_image"]) and torch.equal(fi, ren["f_image"])      # training and rendering share the kernels' arithmetic
        assert ctx.get_option("wgrad_group") == (4.0 if merge == 0 else 1.0)            # merge = 0: the four coarse passes left in ONE weight-gradient launch
        live, total = ctx.tile_stats(reset=True)
        if skip:                                                                        # 2 passes x 4 chunks x 2,048 tiles; the net under test keeps its 256 rays' tiles
            assert total == 2 * C * R * 64 // 32 and 0 < live <= 256 * 2 + total // 2, (live, total)     # the same tiles either way
        n = ctx.param_count
        g = ctx.grads_view().cpu().numpy()[net * n:(net + 1) * n] * (N / 256.0)
        got_loss = float(loss[net]) * (N / 256.0)
        mine = (ci if net == 0 else fi)[ti].cpu().numpy()
        for emu, tol in ((O.FUSED, GRAD_TOL_EMU), (False, GRAD_TOL_FP32)):
            r, l, gr = O.chunk_loss_and_grads(params, so, sd, st, sg, cfg, False, emulate_bf16=emu)
            e = per_tensor_err(g, O.flatten_params(gr), cfg)
            log_stats(f"cfg1_as_written_net{net}_emulate_{emu}", worst=e[0], loss=abs(got_loss - float(l)), img=np.abs(mine - r["image"]).max())
            assert e[0] < tol, (net, emu, e)
            assert abs(got_loss - float(l)) < 3e-3, (got_loss, float(l))
            np.testing.assert_allclose(mine, r["image"], atol=1e-2 if emu else 2e-2)
        ctx.apply_adam()
        assert ctx.step == 1
        ctx.close()


def test_lego_camera_of_the_reference_ray_test():
    """The one camera matrix the reference's tests hold (tests/data/test_rays.py:21-47; committed as data under
    tests/golden/lego_c2w.json) through RaysGenerator with that test's own assertions (:50-87), and against the oracle."""
    import json
    import os
    from keras_nerf_amd.data.rays import RaysGenerator
    F = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "lego_c2w.json")))
    c2w = np.asarray(F["camera_to_world"], np.float32)
    W, H, S = F["image_width"], F["image_height"], F["n_sample"]
    rg = RaysGenerator(F["focal_length"], W, H, F["near"], F["far"], S)
    last = None
    for _ in range(4):
        o, d, t = [x.cpu().numpy() for x in rg(c2w)]
        assert o.shape == (H, W, 3) and d.shape == (H, W, 3) and t.shape == (H, W, S)
        assert o.dtype == np.float32 and d.dtype == np.float32 and t.dtype == np.float32
        assert not np.isnan(o).any() and not np.isnan(d).any() and not np.isnan(t).any()
        assert t.min() >= 2.0 - 4.0 / 32.0 and t.max() <= 6.0 + 4.0 / 32.0
        if last is not None:
            assert np.allclose(last[0], o) and np.allclose(last[1], d) and np.allclose(last[2], t, atol=4.0 / 32.0)
            assert np.abs(last[2] - t).max() > 0                                    # the jitter is redrawn per call (rays.py:122-123)
        last = (o, d, t)
    noise = np.random.default_rng(0).random((H, W, S), dtype=np.float32)
    o, d, t = [x.cpu().numpy() for x in rg(c2w, noise=noise)]
    eo, ed, et = O.generate_rays(c2w, F["focal_length"], W, H, F["near"], F["far"], S, noise)
    np.testing.assert_array_equal(o, eo); np.testing.assert_allclose(d, ed, atol=2e-7); np.testing.assert_allclose(t, et, atol=1e-6)
    np.testing.assert_allclose(np.linalg.norm(d, axis=-1), 1.0, atol=1e-6)            # rays.py:108-109
    np.testing.assert_array_equal(o, np.broadcast_to(c2w[:3, 3], o.shape))            # rays.py:111-113


@pytest.mark.parametrize("nc,nf", [(512, 256), (320, 704), (96, 32)])
def test_sample_counts_beyond_the_defaults(nc, nf):
    """train_single.py:28-29: --num_coarse_samples / --num_fine_samples are free integers.  Rounds 1-3 stopped at 256 coarse and 512 total
    (a lane's run in compositing, static LDS tables in the sampler); now 512 / 1024: compositing templates of 12 and 16 samples per
    lane, sampler tables in dynamic LDS.  16 rays through coarse pass, sampler (merged t bit for bit), fine pass and one train chunk
    against the oracle; the limits themselves are checked at creation."""
    from keras_nerf_amd.runtime import KnerfContext
    from tests.problem import make_problem
    cfg = O.NerfConfig(n_coarse=nc, n_fine=nf)
    P = make_problem(n_images=1, wh=4, weight_scale=1.5, bias_std=0.05, cfg=cfg)
    N = P["N"]
    o, d, t, u, img = P["o"].reshape(N, 3), P["d"].reshape(N, 3), P["t"].reshape(N, nc), P["u"].reshape(N, nf), P["img"].reshape(N, 3)
    ctx = KnerfContext(n_coarse=nc, n_fine=nf, white_background=True)
    ctx.set_weights(0, O.flatten_params(P["cp"])); ctx.set_weights(1, O.flatten_params(P["fp"]))
    out = {k: v.cpu().numpy() for k, v in ctx.render_chunk(o, d, t, u).items()}
    c = O.predict_and_render_chunk_single(P["cp"], o, d, t, cfg, True, emulate_bf16=O.FUSED)
    np.testing.assert_allclose(out["c_image"], c["image"], atol=1e-2); np.testing.assert_allclose(out["c_weights"], c["weights"], atol=1e-2)
    assert out["t_fine"].shape == (N, nc + nf)
    np.testing.assert_array_equal(out["t_fine"], O.fine_points(t, out["c_weights"], u, "zero"))          # sampler + merge: bit-exact on the GPU's own weights
    f = O.predict_and_render_chunk_single(P["fp"], o, d, out["t_fine"], cfg, True, emulate_bf16=O.FUSED)
    np.testing.assert_allclose(out["f_image"], f["image"], atol=1e-2); np.testing.assert_allclose(out["f_weights"], f["weights"], atol=1e-2)
    np.testing.assert_allclose(out["f_depth"], f["depth"], atol=5e-2)
    loss = torch.zeros(2, device="cuda")
    ctx.train_chunk(o, d, t, img, u, loss=loss)
    g = ctx.grads_view().cpu().numpy(); n = ctx.param_count
    t_fine = debug_buffer(ctx, 5).view(torch.float32).cpu().numpy()[:N * (nc + nf)].reshape(N, nc + nf)
    for net, (params, tt) in enumerate(((P["cp"], t), (P["fp"], t_fine))):
        _, l, gr = O.chunk_loss_and_grads(params, o, d, tt, img, cfg, True, emulate_bf16=O.FUSED)
        e = per_tensor_err(g[net * n:(net + 1) * n], O.flatten_params(gr), cfg)
        log_stats(f"sample_counts_{nc}_{nf}_net{net}", worst=e[0], loss=abs(float(loss[net]) - float(l)))
        assert e[0] < 2.5e-2, (net, e)         # 16 rays: the tolerance of the small train test (tests/test_gpu_train.py); measured 2e-3 ... 1.7e-2
        assert abs(float(loss[net]) - float(l)) < 2e-3
    assert bool(ctx.get_option("skip_dead_tiles_active")) == (nc % 32 == 0 and (nc + nf) % 32 == 0)
    ctx.close()
    for bad in (dict(n_coarse=513, n_fine=0), dict(n_coarse=512, n_fine=513), dict(n_coarse=1, n_fine=8)):
        with pytest.raises(ValueError):
            KnerfContext(**bad)
