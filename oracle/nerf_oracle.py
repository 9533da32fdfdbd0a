"""CPU oracle for the keras_nerf hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package.  The product path (``keras_nerf_amd``) never imports it and fails
loudly when the HIP library is missing.

PARITY UNPINNED: the reference (naufalso/keras_nerf) is pure TensorFlow, TensorFlow is not
installed here (ordinary ModuleNotFoundError, no permission denial) and the reference's own
tests assert shapes only.  This file is therefore a restatement, in
NumPy, of the algorithm as written in the reference sources; it is pinned by analytic
known-answer tests, by an fp64-vs-fp32 self check, by a torch-autograd cross check of the
hand-written backward (tests/test_oracle_grad.py) and by the single known answer the reference
holds (tests/data/test_utils.py:5-10, focal length).

Every function cites the reference file:line it restates (paths relative to the reference root).
All randomness (jitter noise, inverse-CDF ``u``, initial weights) is an explicit input because
TensorFlow's RNG streams cannot be reproduced outside TensorFlow.
"""
from __future__ import annotations

import dataclasses
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

# --------------------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------------------


@dataclasses.dataclass(frozen=True)
class NerfConfig:
    """Constructor arguments of ``NeRF`` (keras_nerf/model/nerf/nerf.py:11-14)."""

    n_coarse: int = 64
    n_fine: int = 128
    pos_emb_xyz: int = 10
    pos_emb_dir: int = 4
    n_layers: int = 8
    dense_units: int = 256
    skip_layer: int = 4

    @property
    def xyz_dim(self) -> int:
        return 3 + 6 * self.pos_emb_xyz

    @property
    def dir_dim(self) -> int:
        return 3 + 6 * self.pos_emb_dir


def layer_shapes(cfg: NerfConfig) -> List[Tuple[str, int, int]]:
    """(name, fan_in, fan_out) in Keras layer-creation order (mlp.py:11-27).

    The skip concat ``[h, xyz_enc]`` happens AFTER layer i when ``i % skip == 0 and i > 0``
    (mlp.py:36-38), so layer i+1 sees ``dense_units + xyz_dim`` inputs.
    """
    shapes = []
    fan_in = cfg.xyz_dim
    for i in range(cfg.n_layers):
        shapes.append((f"layer_{i}", fan_in, cfg.dense_units))
        fan_in = cfg.dense_units
        if i % cfg.skip_layer == 0 and i > 0:
            fan_in = cfg.dense_units + cfg.xyz_dim
    trunk_out = fan_in
    shapes.append(("sigma", trunk_out, 1))
    shapes.append(("features", trunk_out, cfg.dense_units))
    shapes.append(("rgb_features", cfg.dense_units + cfg.dir_dim, cfg.dense_units // 2))
    shapes.append(("rgb", cfg.dense_units // 2, 3))
    return shapes


def param_count(cfg: NerfConfig) -> int:
    return sum(i * o + o for _, i, o in layer_shapes(cfg))


def init_params(cfg: NerfConfig, seed: int, dtype=np.float32) -> List[np.ndarray]:
    """Keras ``glorot_uniform`` kernels + zero biases (mlp.py:5, Dense defaults).

    Returns the 24-tensor list ``[k0, b0, k1, b1, ...]`` in ``trainable_variables`` order.
    """
    rng = np.random.default_rng(seed)
    params: List[np.ndarray] = []
    for _, fi, fo in layer_shapes(cfg):
        lim = np.sqrt(6.0 / (fi + fo))
        params.append(rng.uniform(-lim, lim, size=(fi, fo)).astype(dtype))
        params.append(np.zeros((fo,), dtype))
    return params


def flatten_params(params: Sequence[np.ndarray]) -> np.ndarray:
    return np.concatenate([np.asarray(p).reshape(-1) for p in params])


def unflatten_params(flat: np.ndarray, cfg: NerfConfig) -> List[np.ndarray]:
    out, off = [], 0
    for _, fi, fo in layer_shapes(cfg):
        out.append(flat[off:off + fi * fo].reshape(fi, fo)); off += fi * fo
        out.append(flat[off:off + fo]); off += fo
    assert off == flat.size
    return out


# --------------------------------------------------------------------------------------
# bf16 emulation (used to pin the MFMA kernels tightly; NOT part of the reference)
# --------------------------------------------------------------------------------------


def round_bf16(x: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even fp32 -> bf16 -> fp32 (finite inputs): (u + 0x7FFF + bit 16 of u) & 0xFFFF0000 on the bit pattern u.
    In uint32 with in-place steps (round 5; rounds 1-4 widened to uint64 and spent 70 % of every oracle call here): the sum cannot
    wrap for finite inputs (u <= 0xFF7FFFFF).  tests/test_oracle_kat.py pins it against the widened form and hand-picked ties."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    r = u >> np.uint32(16)
    r &= np.uint32(1)
    r += np.uint32(0x7FFF)
    r += u
    r &= np.uint32(0xFFFF0000)
    return r.view(np.float32).reshape(x.shape)


# --------------------------------------------------------------------------------------
# NeRFUtils  (keras_nerf/model/nerf/utils.py)
# --------------------------------------------------------------------------------------


def positional_encoding(x: np.ndarray, L: int) -> np.ndarray:
    """utils.py:176-186: concat([x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)], -1).

    No pi factor; every block is 3 wide.
    """
    parts = [x]
    for i in range(L):
        s = x.dtype.type(2.0 ** i)
        parts.append(np.sin(s * x))
        parts.append(np.cos(s * x))
    return np.concatenate(parts, axis=-1)


def encode_position_and_directions(o, d, t, Lx: int, Ld: int):
    """utils.py:188-210: p = o[...,None,:] + d[...,None,:] * t[...,None]; PE(p, Lx); PE(broadcast d, Ld)."""
    p = o[..., None, :] + d[..., None, :] * t[..., None]
    xyz = positional_encoding(p, Lx)
    dd = np.broadcast_to(d[..., None, :], p.shape)
    dire = positional_encoding(np.ascontiguousarray(dd), Ld)
    return xyz, dire


def render_image_depth_chunk(rgb, sigma, t, white_background: bool, epsilon=1e-10, want_cache=False):
    """utils.py:16-58.  rgb [R,S,3], sigma [R,S,1] or [R,S], t [R,S] -> image [R,3], depth [R], weights [R,S].

    delta_last = epsilon (1e-10, NOT 1e10); alpha = 1-exp(-sigma*delta); exp_alpha = 1-alpha;
    T = cumprod(exp_alpha + epsilon, exclusive); w = alpha*T; white bg adds 1-sum(w); clip to [0,1].
    """
    dt = t.dtype.type
    if sigma.ndim == 3:
        sigma = sigma[..., 0]
    eps = dt(epsilon)
    delta = np.concatenate([t[..., 1:] - t[..., :-1], np.full(t.shape[:-1] + (1,), eps, t.dtype)], axis=-1)
    ex = np.exp(-sigma * delta)
    alpha = dt(1.0) - ex
    e = dt(1.0) - alpha
    x = e + eps
    T = np.concatenate([np.ones_like(x[..., :1]), np.cumprod(x[..., :-1], axis=-1, dtype=t.dtype)], axis=-1)
    w = alpha * T
    image = np.sum(w[..., None] * rgb, axis=-2)
    depth = np.sum(w * t, axis=-1)
    if white_background:
        image = image + (dt(1.0) - np.sum(w, axis=-1)[..., None])
    pre = image
    image = np.clip(image, dt(0.0), dt(1.0))
    if want_cache:
        cache = dict(rgb=rgb, sigma=sigma, delta=delta, ex=ex, alpha=alpha, x=x, T=T, w=w, pre=pre,
                     white=white_background)
        return image, depth, w, cache
    return image, depth, w


def render_backward(cache, dimage):
    """Gradient of render_image_depth_chunk wrt rgb and sigma given dL/dimage [R,3] (depth carries no loss).

    TF semantics restated: clip_by_value passes gradient where min <= x <= max (inclusive);
    cumprod(exclusive) gradient = reverse-exclusive-cumsum(T * dT) / x; exp gradient uses its output.
    """
    rgb, sigma, delta, ex = cache["rgb"], cache["sigma"], cache["delta"], cache["ex"]
    alpha, x, T, w, pre = cache["alpha"], cache["x"], cache["T"], cache["w"], cache["pre"]
    dt = dimage.dtype.type
    g = dimage * ((pre >= dt(0.0)) & (pre <= dt(1.0))).astype(dimage.dtype)
    drgb = w[..., None] * g[..., None, :]
    dw = np.sum(rgb * g[..., None, :], axis=-1)
    if cache["white"]:
        dw = dw - np.sum(g, axis=-1)[..., None]
    dalpha = dw * T
    dT = dw * alpha
    prod = T * dT
    # reverse exclusive cumsum: Q_k = sum_{i>k} prod_i
    rev = np.cumsum(prod[..., ::-1], axis=-1, dtype=prod.dtype)[..., ::-1]
    Q = rev - prod
    dx = Q / x
    dalpha = dalpha - dx  # e = 1 - alpha
    dsigma = dalpha * delta * ex  # alpha = 1 - exp(-sigma*delta)
    return drgb, dsigma


def cdf_from_weights(weights):
    """utils.py:63-69: w += 1e-5; pdf = w / sum(w); cdf = [0, cumsum(pdf)]  (width S+1).

    Both sums run left to right (np.cumsum order).  TensorFlow's own summation order (Eigen packets on CPU, a
    parallel scan on GPU) is not written in the reference; the searchsorted / `denom < 1e-5` decisions downstream
    depend on the last bits, so the order is part of the oracle's DECLARED semantics and the HIP sampler follows it
    bit for bit."""
    dt = weights.dtype.type
    w = weights + dt(1e-5)
    total = np.cumsum(w, axis=-1, dtype=weights.dtype)[..., -1:]
    pdf = w / total
    cdf = np.cumsum(pdf, axis=-1, dtype=weights.dtype)
    return np.concatenate([np.zeros_like(cdf[..., :1]), cdf], axis=-1)


def fine_hierarchical_sampling_chunk(mid_points, weights, u, oob: str = "zero"):
    """utils.py:60-97 with ``u`` injected (the reference draws tf.random.uniform at utils.py:73).

    The reference passes ALL S coarse weights with S-1 mid-points (nerf.py:182-187), so gathers into
    ``mid_points`` can index S-1 or S, one/two past the end.  ``oob='zero'``: out-of-range gather yields 0
    (tf.gather on GPU); ``oob='clamp'``: index clamped to the last mid-point.  On TF-CPU the same input
    raises, i.e. wherever the TF-CPU path runs at all both modes agree.
    """
    dt = weights.dtype.type
    cdf = cdf_from_weights(weights)
    ncdf = cdf.shape[-1]
    nm = mid_points.shape[-1]
    # searchsorted(side='right') = number of cdf entries <= u
    idx = np.sum(cdf[..., None, :] <= u[..., :, None], axis=-1).astype(np.int32)
    below = np.maximum(0, idx - 1)
    above = np.minimum(ncdf - 1, idx)
    cdf_b = np.take_along_axis(cdf, below, axis=-1)
    cdf_a = np.take_along_axis(cdf, above, axis=-1)

    def gather_mid(ix):
        if oob == "clamp":
            return np.take_along_axis(mid_points, np.minimum(ix, nm - 1), axis=-1)
        if oob == "zero":
            ok = ix < nm
            v = np.take_along_axis(mid_points, np.minimum(ix, nm - 1), axis=-1)
            return np.where(ok, v, dt(0.0))
        raise ValueError(oob)

    m_b, m_a = gather_mid(below), gather_mid(above)
    denom = cdf_a - cdf_b
    denom = np.where(denom < dt(1e-5), dt(1.0), denom)
    tt = (u - cdf_b) / denom
    return m_b + tt * (m_a - m_b)


def fine_points(t_coarse, w_coarse, u, oob="zero"):
    """nerf.py:182-191: mids = 0.5(t[1:]+t[:-1]); inverse-CDF; sort(concat(coarse, fine))."""
    dt = t_coarse.dtype.type
    mids = dt(0.5) * (t_coarse[..., 1:] + t_coarse[..., :-1])
    tf_ = fine_hierarchical_sampling_chunk(mids, w_coarse, u, oob)
    return np.sort(np.concatenate([t_coarse, tf_], axis=-1), axis=-1)


# --------------------------------------------------------------------------------------
# NeRFMLP  (keras_nerf/model/nerf/mlp.py)
# --------------------------------------------------------------------------------------


# ``emulate_bf16`` values (a TEST AID, not reference behaviour -- the reference computes in fp32):
#   False    fp32 (or fp64) throughout: the reference's arithmetic
#   True     every Dense layer as written in mlp.py, matmul operands rounded to bf16, fp32 accumulate, fp32 bias and
#            activation (no kernel computes exactly this any more; kept as the layer-by-layer bf16 yardstick)
#   FUSED    the arithmetic of the HIP kernels (csrc/mlp_fwd.hip ... and csrc/generic.hip): as True for the trunk; the three
#            linear layers behind it and the sigma head evaluated as ONE affine map of (h7, dir_enc) with a composed
#            [283,4] matrix (see head_compose), and their gradients recovered from the sums M = [h7;dir]^T dz_rgb, s
FUSED = "fused"


def _rb(x):
    """a matmul operand as the MFMA kernels see it: rounded to bf16, held in fp32"""
    return round_bf16(x)                   # already a fresh fp32 array


def _mm(a, w, emulate_bf16):
    if emulate_bf16:
        return _rb(a) @ _rb(w)
    return a @ w


def head_compose(params, cfg: NerfConfig):
    """mlp.py:21-27,42-48: features and rgb_features are LINEAR Dense layers, so
        rgb_pre = h7 (W_f W_r1 W_c) + dir_enc (W_r2 W_c) + ((b_f W_r1 + b_r) W_c + b_c),   sigma_pre = h7 w_s + b_s
    (W_r1 = first dense_units rows of the rgb_features kernel, W_r2 = its dir rows).  h7 is the trunk output, dense_units
    wide -- or [h, xyz_enc] when the skip concat follows the LAST trunk layer (mlp.py:36-38), which the general-shape kernels
    support.  Returns H [trunk_width + dir_dim, 4] (columns r, g, b, sigma; the dir rows of the sigma column are zero) and the
    bias [4], in the dtype of the params."""
    n, U = cfg.n_layers, cfg.dense_units
    ks, bs, kf, bf, kr, br, kc, bc = params[2 * n:2 * n + 8]
    Tr = kf.shape[0]
    P = kr @ kc                                        # [U + dir_dim, 3]
    H = np.zeros((Tr + kr.shape[0] - U, 4), ks.dtype)    # rows of the rgb_features kernel behind `features`: the direction input's width (mlp.py:23-24, 44-46)
    H[:Tr, :3] = kf @ P[:U]
    H[Tr:, :3] = P[U:]
    H[:Tr, 3] = ks[:, 0]
    hb = np.concatenate([bf @ P[:U] + br @ kc + bc, bs])
    return H, hb.astype(ks.dtype)


def mlp_forward(params: Sequence[np.ndarray], xyz_enc, dir_enc, cfg: NerfConfig, emulate_bf16=False,
                want_cache=False):
    """mlp.py:29-50.  Dense on rank-3 input contracts the last axis.

    trunk: relu Dense x n_layers, concat [h, xyz_enc] after layer i when i % skip == 0 and i > 0;
    sigma = relu(Dense(1)); features = Dense(units) (linear); concat [features, dir_enc];
    rgb_features = Dense(units/2) (LINEAR in this reference); rgb = sigmoid(Dense(3)).

    ``emulate_bf16`` rounds both matmul operands to bf16 (fp32 accumulate, fp32 bias/activation) the way the
    MFMA kernels do; it is a test aid, not reference behaviour.
    """
    dt = xyz_enc.dtype.type
    lead = xyz_enc.shape[:-1]
    x = xyz_enc.reshape(-1, xyz_enc.shape[-1])
    dd = dir_enc.reshape(-1, dir_enc.shape[-1])
    h = x
    ins, outs = [], []
    p = 0
    for i in range(cfg.n_layers):
        k, b = params[p], params[p + 1]; p += 2
        ins.append(h)
        z = _mm(h, k, emulate_bf16) + b
        h = np.maximum(z, dt(0.0))
        outs.append(h)
        if i % cfg.skip_layer == 0 and i > 0:
            h = np.concatenate([h, x], axis=-1)
    trunk = h
    if emulate_bf16 == FUSED:
        H, hb = head_compose(params, cfg)
        z4 = _mm(np.concatenate([trunk, dd], axis=-1), H, True) + hb
        sigma = np.maximum(z4[:, 3:4], dt(0.0))
        with np.errstate(over="ignore"):
            rgb = dt(1.0) / (dt(1.0) + np.exp(-z4[:, :3]))
        if want_cache:
            cache = dict(ins=ins, outs=outs, trunk=trunk, sigma=sigma, rgb=rgb, x=x, dd=dd, H=H, emulate_bf16=emulate_bf16)
            return rgb.reshape(lead + (3,)), sigma.reshape(lead + (1,)), cache
        return rgb.reshape(lead + (3,)), sigma.reshape(lead + (1,))
    ks, bs = params[p], params[p + 1]; p += 2
    kf, bf = params[p], params[p + 1]; p += 2
    kr, br = params[p], params[p + 1]; p += 2
    kc, bc = params[p], params[p + 1]; p += 2
    zs = _mm(trunk, ks, emulate_bf16) + bs
    sigma = np.maximum(zs, dt(0.0))
    feat = _mm(trunk, kf, emulate_bf16) + bf
    fcat = np.concatenate([feat, dd], axis=-1)
    f2 = _mm(fcat, kr, emulate_bf16) + br
    zc = _mm(f2, kc, emulate_bf16) + bc
    with np.errstate(over="ignore"):
        rgb = dt(1.0) / (dt(1.0) + np.exp(-zc))
    rgb_o = rgb.reshape(lead + (3,))
    sigma_o = sigma.reshape(lead + (1,))
    if want_cache:
        cache = dict(ins=ins, outs=outs, trunk=trunk, sigma=sigma, fcat=fcat, f2=f2, rgb=rgb, x=x,
                     emulate_bf16=emulate_bf16)
        return rgb_o, sigma_o, cache
    return rgb_o, sigma_o


def mlp_backward(params, cache, drgb, dsigma, cfg: NerfConfig):
    """Hand-written backward of mlp_forward: returns the 24 gradient tensors in trainable_variables order.

    relu'(z) = [out > 0] (TF ReluGrad), sigmoid' = y(1-y).  No gradient is taken wrt inputs (nerf.py:361-377
    watches only the MLP's trainable variables).
    """
    eb = cache["emulate_bf16"]
    dt = drgb.dtype.type
    drgb = drgb.reshape(-1, 3)
    dsigma = dsigma.reshape(-1, 1)
    n = cfg.n_layers
    ks, kf, kr, kc = params[2 * n], params[2 * n + 2], params[2 * n + 4], params[2 * n + 6]

    def mmT(a, b):  # a^T b  (wgrad)
        if eb:
            return _rb(a).T @ _rb(b)
        return a.T @ b

    def mmW(g, w):  # g w^T (dgrad)
        if eb:
            return _rb(g) @ _rb(w).T
        return g @ w.T

    rgb = cache["rgb"]
    dzc = drgb * rgb * (dt(1.0) - rgb)
    if eb == FUSED:
        # the fused kernels: dz4 = (dz_rgb, dz_sigma) in bf16; M = [h7 ; dir]^T dz_rgb, s = sum dz_rgb (bf16 operands, fp32
        # sums); the six head gradients by the chain rule through the three linear layers, in fp32 on the master weights
        U = cfg.dense_units
        kf, bf_, kr, br, kc = params[2 * n + 2], params[2 * n + 3], params[2 * n + 4], params[2 * n + 5], params[2 * n + 6]
        Tr = kf.shape[0]                                             # trunk width: U, or U + xyz_dim behind a final skip concat
        dzs = dsigma * (cache["sigma"] > 0).astype(dsigma.dtype)
        dz4 = np.concatenate([dzc, dzs], axis=-1)
        hd = np.concatenate([cache["trunk"], cache["dd"]], axis=-1)
        M4 = mmT(hd, dz4)                                            # [Tr + dir_dim, 4]
        s4 = _rb(dz4).sum(0)
        M1, M2, s3 = M4[:Tr, :3], M4[Tr:, :3], s4[:3]
        P1 = kr[:U] @ kc
        Q = kf.T @ M1 + np.outer(bf_, s3)                            # = sum_s features[s]^T dz_rgb[s]
        g_kc = kr[:U].T @ Q + kr[U:].T @ M2 + np.outer(br, s3); g_bc = s3
        g_kr = np.concatenate([Q, M2], axis=0) @ kc.T; g_br = s3 @ kc.T
        g_kf = M1 @ P1.T; g_bf = s3 @ P1.T
        g_ks, g_bs = M4[:Tr, 3:4], s4[3:4]
        dh = mmW(dz4, cache["H"][:Tr])
        grads_trunk = [None] * (2 * n)
        for i in reversed(range(n)):
            if i % cfg.skip_layer == 0 and i > 0:
                dh = dh[:, :cfg.dense_units]
            dz = dh * (cache["outs"][i] > 0).astype(dh.dtype)
            grads_trunk[2 * i] = mmT(cache["ins"][i], dz)
            grads_trunk[2 * i + 1] = _rb(dz).sum(0)
            if i > 0:
                dh = mmW(dz, params[2 * i])
        return grads_trunk + [g_ks, g_bs, g_kf, g_bf, g_kr, g_br, g_kc, g_bc]
    g_kc, g_bc = mmT(cache["f2"], dzc), dzc.sum(0)
    df2 = mmW(dzc, kc)
    g_kr, g_br = mmT(cache["fcat"], df2), df2.sum(0)
    dfcat = mmW(df2, kr)
    dfeat = dfcat[:, :cfg.dense_units]
    g_kf, g_bf = mmT(cache["trunk"], dfeat), dfeat.sum(0)
    dzs = dsigma * (cache["sigma"] > 0).astype(dsigma.dtype)
    g_ks, g_bs = mmT(cache["trunk"], dzs), dzs.sum(0)
    dh = mmW(dfeat, kf) + mmW(dzs, ks)
    grads_trunk = [None] * (2 * n)
    for i in reversed(range(n)):
        if i % cfg.skip_layer == 0 and i > 0:
            dh = dh[:, :cfg.dense_units]  # drop the xyz_enc part of the concat (inputs carry no gradient)
        dz = dh * (cache["outs"][i] > 0).astype(dh.dtype)
        grads_trunk[2 * i] = mmT(cache["ins"][i], dz)
        grads_trunk[2 * i + 1] = dz.sum(0)
        if i > 0:
            dh = mmW(dz, params[2 * i])
    return grads_trunk + [g_ks, g_bs, g_kf, g_bf, g_kr, g_br, g_kc, g_bc]


# --------------------------------------------------------------------------------------
# NeRF chunk forward / train step  (keras_nerf/model/nerf/nerf.py)
# --------------------------------------------------------------------------------------


def mse(y, yhat):
    """Keras MeanSquaredError / train.py:130-136: mean over all R*3 elements."""
    d = y - yhat
    return np.mean(d * d, dtype=y.dtype)


def predict_and_render_chunk_single(params, o, d, t, cfg, white_background, emulate_bf16=False, want_cache=False):
    """nerf.py:175-216 for an already chosen set of t-values (coarse t, or the merged fine t)."""
    xyz, dire = encode_position_and_directions(o, d, t, cfg.pos_emb_xyz, cfg.pos_emb_dir)
    if want_cache:
        rgb, sigma, mc = mlp_forward(params, xyz, dire, cfg, emulate_bf16, True)
        image, depth, w, rc = render_image_depth_chunk(rgb, sigma, t, white_background, want_cache=True)
        return dict(image=image, depth=depth, weights=w, t=t, rgb=rgb, sigma=sigma), (mc, rc)
    rgb, sigma = mlp_forward(params, xyz, dire, cfg, emulate_bf16)
    image, depth, w = render_image_depth_chunk(rgb, sigma, t, white_background)
    return dict(image=image, depth=depth, weights=w, t=t, rgb=rgb, sigma=sigma)


def predict_and_render_chunk(coarse_params, fine_params, o, d, t, u, cfg, white_background, oob="zero",
                             emulate_bf16=False):
    """nerf.py:218-227: coarse pass, then fine pass fed with the coarse weights."""
    c = predict_and_render_chunk_single(coarse_params, o, d, t, cfg, white_background, emulate_bf16)
    tf_ = fine_points(t, c["weights"], u, oob)
    f = predict_and_render_chunk_single(fine_params, o, d, tf_, cfg, white_background, emulate_bf16)
    return c, f


def predict_and_render_images(coarse_params, fine_params, o, d, t, u, cfg, ray_chunks, white_background,
                              oob="zero", emulate_bf16=False):
    """nerf.py:229-304: flatten [B,H,W,.] -> [N,.], loop over N/ray_chunks chunks, stitch."""
    lead = o.shape[:-1]
    N = int(np.prod(lead))
    R = min(ray_chunks, N)
    assert N % R == 0, f"ray_chunks {R} must be a divisor of the number of rays {N}"  # nerf.py:100
    of, df, tf_, uf = o.reshape(N, 3), d.reshape(N, 3), t.reshape(N, -1), u.reshape(N, -1)
    cs, fs = [], []
    for i in range(N // R):
        sl = slice(i * R, (i + 1) * R)
        c, f = predict_and_render_chunk(coarse_params, fine_params, of[sl], df[sl], tf_[sl], uf[sl], cfg,
                                        white_background, oob, emulate_bf16)
        cs.append(c); fs.append(f)

    def stitch(parts):
        return {k: np.concatenate([p[k] for p in parts], 0).reshape(lead + parts[0][k].shape[1:])
                for k in ("image", "depth", "weights", "t")}
    return stitch(cs), stitch(fs)


def chunk_loss_and_grads(params, o, d, t, target, cfg, white_background, emulate_bf16=False):
    """One GradientTape block of nerf.py:361-377 / 390-406: forward, MSE, gradients wrt the 24 tensors."""
    res, (mc, rc) = predict_and_render_chunk_single(params, o, d, t, cfg, white_background, emulate_bf16, True)
    img = res["image"]
    loss = mse(target, img)
    dimg = (img.dtype.type(2.0) / img.dtype.type(img.size)) * (img - target)
    drgb, dsigma = render_backward(rc, dimg)
    grads = mlp_backward(params, mc, drgb, dsigma, cfg)
    return res, loss, grads


class KerasAdam:
    """tf.keras.optimizers.get('adam') (nerf.py:163-165): lr 1e-3, b1 .9, b2 .999, eps 1e-7, Keras form

        m <- b1 m + (1-b1) g ;  v <- b2 v + (1-b2) g^2
        theta <- theta - lr * sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps)
    """

    def __init__(self, params, lr=1e-3, b1=0.9, b2=0.999, eps=1e-7):
        self.lr, self.b1, self.b2, self.eps = lr, b1, b2, eps
        self.m = [np.zeros_like(p) for p in params]
        self.v = [np.zeros_like(p) for p in params]
        self.t = 0

    def apply(self, params, grads):
        self.t += 1
        dt = params[0].dtype.type
        lr_t = dt(self.lr * np.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t))
        for i, (p, g) in enumerate(zip(params, grads)):
            self.m[i] = dt(self.b1) * self.m[i] + dt(1 - self.b1) * g
            self.v[i] = dt(self.b2) * self.v[i] + dt(1 - self.b2) * (g * g)
            p -= lr_t * self.m[i] / (np.sqrt(self.v[i]) + dt(self.eps))


def train_step(coarse_params, fine_params, opt_c: Optional[KerasAdam], opt_f: Optional[KerasAdam], images, o, d, t,
               u, cfg, ray_chunks, white_background, oob="zero", emulate_bf16=False, grad_scale_hook=None):
    """nerf.py:332-473.  One optimizer step over B whole images.

    Per chunk: coarse tape (loss, grads), accumulate grads/C; fine tape fed with the coarse weights as a
    constant (nerf.py:390-398), accumulate grads/C.  Then both Adams, then accumulators are zeroed.
    ``grad_scale_hook(gc, gf)`` lets the data-parallel test insert the cross-replica SUM (train.py:75).
    Returns (metrics dict, coarse image, fine image, accumulated grads).
    """
    images = images[..., :3]
    lead = o.shape[:-1]
    N = int(np.prod(lead))
    R = min(ray_chunks, N)
    assert N % R == 0, f"ray_chunks {R} must be a divisor of the number of rays {N}"
    C = N // R
    dt = o.dtype.type
    im, of, df, tf_, uf = images.reshape(N, 3), o.reshape(N, 3), d.reshape(N, 3), t.reshape(N, -1), u.reshape(N, -1)
    acc_c = [np.zeros_like(p) for p in coarse_params]
    acc_f = [np.zeros_like(p) for p in fine_params]
    loss_c = dt(0); loss_f = dt(0)
    img_c, img_f = [], []
    for i in range(C):
        sl = slice(i * R, (i + 1) * R)
        rc, lc, gc = chunk_loss_and_grads(coarse_params, of[sl], df[sl], tf_[sl], im[sl], cfg, white_background,
                                          emulate_bf16)
        for a, g in zip(acc_c, gc):
            assert np.all(np.isfinite(g)), "Coarse Gradient is not finite"  # nerf.py:381-382
            a += g / dt(C)
        loss_c += lc / dt(C)
        tfine = fine_points(tf_[sl], rc["weights"], uf[sl], oob)
        rf, lf, gf = chunk_loss_and_grads(fine_params, of[sl], df[sl], tfine, im[sl], cfg, white_background,
                                          emulate_bf16)
        for a, g in zip(acc_f, gf):
            assert np.all(np.isfinite(g)), "Fine Gradient is not finite"  # nerf.py:410-411
            a += g / dt(C)
        loss_f += lf / dt(C)
        img_c.append(rc["image"]); img_f.append(rf["image"])
    if grad_scale_hook is not None:
        acc_c, acc_f = grad_scale_hook(acc_c, acc_f)
    if opt_c is not None:
        opt_c.apply(coarse_params, acc_c)
        opt_f.apply(fine_params, acc_f)
    ci = np.concatenate(img_c, 0).reshape(lead + (3,))
    fi = np.concatenate(img_f, 0).reshape(lead + (3,))
    return dict(coarse_loss=loss_c, fine_loss=loss_f), ci, fi, (acc_c, acc_f)


# --------------------------------------------------------------------------------------
# metrics (tf.image.psnr semantics; nerf.py:306-312)
# --------------------------------------------------------------------------------------


def psnr(a, b, max_val=1.0):
    """tf.image.psnr per image over the last three axes: 20 log10(max) - 10 log10(mse)."""
    m = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2, axis=(-3, -2, -1))
    return 20.0 * np.log10(max_val) - 10.0 * np.log10(m)


def ssim(a, b, max_val=1.0, filter_size=11, filter_sigma=1.5, k1=0.01, k2=0.03):
    """tf.image.ssim with its defaults as called at nerf.py:313-321 (TF documentation semantics, not runnable here):
    Gaussian window 11x11 sigma 1.5, VALID positions, per channel luminance x contrast-structure, mean over windows and
    channels.  a, b [B,H,W,C]; float64 arithmetic."""
    a = a.astype(np.float64); b = b.astype(np.float64)
    x = np.arange(filter_size, dtype=np.float64) - (filter_size - 1) / 2.0
    g = np.exp(-(x ** 2) / (2 * filter_sigma ** 2)); g /= g.sum()
    w = np.outer(g, g)
    win = lambda z: np.lib.stride_tricks.sliding_window_view(z, (filter_size, filter_size), axis=(1, 2))   # [B,h,w,C,f,f]
    conv = lambda z: np.einsum("bhwcij,ij->bhwc", win(z), w)
    c1, c2 = (k1 * max_val) ** 2, (k2 * max_val) ** 2
    mx, my = conv(a), conv(b)
    sxx, syy, sxy = conv(a * a) - mx * mx, conv(b * b) - my * my, conv(a * b) - mx * my
    lum = (2 * mx * my + c1) / (mx * mx + my * my + c1)
    cs = (2 * sxy + c2) / (sxx + syy + c2)
    return np.mean(lum * cs, axis=(1, 2, 3))


# --------------------------------------------------------------------------------------
# data side: rays and poses  (keras_nerf/data/rays.py, keras_nerf/data/utils.py)
# --------------------------------------------------------------------------------------


def get_focal_from_fov(fov: float, width: int) -> float:
    """data/utils.py:5-16: 0.5 * width / tan(0.5 * fov) (float32 arithmetic in the reference)."""
    return float(np.float32(0.5) * np.float32(width) / np.tan(np.float32(0.5) * np.float32(fov)))


def pose_spherical(theta: float, phi: float, t: float) -> np.ndarray:
    """data/utils.py:19-63: flip @ rot_theta(theta deg) @ rot_phi(phi deg) @ trans_t(t), float32 4x4."""
    f = np.float32
    tr = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, t], [0, 0, 0, 1]], f)
    ph = f(phi / 180.0 * np.pi)
    rp = np.array([[1, 0, 0, 0], [0, np.cos(ph), -np.sin(ph), 0], [0, np.sin(ph), np.cos(ph), 0], [0, 0, 0, 1]], f)
    th = f(theta / 180.0 * np.pi)
    rt = np.array([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0], [np.sin(th), 0, np.cos(th), 0], [0, 0, 0, 1]], f)
    flip = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], f)
    return (flip @ (rt @ (rp @ tr))).astype(f)


def generate_rays(c2w, focal, W, H, near, far, n_sample, noise):
    """data/rays.py:69-130 with the uniform ``noise`` in [0,1) injected (shape [H,W,n] — the reference draws
    it as [W,H,n], rays.py:122-123, which only broadcasts for square images).

    x,y pixel-corner grid, camera vector (x-W/2)/f, -(y-H/2)/f, -1; d = sum(cam[...,None,:] * R, -1), normalised;
    o = c2w[:3,3]; t = clip(linspace(near,far,n) + noise*interval - interval/2, near, far).
    """
    f = np.float32
    x, y = np.meshgrid(np.arange(W, dtype=f), np.arange(H, dtype=f), indexing="xy")
    xc = (x - f(W) * f(0.5)) / f(focal)
    yc = (y - f(H) * f(0.5)) / f(focal)
    cam = np.stack([xc, -yc, -np.ones_like(xc)], axis=-1)
    R = c2w[:3, :3].astype(f)
    tr = c2w[:3, -1].astype(f)
    dvec = np.sum(cam[..., None, :] * R, axis=-1)
    dvec = dvec / np.linalg.norm(dvec, axis=-1, keepdims=True)
    o = np.broadcast_to(tr, dvec.shape).astype(f)
    tv = np.linspace(f(near), f(far), n_sample, dtype=f)
    interval = f((far - near) / n_sample)
    t = tv + noise.astype(f) * interval - interval / f(2)
    t = np.clip(t, f(near), f(far)).astype(f)
    return o.copy(), dvec.astype(f), t


# --------------------------------------------------------------------------------------
# Philox4x32-10 (the build's own in-kernel RNG for ``u``; restated here so that the RNG mode can be parity
# tested too.  Not reference behaviour: the reference uses tf.random.uniform.)
# --------------------------------------------------------------------------------------

_PH_M0, _PH_M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_PH_W0, _PH_W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32(counter: np.ndarray, key: np.ndarray) -> np.ndarray:
    """counter [...,4] uint32, key [...,2] uint32 -> [...,4] uint32 (10 rounds)."""
    c = [counter[..., i].astype(np.uint32) for i in range(4)]
    k0 = key[..., 0].astype(np.uint32); k1 = key[..., 1].astype(np.uint32)
    for _ in range(10):
        p0 = c[0].astype(np.uint64) * _PH_M0
        p1 = c[2].astype(np.uint64) * _PH_M1
        hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
        hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
        c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
        with np.errstate(over="ignore"):
            k0 = (k0 + _PH_W0).astype(np.uint32); k1 = (k1 + _PH_W1).astype(np.uint32)
    return np.stack(c, axis=-1)


def philox_uniform_u(seed: int, stream: int, ray_index: np.ndarray, n_fine: int) -> np.ndarray:
    """u[r, j] in [0,1): counter = (j//4, ray, stream, 0), key = (seed_lo, seed_hi); lane j%4;
    u = (x >> 8) * 2^-24  (24-bit mantissa, never reaches 1.0)."""
    ray_index = np.asarray(ray_index, np.uint32)
    nblk = (n_fine + 3) // 4
    ctr = np.zeros(ray_index.shape + (nblk, 4), np.uint32)
    ctr[..., 0] = np.arange(nblk, dtype=np.uint32)
    ctr[..., 1] = ray_index[..., None]
    ctr[..., 2] = np.uint32(stream & 0xFFFFFFFF)
    key = np.zeros(ray_index.shape + (nblk, 2), np.uint32)
    key[..., 0] = np.uint32(seed & 0xFFFFFFFF)
    key[..., 1] = np.uint32((seed >> 32) & 0xFFFFFFFF)
    x = philox4x32(ctr, key).reshape(ray_index.shape + (nblk * 4,))[..., :n_fine]
    return ((x >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)
