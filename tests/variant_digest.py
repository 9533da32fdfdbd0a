"""Helper of tests/test_gpu_variants.py: run with KNERF_LIB pointing at a build variant; prints SHA-1 digests of one render chunk's
outputs and of every saved block / dZ of one training chunk at bench size (fixed inputs, injected u)."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def digest(t):
    return hashlib.sha1(t.contiguous().view(torch.uint8).cpu().numpy().tobytes()).hexdigest()


def main():
    from keras_nerf_amd.data.utils import get_focal_from_fov, pose_spherical
    from keras_nerf_amd.debug import debug_buffer
    from keras_nerf_amd.model.nerf.mlp import NeRFMLP
    from keras_nerf_amd.runtime import KnerfContext
    ctx = KnerfContext(white_background=True, options=dict(deterministic=1))
    for net in (0, 1):
        m = NeRFMLP(seed=net); m.build(); ctx.set_weights(net, m.get_flat_weights() * 1.5)
    o, d, t = ctx.generate_rays(pose_spherical(33.0, -30.0, 4.0)[None], get_focal_from_fov(0.6911112070083618, 64), 64, 64, 2.0, 6.0, 64, None, seed=3)
    o, d, t = o.reshape(-1, 3), d.reshape(-1, 3), t.reshape(-1, 64)
    g = torch.Generator(device="cuda").manual_seed(1)
    tgt = torch.rand((4096, 3), device="cuda", generator=g); u = torch.rand((4096, 128), device="cuda", generator=g)
    out = {k: digest(v) for k, v in ctx.render_chunk(o, d, t, u).items()}
    ctx.zero_grads()
    ctx.train_chunk(o, d, t, tgt, u)
    torch.cuda.synchronize()
    n_tiles = 4096 * 192 // 32
    for name, which, stride in () if os.environ.get("KNERF_DIGEST_NO_BUFFERS") else (("act", 0, 118 * 1024 + 256), ("mask", 1, 8 * 1024 + 256), ("dz", 2, 130 * 1024 + 256)):
        out[name] = digest(debug_buffer(ctx, which)[:n_tiles * stride])
    out["grads"] = digest(ctx.grads_view())          # deterministic mode: a function of the kernels' arithmetic only
    print(json.dumps(out))


if __name__ == "__main__":
    main()
