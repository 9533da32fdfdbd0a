#!/usr/bin/env python3
"""A training script with the SHAPE of the reference's multi-GPU train.py (train.py:75-157: strategy -> global batch -> loader ->
monitor -> model under strategy.scope() -> fit -> save_model), written against this implementation for
tests/test_gpu_api.py::test_train_script_shaped_like_the_reference_uses_every_rank.  Started as plain `python train_like_reference.py ...`:
`parallel.MirroredStrategy()` turns the process into the launcher of one rank per GPU (here: --devices ranks; the test shares one GPU
over gloo) and every rank runs the rest.  Each rank leaves an exact checksum of its weights so that the test can see the mirrored
variables stayed mirrored."""
import argparse
import json
import logging
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from keras_nerf.data.loader import DatasetLoader                      # noqa: E402  -- the reference's import lines (train.py:6-8):
from keras_nerf.model.nerf.callback import NeRFTrainMonitor           # noqa: E402     the alias package resolves them to keras_nerf_amd
from keras_nerf.model.nerf.nerf import NeRF                           # noqa: E402
from keras_nerf_amd import parallel                                   # noqa: E402  -- instead of `import tensorflow as tf`


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data_dir", required=True)
    ap.add_argument("--log_dir", required=True)
    ap.add_argument("--model_dirs", required=True)
    ap.add_argument("--name", default="scene")
    ap.add_argument("--img_wh", type=int, default=16)
    ap.add_argument("--batch_size", type=int, default=1)
    ap.add_argument("--ray_chunks", type=int, default=128)
    ap.add_argument("--num_epochs", type=int, default=2)
    ap.add_argument("--devices", type=int, default=None)
    args = ap.parse_args()
    logging.basicConfig(level=logging.INFO)

    strategy = parallel.MirroredStrategy(devices=args.devices)         # train.py:75
    print("Number of devices: {}".format(strategy.num_replicas_in_sync))
    loader = DatasetLoader(args.data_dir, True)
    global_batch_size = args.batch_size * strategy.num_replicas_in_sync          # train.py:84
    train_dataset, val_dataset, test_dataset = loader.load_dataset(batch_size=global_batch_size, image_width=args.img_wh, image_height=args.img_wh,
                                                                   near=2.0, far=6.0, n_sample=64)
    monitor = NeRFTrainMonitor(dataset=test_dataset, log_dir=os.path.join(args.log_dir, args.name), batch_size=args.batch_size, update_freq=1)
    with strategy.scope():                                            # train.py:110
        nerf = NeRF(n_coarse=64, n_fine=128, pos_emb_xyz=10, pos_emb_dir=4, n_layers=8, dense_units=256, skip_layer=4, model_path=None,
                    seed=100 + parallel.rank())                       # own initial weights per rank: compile() must mirror rank 0's
        nerf.compile(optimizer="adam", loss="mse", batch_size=args.batch_size, image_width=args.img_wh, image_height=args.img_wh,
                     ray_chunks=args.ray_chunks, white_background=True)
    history = nerf.fit(train_dataset, epochs=args.num_epochs, validation_data=val_dataset, callbacks=[monitor], initial_epoch=monitor.last_epoch)
    if parallel.is_main():
        os.makedirs(args.model_dirs, exist_ok=True)
        nerf.save_model(os.path.join(args.model_dirs, args.name))      # train.py:152-155
    import numpy as np
    w = np.concatenate([nerf.coarse.get_flat_weights(), nerf.fine.get_flat_weights()])
    chk = int((w.view(np.uint32).astype(np.uint64) * (np.arange(w.size, dtype=np.uint64) % 8191 + 1)).sum())
    with open(os.path.join(args.log_dir, f"rank{parallel.rank()}.json"), "w") as f:
        json.dump({"rank": parallel.rank(), "world": strategy.num_replicas_in_sync, "weight_checksum": chk, "steps_per_epoch": len(train_dataset),
                   "images_per_step": int(next(iter(train_dataset))[0].shape[0]), "history": {k: [float(x) for x in v] for k, v in history.items()}}, f)
    parallel.barrier()


if __name__ == "__main__":
    main()
