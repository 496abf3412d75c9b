"""Golden camera batches from the REFERENCE's own datasets (build container only):
  threestudio/data/uncond.py           RandomCameraIterableDataset.collate           (asd_sd_nerf.yaml data block)
  threestudio/data/uncond_multiview.py RandomMultiviewCameraIterableDataset.collate  (asd_mv_nerf.yaml data block)
Both are seeded with torch.manual_seed / random.seed so the product datamodules, which draw in the same order, must
reproduce every key.    python tests/golden/make_goldens_camera.py  ->  tests/golden/camera_{sv,mv}.npz
"""
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as H  # noqa: E402

H.install()
H._mod("cv2")
import pytorch_lightning as pl  # noqa: E402  (stub)

pl.LightningDataModule = object

SV = dict(batch_size=[2, 1], width=[16, 32], height=[16, 32], resolution_milestones=[10000], camera_distance_range=[1.0, 1.5],
          fovy_range=[40, 70], elevation_range=[-10, 45], camera_perturb=0.0, center_perturb=0.0, up_perturb=0.0,
          eval_camera_distance=1.2, eval_fovy_deg=70.0, n_val_views=30)
MV = dict(batch_size=[8, 4], n_view=4, width=[16, 32], height=[16, 32], resolution_milestones=[10000],
          camera_distance_range=[0.8, 1.0], fovy_range=[15, 60], elevation_range=[0, 30], camera_perturb=0.0, center_perturb=0.0,
          up_perturb=0.0, eval_camera_distance=3.0, eval_fovy_deg=40.0, n_val_views=30)
KEYS = ["rays_o", "rays_d", "mvp_mtx", "camera_positions", "c2w", "light_positions", "elevation", "azimuth", "camera_distances", "fovy"]


def run(ds_cls, cfg_cls, cfg, seeds, extra=None):
    out = {}
    for s in seeds:
        c = dict(cfg)
        c.update(extra or {})
        ds = ds_cls(cfg_cls(**c))
        torch.manual_seed(s)
        random.seed(s)
        b = ds.collate(None)
        for k in KEYS:
            out[f"s{s}.{k}"] = b[k].numpy()
    return out


if __name__ == "__main__":
    from threestudio.data.uncond import RandomCameraDataModuleConfig, RandomCameraIterableDataset
    from threestudio.data.uncond_multiview import RandomMultiviewCameraDataModuleConfig, RandomMultiviewCameraIterableDataset

    seeds = [0, 1, 2, 3]  # both branches of the 50/50 elevation rule occur among these
    np.savez_compressed(os.path.join(HERE, "camera_sv.npz"), seeds=seeds, **run(RandomCameraIterableDataset, RandomCameraDataModuleConfig, SV, seeds))
    np.savez_compressed(os.path.join(HERE, "camera_mv.npz"), seeds=seeds,
                        **run(RandomMultiviewCameraIterableDataset, RandomMultiviewCameraDataModuleConfig, MV, seeds))
    np.savez_compressed(os.path.join(HERE, "camera_mv_magic3d.npz"), seeds=seeds,
                        **run(RandomMultiviewCameraIterableDataset, RandomMultiviewCameraDataModuleConfig, MV, seeds,
                              dict(light_sample_strategy="magic3d", camera_perturb=0.1, center_perturb=0.2, up_perturb=0.02, zoom_range=[0.8, 1.0])))
    print("camera goldens written")
