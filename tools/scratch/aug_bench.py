#!/usr/bin/env python
"""Diagnostic: ms per batch of the device augmentation chain at the benchmarked shape (2 x 4 x 205^3 raw -> 2 x 4 x 128^3),
worst case (every transform forced on) and as drawn (the reference's probabilities)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from e2enet_medical_amd.training.data_augmentation.data_augmentation_moreDA import DeviceAugmenter
from e2enet_medical_amd.training.data_augmentation.default_data_augmentation import default_3D_augmentation_params, get_patch_size
p = dict(default_3D_augmentation_params)
rot = (-30. / 360 * 2. * np.pi, 30. / 360 * 2. * np.pi)
p.update(do_elastic=False, rotation_x=rot, rotation_y=rot, rotation_z=rot, selected_seg_channels=[0])
raw = tuple(int(v) for v in get_patch_size((128, 128, 128), rot, rot, rot, (0.85, 1.25)))
p["scale_range"] = (0.7, 1.4)
scales = [[1, 1, 1], [0.5] * 3, [0.25] * 3, [0.125] * 3]
a = DeviceAugmenter((128, 128, 128), p, deep_supervision_scales=scales, seed=0)
data = torch.randn((2, 4) + raw, device="cuda")
seg = torch.randint(0, 4, (2, 1) + raw, device="cuda").float()
print("raw patch", raw)
for _ in range(3):
    a(data, seg)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 40
for _ in range(n):
    a(data, seg)
torch.cuda.synchronize()
print("as drawn: %.2f ms / batch" % ((time.perf_counter() - t0) / n * 1e3))
d = a._draw(2, 4, raw)
from e2enet_medical_amd.training.data_augmentation.default_data_augmentation import rotation_matrix_3d
for b in range(2):
    A = rotation_matrix_3d(0.3, -0.2, 0.4).T * 1.2
    d["mat"][b] = np.concatenate([A, (np.array(raw) / 2. - 0.5)[:, None]], 1).reshape(-1)
d["modified"][:] = True
d.update(noise=np.full((2, 4), 0.05), blur=np.full((2, 4), 0.8), mul=np.full((2, 4), 1.1), contrast=np.full((2, 4), 0.9),
         zoom=np.full((2, 4), 0.7), gamma_inv=np.full((2, 4), 0.8), gamma=np.full((2, 4), 1.2), mirror=np.ones((2, 3), dtype=bool))
for _ in range(2):
    a.apply(data, seg, d)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    a.apply(data, seg, d)
torch.cuda.synchronize()
print("every transform on: %.2f ms / batch" % ((time.perf_counter() - t0) / 10 * 1e3))
