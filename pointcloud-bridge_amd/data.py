"""Batch preparation on the device (the caller side in FRONT of the hot path).

The reference prepares every sample in DataLoader workers on the host, utils/simpdataset.py:107-193
(`SimplePointCloudDataset.__getitem__`): draw `num_points` indices (`np.random.choice`, :136-142),
gather points / colours / labels, centre and scale into the unit ball (`normalize_points`, :47-63),
augment (`apply_transform`, :65-100: rotation about z, isotropic scale, translation, colour jitter).
At 10^7 points/s a training step consumes a batch of 16 x 16384 points every 10 ms, more than six
such workers deliver; here the same per-sample operations run batched on the GPU on raw scenes that
are already resident (tensors of different lengths), producing the trainer's batch dict
(`points`, `colors`, `labels`, plus `original_points`, `indices` like the reference's item).

Plain torch ops (gathers and elementwise work on [B,N,3]); no kernel of its own, device-agnostic.
The random draws come from a torch.Generator instead of numpy's global state: same distributions,
different streams -- so parity is on the deterministic parts (normalisation, the transform for given
draws, the index-set properties), checked against a numpy restatement in tests/test_data_cpu.py.
The reference module itself cannot be imported here (it imports h5py at load time).
"""
import math

import torch


def subsample_indices(num_available, num_points, generator=None, device="cpu"):
    """simpdataset.py:136-142: `num_points` indices into a scene of `num_available` points -- a
    random subset without replacement if there are enough, otherwise every point once plus random
    repeats, shuffled."""
    if num_available <= 0:
        raise ValueError("scene without points")
    if num_available >= num_points:
        return torch.randperm(num_available, generator=generator, device=device)[:num_points]
    extra = torch.randint(0, num_available, (num_points - num_available,), generator=generator, device=device)
    idx = torch.cat([torch.arange(num_available, device=device), extra])
    return idx[torch.randperm(num_points, generator=generator, device=device)]


def normalize_points(points):
    """simpdataset.py:47-63 per scene: points [B,N,3] -> centred, divided by the largest distance
    from the centroid (left centred if that distance is <= 1e-6)."""
    centred = points - points.mean(dim=1, keepdim=True)
    radius = centred.pow(2).sum(dim=-1).sqrt().amax(dim=1).view(-1, 1, 1)
    return torch.where(radius > 1e-6, centred / radius.clamp_min(1e-30), centred)


def apply_transform(points, colors, theta, scale, translation, color_noise=None):
    """simpdataset.py:65-100 for given draws: theta [B] (rotation about z, row vectors times R),
    scale [B], translation [B,3], color_noise like colors or None."""
    c, s = torch.cos(theta), torch.sin(theta)
    zero, one = torch.zeros_like(c), torch.ones_like(c)
    rot = torch.stack([torch.stack([c, -s, zero], dim=-1), torch.stack([s, c, zero], dim=-1),
                       torch.stack([zero, zero, one], dim=-1)], dim=-2).to(points.dtype)   # :77-82
    out = torch.bmm(points, rot) * scale.view(-1, 1, 1).to(points.dtype)                    # :82, :86
    out = out + translation.view(-1, 1, 3).to(points.dtype)                                 # :90
    if colors is not None and color_noise is not None:
        colors = (colors + color_noise.to(colors.dtype)).clamp(0, 1)                        # :94-95
    return out, colors


def random_transform(points, colors, generator=None):
    """apply_transform with the reference's distributions (:74, :85, :89, :94), one draw per scene."""
    B, dev = points.shape[0], points.device

    def uniform(lo, hi, *shape):
        return torch.rand(*shape, generator=generator, device=dev) * (hi - lo) + lo

    noise = None if colors is None else torch.randn(colors.shape, generator=generator, device=dev) * 0.02
    return apply_transform(points, colors, uniform(0.0, 2 * math.pi, B), uniform(0.9, 1.1, B),
                           uniform(-0.05, 0.05, B, 3), noise)


class DeviceBatcher:
    """Raw scenes resident on the device -> training batches, the work of `__getitem__` + collate.

    scenes: list of dicts with `points` [n_i,3] fp32, optional `colors` [n_i,3] (zeros if absent,
    :118) and `labels` [n_i] int64 (zeros if absent, :119) -- the datasets of the reference's .h5 files."""

    def __init__(self, scenes, num_points, transform=True, device="cuda", seed=None):
        self.num_points = int(num_points)
        self.transform = transform
        self.device = torch.device(device)
        self.generator = torch.Generator(device=self.device)
        if seed is not None:
            self.generator.manual_seed(int(seed))
        self.scenes = []
        for sc in scenes:
            pts = torch.as_tensor(sc["points"], dtype=torch.float32).to(self.device)
            col = sc.get("colors")
            lab = sc.get("labels")
            self.scenes.append({
                "points": pts,
                "colors": torch.zeros_like(pts) if col is None else torch.as_tensor(col, dtype=torch.float32).to(self.device),
                "labels": (torch.zeros(pts.shape[0], dtype=torch.int64, device=self.device) if lab is None
                           else torch.as_tensor(lab, dtype=torch.int64).to(self.device))})

    def __len__(self):
        return len(self.scenes)

    def batch(self, scene_ids):
        """One batch dict for the given scene numbers (keys as simpdataset.py:181-189)."""
        idx = [subsample_indices(self.scenes[i]["points"].shape[0], self.num_points, self.generator, self.device)
               for i in scene_ids]
        raw = torch.stack([self.scenes[i]["points"][j] for i, j in zip(scene_ids, idx)])
        colors = torch.stack([self.scenes[i]["colors"][j] for i, j in zip(scene_ids, idx)])
        labels = torch.stack([self.scenes[i]["labels"][j] for i, j in zip(scene_ids, idx)])
        points = normalize_points(raw)
        out_colors = colors
        if self.transform:
            points, out_colors = random_transform(points, colors, self.generator)
        points = torch.nan_to_num(points)            # :164-166
        out_colors = torch.nan_to_num(out_colors)    # :168-170
        return {"points": points.contiguous(), "colors": out_colors.contiguous(), "labels": labels,
                "original_points": raw, "original_colors": colors, "indices": torch.stack(idx)}
