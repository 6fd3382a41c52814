"""CPU: device-side batch preparation (row f3) against a numpy restatement of the reference's
per-sample pipeline, utils/simpdataset.py:47-100, :136-142.  The reference module imports h5py at load
time (absent here), so these checks rest on the restatement below: "parity unpinned" by a reference run;
the arithmetic is three lines per function."""
import numpy as np
import pytest
import torch

from pointcloud_bridge_amd import data


def np_normalize(points):
    """simpdataset.py:47-63."""
    centred = points - np.mean(points, axis=0)
    radius = np.max(np.sqrt(np.sum(centred ** 2, axis=1)))
    return centred / radius if radius > 1e-6 else centred


def np_transform(points, colors, theta, scale, translation, noise):
    """simpdataset.py:73-95 with the draws passed in."""
    rot = np.array([[np.cos(theta), -np.sin(theta), 0], [np.sin(theta), np.cos(theta), 0], [0, 0, 1]], dtype=points.dtype)
    out = np.dot(points.copy(), rot)
    out *= scale
    out += translation.reshape(1, 3).astype(points.dtype)
    return out, np.clip(colors + noise.astype(colors.dtype), 0, 1)


def test_normalize_matches_numpy_pipeline():
    rng = np.random.default_rng(0)
    pts = (rng.normal(size=(3, 500, 3)) * [30, 5, 2] + [100, -40, 7]).astype(np.float32)
    pts[2] = np.array([2.0, -4.0, 8.0], np.float32)  # degenerate scene: every point the same (mean exact in fp32)
    got = data.normalize_points(torch.from_numpy(pts)).numpy()
    for b in range(3):
        np.testing.assert_allclose(got[b], np_normalize(pts[b]), rtol=1e-5, atol=1e-6)
    assert np.abs(np.linalg.norm(got[0], axis=1).max() - 1) < 1e-6
    assert np.all(got[2] == 0)


def test_transform_matches_numpy_pipeline_for_given_draws():
    rng = np.random.default_rng(1)
    pts = rng.uniform(-1, 1, size=(4, 300, 3)).astype(np.float32)
    col = rng.uniform(0, 1, size=(4, 300, 3)).astype(np.float32)
    theta = rng.uniform(0, 2 * np.pi, 4)
    scale = rng.uniform(0.9, 1.1, 4)
    trans = rng.uniform(-0.05, 0.05, (4, 3))
    noise = rng.normal(0, 0.02, col.shape)
    gp, gc = data.apply_transform(torch.from_numpy(pts), torch.from_numpy(col), torch.from_numpy(theta),
                                  torch.from_numpy(scale), torch.from_numpy(trans), torch.from_numpy(noise))
    for b in range(4):
        wp, wc = np_transform(pts[b], col[b], theta[b], scale[b], trans[b], noise[b])
        np.testing.assert_allclose(gp[b].numpy(), wp, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(gc[b].numpy(), wc, rtol=1e-6, atol=1e-7)
    assert gp.dtype == torch.float32 and gc.dtype == torch.float32


def test_random_transform_draws_stay_in_the_reference_ranges():
    g = torch.Generator().manual_seed(5)
    pts = torch.zeros(2000, 4, 3)
    pts[:, 0, 0] = 1.0   # unit x vector: its image gives scale and angle back
    col = torch.full((2000, 4, 3), 0.5)
    p, c = data.random_transform(pts, col, g)
    t = p[:, 1]                                  # image of the origin = translation
    assert float(t.abs().max()) <= 0.05 and float(t.abs().max()) > 0.045
    v = p[:, 0] - t
    s = v.norm(dim=-1)
    assert float(s.min()) >= 0.9 - 1e-6 and float(s.max()) <= 1.1 + 1e-6 and float(s.std()) > 0.04
    assert float(v[:, 2].abs().max()) < 1e-6     # rotation about z only
    ang = torch.atan2(-v[:, 1], v[:, 0])         # row vector times R: (1,0,0) -> (cos, -sin, 0)
    assert float(ang.min()) < -3.0 and float(ang.max()) > 3.0
    assert 0.015 < float((c - 0.5).std()) < 0.025 and float(c.min()) >= 0 and float(c.max()) <= 1


@pytest.mark.parametrize("n,k", [(1000, 256), (256, 256), (100, 256), (1, 8)])
def test_subsample_index_sets(n, k):
    g = torch.Generator().manual_seed(n + k)
    idx = data.subsample_indices(n, k, g)
    assert idx.shape == (k,) and int(idx.min()) >= 0 and int(idx.max()) < n
    if n >= k:
        assert idx.unique().numel() == k                       # without replacement
    else:
        assert idx.unique().numel() == n                       # every point at least once
    with pytest.raises(ValueError):
        data.subsample_indices(0, k, g)


def test_device_batcher_builds_the_trainers_batch_dict():
    rng = np.random.default_rng(3)
    scenes = [{"points": rng.normal(size=(n, 3)).astype(np.float32) * 10 + 50,
               "colors": rng.uniform(size=(n, 3)).astype(np.float32),
               "labels": rng.integers(0, 5, n)} for n in (900, 300)]
    scenes.append({"points": rng.normal(size=(700, 3)).astype(np.float32)})   # no colours / labels in the file
    b = data.DeviceBatcher(scenes, 512, transform=False, device="cpu", seed=0).batch([0, 1, 2])
    assert b["points"].shape == (3, 512, 3) and b["colors"].shape == (3, 512, 3) and b["labels"].shape == (3, 512)
    assert b["labels"].dtype == torch.int64 and b["points"].dtype == torch.float32
    for i, sc in enumerate(scenes):
        sel = b["indices"][i].numpy()
        np.testing.assert_array_equal(b["original_points"][i].numpy(), sc["points"][sel])
        np.testing.assert_allclose(b["points"][i].numpy(), np_normalize(sc["points"][sel]), rtol=1e-5, atol=1e-6)
        if "labels" in sc:
            np.testing.assert_array_equal(b["labels"][i].numpy(), sc["labels"][sel])
    assert float(b["colors"][2].abs().max()) == 0 and int(b["labels"][2].abs().max()) == 0
    aug = data.DeviceBatcher(scenes, 512, transform=True, device="cpu", seed=0).batch([0, 1])
    assert float(aug["points"].norm(dim=-1).max()) < 1.1 * 1.0 + 0.05 * 3 ** 0.5 + 1e-5
