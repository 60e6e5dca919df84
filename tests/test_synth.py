import numpy as np


def test_mountain_cloud_is_deterministic_and_shuffled(synth):
    a = synth.mountain_cloud(64, 48, seed=9)
    b = synth.mountain_cloud(64, 48, seed=9)
    assert a.dtype == np.float32 and a.shape == (64 * 48, 3)
    assert np.array_equal(a, b)
    assert not np.all(np.diff(a[:, 0]) >= 0)
    c = synth.mountain_cloud(64, 48, seed=9, origin=(100.0, 0.0))
    assert abs(float(c[:, 0].min()) - 100.0) < 0.05


def test_voxel_centroids(synth):
    pts = np.array([[0.01, 0.01, 0.0], [0.05, 0.07, 0.1], [0.31, 0.02, 0.0], [0.02, 0.33, 0.5]],
                   np.float32)
    v = synth.voxel_centroids(pts, 0.2)
    assert v.shape == (3, 3)
    assert np.allclose(v[0], pts[:2].mean(0))


def test_indoor_cloud(synth):
    pts, boxes = synth.indoor_cloud(seed=1, size=(12.0, 9.0), n_boxes=4)
    assert pts.shape[1] == 3 and len(boxes) == 8 and pts[:, 2].max() > 2.0
