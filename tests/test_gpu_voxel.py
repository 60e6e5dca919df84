"""GPU: the voxel-grid filter of the map-ingest stage against the CPU
restatement of pcl::VoxelGrid in oracle/ -- bit-exact (same fp32 summation order)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(oa):
    import trg_planner
    return trg_planner.Engine(**oa.INDOOR)


@pytest.mark.parametrize("leaf", [0.2, 0.05, 1.0])
def test_voxel_filter_matches_oracle(oa, eng, synth, leaf):
    pts, _ = synth.indoor_cloud(seed=2, size=(16.0, 12.0), n_boxes=5)
    out = eng.voxel_filter(pts, leaf)
    ref, passthrough = oa.voxel_grid(pts, leaf)
    assert not passthrough
    assert out.shape == ref.shape and out.shape[0] < pts.shape[0]
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))


def test_voxel_filter_edge_cases(oa, eng):
    rng = np.random.default_rng(1)
    pts = rng.uniform(-5, 5, (5000, 3)).astype(np.float32)
    pts[::97] = np.nan                       # non-finite points are dropped (cloud not dense)
    out = eng.voxel_filter(pts, 0.5)
    ref, _ = oa.voxel_grid(pts, 0.5)
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))
    assert eng.voxel_filter(np.zeros((0, 3), np.float32), 0.2).shape == (0, 3)
    one = eng.voxel_filter(np.asarray([[1.0, 2.0, 3.0]], np.float32), 0.2)
    assert np.array_equal(one, [[1.0, 2.0, 3.0]])
    # leaf too small for int32 voxel indices: PCL warns and hands the input through
    out = eng.voxel_filter(pts[:200], 1e-4)
    ref, passthrough = oa.voxel_grid(pts[:200], 1e-4)
    assert passthrough and np.array_equal(out.view(np.uint32), ref.view(np.uint32))


def test_voxel_filter_large(oa, eng, synth):
    cloud = synth.mountain_tile(0, 1000, 0, 1000, seed=3)      # 1 M points -> 0.2 m voxels
    out = eng.voxel_filter(cloud, 0.2)
    ref, _ = oa.voxel_grid(cloud, 0.2)
    assert out.shape == ref.shape and np.array_equal(out.view(np.uint32), ref.view(np.uint32))
