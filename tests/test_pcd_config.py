"""CPU: PCD reader/writer (ascii, binary, binary_compressed) and YAML parameter defaults."""
import numpy as np
import pytest


@pytest.mark.parametrize("mode", ["ascii", "binary", "binary_compressed"])
def test_pcd_roundtrip(tmp_path, mode):
    from trg_planner import pcd
    rng = np.random.default_rng(0)
    xyz = rng.normal(size=(257, 3)).astype(np.float32)
    f = tmp_path / f"c_{mode}.pcd"
    pcd.write_pcd(f, xyz, mode)
    back = pcd.read_pcd(f)
    assert back.shape == xyz.shape and np.array_equal(back, xyz)


def test_lzf_back_references():
    from trg_planner import pcd
    # "abcabcabcabc": literal "abc" then a back reference of length 9 at distance 3
    stream = bytes([2]) + b"abc" + bytes([(7 << 5) | 0, 9 - 2 - 7, 2])
    assert pcd.lzf_decompress(stream, 12) == b"abcabcabcabc"


def test_yaml_defaults_and_reference_configs(tmp_path):
    from trg_planner import config
    f = tmp_path / "c.yaml"
    f.write_text("trg:\n  expandDist: 0.4\n")
    p = config.load_params(f)
    assert p.expandDist == 0.4 and p.robotSize == 0.3 and p.sampleNum == 20  # PL.cpp:108-128
    assert p.isVerbose is True and p.graph_rate == 1.0 and p.collisionThreshold == 0.2
    f.write_text("isVerbose: false\ntimer:\n  graphRate: 5.0\ntrg:\n  sampleNum: 7\n"
                 "  heightThreshold: 0.16\n  collisionThreshold: 0.1\n")
    p = config.load_params(f)
    assert p.sampleNum == 7 and p.graph_rate == 5.0 and not p.isVerbose


def test_voxel_grid_oracle_semantics(oa):
    """The CPU restatement of pcl::VoxelGrid: centroid per voxel, ascending voxel index, fp32 sums
    in point order; agrees with the numpy reference within fp32 rounding."""
    from trg_planner import synth
    rng = np.random.default_rng(0)
    pts = rng.uniform([-3, -2, 0], [5, 4, 2.5], (20000, 3)).astype(np.float32)
    out, passthrough = oa.voxel_grid(pts, 0.2)
    ref = synth.voxel_centroids(pts, 0.2)
    assert not passthrough and out.shape == ref.shape
    assert np.abs(out - ref).max() < 2e-6
    # a lattice with one point per voxel is returned re-ordered by voxel index, values untouched
    ii, jj = np.meshgrid(np.arange(10), np.arange(12), indexing="ij")
    lat = np.stack([ii.ravel() * 0.5 + 0.25, jj.ravel() * 0.5 + 0.25, np.zeros(120)], 1).astype(np.float32)
    out, _ = oa.voxel_grid(lat[rng.permutation(120)], 0.5)
    key = np.lexsort((out[:, 0], out[:, 1]))
    assert out.shape == (120, 3) and np.array_equal(key, np.arange(120))
    # non-finite points are skipped; an absurdly small leaf hands the input through (PCL's warning path)
    bad = np.concatenate([pts[:100], [[np.nan, 0, 0], [np.inf, 1, 1]]]).astype(np.float32)
    out, _ = oa.voxel_grid(bad, 0.2)
    assert np.isfinite(out).all()
    out, passthrough = oa.voxel_grid(pts, 1e-4)
    assert passthrough and out.shape == pts.shape
