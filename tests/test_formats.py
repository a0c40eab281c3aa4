"""CPU tests of the on-disk formats either side of the render path (exavatar_release_amd/formats.py): COLMAP text models,
per-frame JSON cameras / parameters, snapshots -- against files written here in the layouts the reference's dataset classes
read (avatar/data/NeuMan/NeuMan.py:34-104, avatar/data/Custom/Custom.py:38-125, avatar/common/base.py:147-158)."""
import json
import math
import os

import numpy as np
import pytest
import torch

from exavatar_release_amd import formats
from exavatar_release_amd.camera import make_raster_matrices

CAMERAS = """# Camera list with one line of data per camera:
#   CAMERA_ID, MODEL, WIDTH, HEIGHT, PARAMS[]
# Number of cameras: 1
1 PINHOLE 1280 720 1100.5 1101.25 640.0 360.5
"""

# a rotation of 90 degrees about y (w, x, y, z) = (cos 45, 0, sin 45, 0) and the identity
IMAGES = """# Image list with two lines of data per image:
#   IMAGE_ID, QW, QX, QY, QZ, TX, TY, TZ, CAMERA_ID, NAME
#   POINTS2D[] as (X, Y, POINT3D_ID)
1 0.7071067811865476 0.0 0.7071067811865476 0.0 0.5 -0.25 2.0 1 00012.png
10.5 20.25 -1 33.0 44.0 7
2 1.0 0.0 0.0 0.0 0.0 0.0 0.0 1 00013.png

"""


def _points(n):
    lines = ['# 3D point list with one line of data per point:', '#   POINT3D_ID, X, Y, Z, R, G, B, ERROR, TRACK[] as (IMAGE_ID, POINT2D_IDX)']
    for i in range(n):
        lines.append('%d %.3f %.3f %.3f %d %d %d 0.5 1 %d 2 %d' % (i + 1, 0.1 * i, -0.2 * i, 1.0 + i, i % 256, 255, 0, i, i))
    return '\n'.join(lines) + '\n'


def _write_sparse(root):
    sparse = os.path.join(root, 'sparse')
    os.makedirs(sparse)
    for name, text in (('cameras.txt', CAMERAS), ('images.txt', IMAGES), ('points3D.txt', _points(40))):
        with open(os.path.join(sparse, name), 'w') as f:
            f.write(text)
    return sparse


def test_frame_index_of_both_naming_schemes():
    assert formats.frame_index_of('/a/b/00042.png') == 42           # NeuMan
    assert formats.frame_index_of('image0042.jpg') == 42            # Custom ("image" prefix)
    assert formats.frame_index_of('17.json') == 17
    with pytest.raises(ValueError):
        formats.frame_index_of('frame.png')


def test_colmap_text_model(tmp_path):
    sparse = _write_sparse(str(tmp_path))
    cams = formats.read_colmap_cameras(os.path.join(sparse, 'cameras.txt'))
    assert set(cams) == {1} and cams[1]['width'] == 1280 and cams[1]['height'] == 720
    assert np.allclose(cams[1]['focal'], [1100.5, 1101.25]) and np.allclose(cams[1]['princpt'], [640.0, 360.5])
    poses, pts = formats.read_colmap_model(sparse)
    assert set(poses) == {12, 13}                                   # the POINTS2D lines are not poses
    p = poses[12]
    assert p['R'].dtype == np.float32 and p['R'].shape == (3, 3) and p['t'].dtype == np.float32
    # 90 degrees about +y: x -> -z, z -> x
    assert np.allclose(p['R'], [[0, 0, 1], [0, 1, 0], [-1, 0, 0]], atol=1e-6)
    assert np.allclose(p['t'], [0.5, -0.25, 2.0])
    assert np.allclose(poses[13]['R'], np.eye(3)) and np.allclose(poses[13]['focal'], [1100.5, 1101.25])
    # 40 points with z = 1..40: the 95 % quantile (interpolated, 38.05) keeps z < 38.05 -> 38 points; colours / 255
    assert pts.shape == (38, 6) and pts.dtype == torch.float32
    assert float(pts[:, 2].max()) == 38.0
    assert torch.allclose(pts[5], torch.tensor([0.5, -1.0, 6.0, 5 / 255, 1.0, 0.0]))
    assert formats.read_colmap_points3d(os.path.join(sparse, 'points3D.txt'), z_quantile=None).shape == (40, 6)


def test_colmap_cam_params_drive_the_raster_matrices(tmp_path):
    """A pose read from disk goes through the same camera maths as a synthetic one (reference module.py:598-607)."""
    poses, _ = formats.read_colmap_model(_write_sparse(str(tmp_path)))
    cam = formats.cam_param_to(poses[12], 'cpu')
    assert all(v.dtype == torch.float32 for v in cam.values())
    tanx, tany, view, proj, cpos = make_raster_matrices(cam, (720, 1280))
    assert math.isclose(tanx, 1280 / (2 * 1100.5), rel_tol=1e-6) and math.isclose(tany, 720 / (2 * 1101.25), rel_tol=1e-6)
    # camera centre = -R^T t
    c = -(torch.tensor(poses[12]['R']).T @ torch.tensor(poses[12]['t']))
    assert torch.allclose(cpos.view(-1)[:3], c, atol=1e-5)
    assert view.shape == (4, 4) and proj.shape == (4, 4)


def test_unsupported_camera_model_is_refused(tmp_path):
    path = tmp_path / 'cameras.txt'
    path.write_text('1 OPENCV 640 480 500 500 320 240 0.1 0.0 0.0 0.0\n')
    with pytest.raises(ValueError):
        formats.read_colmap_cameras(str(path))
    path.write_text('3 SIMPLE_PINHOLE 640 480 500 320 240\n')
    cams = formats.read_colmap_cameras(str(path))
    assert np.allclose(cams[3]['focal'], [500, 500]) and np.allclose(cams[3]['princpt'], [320, 240])


def test_virtual_cam_params_round_trip(tmp_path):
    d = str(tmp_path / 'cam_params')
    formats.write_virtual_cam_params(d, [3, 11, 7], (480, 640))
    cams = formats.read_cam_params_json(d)
    assert set(cams) == {3, 7, 11}
    for c in cams.values():
        assert np.array_equal(c['R'], np.eye(3, dtype=np.float32)) and np.array_equal(c['t'], np.zeros(3, dtype=np.float32))
        assert np.array_equal(c['focal'], np.array([2000, 2000], dtype=np.float32))
        assert np.array_equal(c['princpt'], np.array([320, 240], dtype=np.float32))
    with open(os.path.join(d, '11.json')) as f:                     # the file itself has the reference's four keys
        assert set(json.load(f)) == {'R', 't', 'focal', 'princpt'}


def test_float_params_json(tmp_path):
    d = tmp_path / 'smplx_params'
    d.mkdir()
    (d / '5.json').write_text(json.dumps({'root_pose': [0.1, 0.2, 0.3], 'body_pose': [[0.0] * 3] * 21, 'trans': [0, 0, 2.5]}))
    out = formats.read_float_params_json(str(d))
    assert set(out) == {5} and out[5]['body_pose'].shape == (21, 3) and out[5]['trans'].dtype == torch.float32
    assert torch.allclose(out[5]['root_pose'], torch.tensor([0.1, 0.2, 0.3]))


def test_snapshots(tmp_path):
    d = str(tmp_path / 'model_dump')
    assert formats.latest_snapshot_epoch(d) is None
    with pytest.raises(FileNotFoundError):
        formats.load_snapshot(d)
    for epoch in (0, 3, 12):
        path = formats.save_snapshot({'epoch': epoch, 'network': {'w': torch.full((2,), float(epoch))}}, d, epoch)
        assert os.path.basename(path) == 'snapshot_%d.pth' % epoch
    open(os.path.join(d, 'notes.pth.txt'), 'w').close()             # not a snapshot
    assert formats.latest_snapshot_epoch(d) == 12                   # numeric, not lexicographic, order
    ck = formats.load_snapshot(d)
    assert ck['epoch'] == 12 and torch.equal(ck['network']['w'], torch.full((2,), 12.0))
    assert formats.load_snapshot(d, 3)['epoch'] == 3
