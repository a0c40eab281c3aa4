"""On-disk formats either side of the render path ("data formats"): what the reference's
dataset classes read before they hand cameras and scene points to ``GaussianRenderer`` and what its trainer writes.

* COLMAP text models -- ``sparse/cameras.txt`` (one shared PINHOLE camera), ``sparse/images.txt`` (world-to-camera
  quaternion (w, x, y, z) + translation per registered image), ``sparse/points3D.txt`` (xyz + 8-bit rgb):
  reference ``avatar/data/NeuMan/NeuMan.py:34-104`` and ``avatar/data/Custom/Custom.py:38-125``.
* per-frame JSON camera files ``cam_params/<frame>.json`` with keys R, t, focal, princpt
  (``Custom.py:67-74``; written by ``fitting/tools/make_virtual_cam_params.py:27``) and per-frame JSON SMPL-X parameter
  files (``NeuMan.py:82-88``).
* ``snapshot_<epoch>.pth`` checkpoints (``avatar/common/base.py:147-158``).

Plain host code (numpy / torch CPU); the results are the ``cam_param`` dicts ``GaussianRenderer.forward`` takes
(R [3,3], t [3], focal [2], princpt [2], float32) and [N, 6] xyz-rgb point tensors for the scene Gaussians' initialisation.
"""
import glob
import json
import os
import re

import numpy as np
import torch

from .p3d_standins import quaternion_to_matrix

_FRAME_RE = re.compile(r'(\d+)(?=\.[A-Za-z0-9]+$)')


def frame_index_of(name):
    """Frame number encoded in an image file name: the digits in front of the extension (``00042.png`` -> 42 as in
    NeuMan.py:53, ``image0042.jpg`` -> 42 as in Custom.py:61)."""
    m = _FRAME_RE.search(os.path.basename(name))
    if m is None:
        raise ValueError('no frame number in %r' % (name,))
    return int(m.group(1))


def _data_lines(path):
    with open(path) as f:
        for line in f:
            line = line.strip()
            if line and not line.startswith('#'):
                yield line


def read_colmap_cameras(path):
    """``cameras.txt``: ``CAMERA_ID MODEL WIDTH HEIGHT PARAMS...``.  Returns {camera_id: dict(model, width, height,
    focal[2], princpt[2])}; PINHOLE (fx fy cx cy) is what the reference expects (NeuMan.py:41), SIMPLE_PINHOLE
    (f cx cy) is widened to two equal focal lengths."""
    cams = {}
    for line in _data_lines(path):
        tok = line.split()
        cam_id, model, width, height = int(tok[0]), tok[1], int(tok[2]), int(tok[3])
        p = [float(x) for x in tok[4:]]
        if model == 'PINHOLE' and len(p) == 4:
            focal, princpt = (p[0], p[1]), (p[2], p[3])
        elif model == 'SIMPLE_PINHOLE' and len(p) == 3:
            focal, princpt = (p[0], p[0]), (p[1], p[2])
        else:
            raise ValueError('%s: unsupported COLMAP camera model %s with %d parameters' % (path, model, len(p)))
        cams[cam_id] = {'model': model, 'width': width, 'height': height,
                        'focal': np.asarray(focal, dtype=np.float32), 'princpt': np.asarray(princpt, dtype=np.float32)}
    if not cams:
        raise ValueError('%s: no camera' % path)
    return cams


def read_colmap_images(path, cameras, frame_of=frame_index_of):
    """``images.txt``: per image one pose line ``IMAGE_ID QW QX QY QZ TX TY TZ CAMERA_ID NAME`` followed by one line of
    2-D points (skipped: only lines whose last token is an image file name are poses, NeuMan.py:49-50).  Returns
    {frame index: cam_param} with the world-to-camera rotation R = matrix of the (w, x, y, z) quaternion and t."""
    out = {}
    for line in _data_lines(path):
        tok = line.split()
        if len(tok) != 10 or not re.search(r'\.(png|jpg|jpeg)$', tok[-1], re.IGNORECASE):
            continue
        q = torch.tensor([float(x) for x in tok[1:5]], dtype=torch.float32)
        cam = cameras[int(tok[8])] if int(tok[8]) in cameras else next(iter(cameras.values()))
        out[frame_of(tok[-1])] = {'R': quaternion_to_matrix(q).numpy().astype(np.float32),
                                  't': np.asarray([float(x) for x in tok[5:8]], dtype=np.float32),
                                  'focal': cam['focal'].copy(), 'princpt': cam['princpt'].copy()}
    return out


def read_colmap_points3d(path, z_quantile=0.95):
    """``points3D.txt``: ``POINT3D_ID X Y Z R G B ERROR TRACK...`` -> float32 tensor [N, 6] (xyz, rgb in [0, 1]).
    Points at or beyond the ``z_quantile`` quantile of z are dropped as outliers, as NeuMan.py:100-101 does
    (``None`` keeps everything)."""
    rows = [[float(x) for x in line.split()[1:7]] for line in _data_lines(path)]
    pts = torch.tensor(rows, dtype=torch.float32).reshape(-1, 6)
    pts[:, 3:] /= 255.0
    if z_quantile is not None and pts.shape[0] > 0:
        pts = pts[pts[:, 2] < torch.quantile(pts[:, 2], z_quantile)]
    return pts


def read_colmap_model(sparse_dir, frame_of=frame_index_of, z_quantile=0.95):
    """cameras + images + points of one ``sparse/`` directory: ({frame: cam_param}, points [N, 6])."""
    cams = read_colmap_cameras(os.path.join(sparse_dir, 'cameras.txt'))
    poses = read_colmap_images(os.path.join(sparse_dir, 'images.txt'), cams, frame_of)
    pts_path = os.path.join(sparse_dir, 'points3D.txt')
    pts = read_colmap_points3d(pts_path, z_quantile) if os.path.isfile(pts_path) else torch.zeros(0, 6)
    return poses, pts


def read_cam_params_json(directory):
    """``<directory>/<frame>.json`` files with R, t, focal, princpt (Custom.py:67-74) -> {frame: cam_param}."""
    out = {}
    for path in sorted(glob.glob(os.path.join(directory, '*.json'))):
        with open(path) as f:
            out[frame_index_of(path)] = {k: np.asarray(v, dtype=np.float32) for k, v in json.load(f).items()}
    return out


def write_virtual_cam_params(directory, frame_indices, img_shape, focal=2000.0):
    """The fixed virtual camera of fitting/tools/make_virtual_cam_params.py:27 for footage without COLMAP poses: identity
    rotation, zero translation, focal 2000 px, principal point at the image centre; one JSON per frame."""
    H, W = img_shape
    os.makedirs(directory, exist_ok=True)
    cam = {'R': np.eye(3, dtype=np.float32).tolist(), 't': [0.0, 0.0, 0.0], 'focal': [float(focal), float(focal)],
           'princpt': [W / 2, H / 2]}
    for fi in frame_indices:
        with open(os.path.join(directory, '%d.json' % int(fi)), 'w') as f:
            json.dump(cam, f)


def read_float_params_json(directory):
    """Per-frame JSON dictionaries of number arrays (the SMPL-X parameter files of NeuMan.py:82-88) ->
    {frame: {key: float32 tensor}}."""
    out = {}
    for path in sorted(glob.glob(os.path.join(directory, '*.json'))):
        with open(path) as f:
            out[frame_index_of(path)] = {k: torch.tensor(v, dtype=torch.float32) for k, v in json.load(f).items()}
    return out


def cam_param_to(cam_param, device):
    """numpy / tensor cam_param -> float32 tensors on ``device`` (what ``GaussianRenderer.forward`` consumes)."""
    return {k: torch.as_tensor(np.asarray(v) if not torch.is_tensor(v) else v, dtype=torch.float32).to(device)
            for k, v in cam_param.items()}


_SNAPSHOT_RE = re.compile(r'snapshot_(\d+)\.pth$')


def snapshot_path(model_dir, epoch):
    return os.path.join(model_dir, 'snapshot_%d.pth' % int(epoch))


def latest_snapshot_epoch(model_dir):
    """Largest N with ``snapshot_N.pth`` in ``model_dir`` (base.py:153-154); None when there is none."""
    epochs = [int(m.group(1)) for m in (_SNAPSHOT_RE.search(p) for p in glob.glob(os.path.join(model_dir, '*.pth'))) if m]
    return max(epochs) if epochs else None


def save_snapshot(state, model_dir, epoch):
    """torch.save of the trainer's state dict under the reference's file name (base.py:147-150)."""
    os.makedirs(model_dir, exist_ok=True)
    path = snapshot_path(model_dir, epoch)
    torch.save(state, path)
    return path


def load_snapshot(model_dir, epoch=None):
    """The newest (or the given) snapshot, tensors mapped to the CPU as base.py:157 does."""
    if epoch is None:
        epoch = latest_snapshot_epoch(model_dir)
        if epoch is None:
            raise FileNotFoundError('no snapshot_*.pth in %s' % model_dir)
    return torch.load(snapshot_path(model_dir, epoch), map_location='cpu')
