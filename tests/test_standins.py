"""CPU tests of the torch-only pytorch3d stand-ins: brute-force and analytic known answers.
pytorch3d itself is not installable here, so these pin the published semantics the reference relies on
(avatar/common/nets/module.py:86,502,543; avatar/common/utils/smpl_x.py:73-100)."""
import math

import torch

from exavatar_release_amd import p3d_standins as p3d


def test_knn_points_matches_brute_force_and_reference_usage():
    g = torch.Generator().manual_seed(0)
    a = torch.randn(2, 300, 3, generator=g)
    b = torch.randn(2, 500, 3, generator=g)
    out = p3d.knn_points(a, b, K=4, return_nn=True)
    d = ((a[:, :, None, :] - b[:, None, :, :]) ** 2).sum(3)
    dd, ii = torch.topk(d, 4, dim=2, largest=False, sorted=True)
    assert torch.equal(out.idx, ii)
    assert torch.allclose(out.dists, dd, atol=1e-6)
    assert torch.allclose(out.knn, torch.gather(b[:, None].expand(2, 300, 500, 3), 2, ii[..., None].expand(2, 300, 4, 3)))
    # module.py:86-87: self-query with K = 4, neighbour 0 is the point itself (distance 0), then the 3 nearest others
    xyz = torch.randn(1, 200, 3, generator=g)
    pts = p3d.knn_points(xyz, xyz, K=4, return_nn=True)
    assert torch.equal(pts.idx[0, :, 0], torch.arange(200)) and float(pts.dists[0, :, 0].abs().max()) == 0.0
    dist = torch.sum((xyz[0, :, None, :] - pts.knn[0, :, 1:, :]) ** 2, 2).mean(1)
    full = ((xyz[0, :, None] - xyz[0, None]) ** 2).sum(2)
    full.fill_diagonal_(float('inf'))
    assert torch.allclose(dist, torch.topk(full, 3, dim=1, largest=False).values.mean(1), atol=1e-6)
    # module.py:543: K = 1 -> .idx[0, :, 0]
    nn1 = p3d.knn_points(a[:1], b[:1], K=1, return_nn=True).idx[0, :, 0]
    assert torch.equal(nn1, d[0].argmin(1))
    # gradients reach both point sets
    a1, b1 = a[:1].clone().requires_grad_(True), b[:1].clone().requires_grad_(True)
    p3d.knn_points(a1, b1, K=2).dists.sum().backward()
    assert float(a1.grad.abs().sum()) > 0 and float(b1.grad.abs().sum()) > 0


def _tetra():
    v = torch.tensor([[0.0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]])
    f = torch.tensor([[0, 2, 1], [0, 1, 3], [0, 3, 2], [1, 2, 3]])          # outward orientation
    return v, f


def test_meshes_edges_and_vertex_normals():
    v, f = _tetra()
    m = p3d.Meshes(v[None], f[None])
    e = m.edges_packed()
    assert e.tolist() == [[0, 1], [0, 2], [0, 3], [1, 2], [1, 3], [2, 3]]     # unique, sorted by (low, high)
    f2e = m.faces_packed_to_edges_packed()
    for fi in range(4):
        for c in range(3):                       # edge c of a face is the one opposite corner c
            a_, b_ = f[fi, (c + 1) % 3].item(), f[fi, (c + 2) % 3].item()
            assert e[f2e[fi, c]].tolist() == sorted((a_, b_))
    n = m.verts_normals_packed()
    assert torch.allclose(n.norm(dim=1), torch.ones(4), atol=1e-6)
    assert torch.allclose(n[0], -torch.ones(3) / math.sqrt(3), atol=1e-6)    # the corner at the origin points to (-1,-1,-1)
    # area weighting: a unit sphere's vertex normals are the positions themselves
    ico_v, ico_f = _icosphere()
    nn_ = p3d.Meshes(ico_v[None], ico_f[None]).verts_normals_packed()
    assert float((nn_ - ico_v).abs().max()) < 0.05


def _icosphere():
    t = (1.0 + math.sqrt(5.0)) / 2.0
    v = torch.tensor([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                      [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=torch.float32)
    f = torch.tensor([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2],
                      [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11],
                      [6, 2, 10], [8, 6, 7], [9, 8, 1]])
    v = v / v.norm(dim=1, keepdim=True)
    m = p3d.Meshes(v[None], f[None])
    for _ in range(2):
        m = p3d.SubdivideMeshes(m)(m)
        m = p3d.Meshes(torch.nn.functional.normalize(m.verts_padded(), dim=2), m.faces_padded())
    return m.verts_list()[0], m.faces_list()[0]


def test_subdivide_meshes_counts_midpoints_features_and_orientation():
    v, f = _tetra()
    mesh = p3d.Meshes(v[None], f[None])
    sub = p3d.SubdivideMeshes(mesh)
    feats = torch.arange(4, dtype=torch.float32)[:, None] * torch.tensor([[1.0, 10.0]])
    new, nf = sub(mesh, feats)
    V, E, F_ = 4, 6, 4
    nv, nfaces = new.verts_list()[0], new.faces_list()[0]
    assert nv.shape == (V + E, 3) and nfaces.shape == (4 * F_, 3) and nf.shape == (1, V + E, 2)
    e = mesh.edges_packed()
    assert torch.equal(nv[:V], v) and torch.allclose(nv[V:], v[e].mean(1))           # midpoints in edge order
    assert torch.allclose(nf[0, V:], feats[e].mean(1))
    # faces: (v0, m2, m1), (v1, m0, m2), (v2, m1, m0), (m0, m1, m2), group by group
    m_ = mesh.faces_packed_to_edges_packed() + V
    assert torch.equal(nfaces[:F_], torch.stack((f[:, 0], m_[:, 2], m_[:, 1]), 1))
    assert torch.equal(nfaces[3 * F_:], m_)
    # same surface, same orientation: total signed volume and area are preserved
    def vol_area(vv, ff):
        a, b, c = vv[ff[:, 0]], vv[ff[:, 1]], vv[ff[:, 2]]
        return float((a * torch.cross(b, c, dim=1)).sum() / 6), float(torch.cross(b - a, c - a, dim=1).norm(dim=1).sum() / 2)
    assert abs(vol_area(v, f)[0] - vol_area(nv, nfaces)[0]) < 1e-6 and abs(vol_area(v, f)[1] - vol_area(nv, nfaces)[1]) < 1e-6
    # two levels, as smpl_x.get_subdivider(2): V2 = V + E0 + E1 (the count the survey derives for the 167 k avatar)
    sub2 = p3d.SubdivideMeshes(new)
    new2 = sub2(new)
    assert new2.verts_list()[0].shape[0] == V + E + new.edges_packed().shape[0]
    # a batch of vertex sets through the precomputed topology (smpl_x.upsample_mesh re-uses the subdividers)
    batch = torch.stack((v, 2 * v))
    nb = sub(p3d.Meshes(batch, f[None].expand(2, -1, -1)))
    assert torch.allclose(nb.verts_padded()[1], 2 * nv)


def test_rotation_conversions_round_trip():
    g = torch.Generator().manual_seed(3)
    aa = torch.randn(50, 3, generator=g)
    R = p3d.axis_angle_to_matrix(aa)
    assert torch.allclose(R @ R.transpose(1, 2), torch.eye(3).expand(50, 3, 3), atol=1e-5)
    assert torch.allclose(torch.linalg.det(R), torch.ones(50), atol=1e-5)
    # rotation by angle |aa| about aa: trace = 1 + 2 cos(angle)
    assert torch.allclose(R.diagonal(dim1=1, dim2=2).sum(1), 1 + 2 * torch.cos(aa.norm(dim=1)), atol=1e-5)
    q = p3d.matrix_to_quaternion(R)
    assert torch.all(q[:, 0] >= 0) and torch.allclose(p3d.quaternion_to_matrix(q), R, atol=1e-5)
    d6 = p3d.matrix_to_rotation_6d(R)
    assert d6.shape == (50, 6) and torch.allclose(p3d.rotation_6d_to_matrix(d6), R, atol=1e-5)
    # module.py:90: identity -> 6D (1, 0, 0, 0, 1, 0)
    assert p3d.matrix_to_rotation_6d(torch.eye(3)[None]).tolist() == [[1.0, 0.0, 0.0, 0.0, 1.0, 0.0]]


def test_matrix_to_axis_angle_round_trip_and_reference_usage():
    g = torch.Generator().manual_seed(4)
    aa = torch.randn(64, 3, generator=g)
    aa = aa / aa.norm(dim=1, keepdim=True) * (torch.rand(64, 1, generator=g) * 3.0 + 0.01)      # angles in (0, pi)
    back = p3d.matrix_to_axis_angle(p3d.axis_angle_to_matrix(aa))
    assert torch.allclose(back, aa, atol=1e-5)
    # module.py:363-364: the inverse pose = the negated axis-angle
    inv = p3d.matrix_to_axis_angle(torch.inverse(p3d.axis_angle_to_matrix(aa)))
    assert torch.allclose(inv, -aa, atol=1e-5)
    # tiny rotations go through the series branch
    tiny = torch.tensor([[1e-8, 0.0, 0.0], [0.0, 0.0, 0.0]])
    assert torch.allclose(p3d.matrix_to_axis_angle(p3d.axis_angle_to_matrix(tiny)), tiny, atol=1e-7)
    # module.py:680: 6D -> matrix -> axis-angle
    d6 = p3d.matrix_to_rotation_6d(p3d.axis_angle_to_matrix(aa))
    assert torch.allclose(p3d.matrix_to_axis_angle(p3d.rotation_6d_to_matrix(d6)), aa, atol=1e-4)


def test_look_at_view_transform_turntable_like_the_reference_drivers():
    """get_neutral_pose.py:76-82 / animate_view_rot.py:93-95: R, t = look_at_view_transform(dist, elev, azim,
    degrees=False, at=at_point[None], up=((0, 1, 0),)); R = torch.inverse(R) -> cam_param for the renderer."""
    import math
    at = torch.tensor([0.1, -0.2, 3.0])
    dist, elev = 2.5, -math.pi / 6
    for i in range(8):
        azim = math.pi + 2 * math.pi * i / 8
        R, T = p3d.look_at_view_transform(dist=dist, elev=elev, azim=azim, degrees=False, at=at[None, :], up=((0, 1, 0),))
        assert R.shape == (1, 3, 3) and T.shape == (1, 3)
        assert torch.allclose(R[0] @ R[0].t(), torch.eye(3), atol=1e-5) and abs(float(torch.linalg.det(R[0])) - 1) < 1e-5
        # camera centre: the spherical formula around `at`; in pytorch3d's row convention X_cam = X_world R + T
        C = at + dist * torch.tensor([math.cos(elev) * math.sin(azim), math.sin(elev), math.cos(elev) * math.cos(azim)])
        assert torch.allclose(C @ R[0] + T[0], torch.zeros(3), atol=1e-5)              # the centre maps to the origin
        at_cam = at @ R[0] + T[0]
        assert torch.allclose(at_cam, torch.tensor([0.0, 0.0, dist]), atol=1e-5)        # `at` sits on the optical axis
        # the reference's column-vector cam_param: x_cam = R_ref x + t with R_ref = inverse(R)
        cp = p3d.turntable_cam_param(dist, elev, azim, at, (1500.0, 1500.0), (512.0, 512.0))
        assert torch.allclose(cp['R'], torch.inverse(R[0]), atol=1e-6) and torch.allclose(cp['t'], T[0])
        assert torch.allclose(cp['R'] @ at + cp['t'], torch.tensor([0.0, 0.0, dist]), atol=1e-5)
        # "up" stays up: the world's +Y has a positive component along the camera's +Y axis
        assert float((torch.tensor([0.0, 1.0, 0.0]) @ R[0])[1]) > 0
    # degrees, eye= and the degenerate up-parallel case
    R, T = p3d.look_at_view_transform(dist=2.0, elev=0.0, azim=90.0)
    assert torch.allclose(torch.tensor([2.0, 0.0, 0.0]) @ R[0] + T[0], torch.zeros(3), atol=1e-5)
    R2, T2 = p3d.look_at_view_transform(eye=((0.0, 0.0, -4.0),), at=((0.0, 0.0, 0.0),))
    assert torch.allclose(R2[0], torch.eye(3), atol=1e-6) and torch.allclose(T2[0], torch.tensor([0.0, 0.0, 4.0]))
    R3, _ = p3d.look_at_view_transform(eye=((0.01, 5.0, 0.0),), at=((0.0, 0.0, 0.0),))    # looking (almost) straight down
    assert torch.isfinite(R3).all() and abs(float(torch.linalg.det(R3[0])) - 1) < 1e-4
    # up exactly parallel to the viewing direction is degenerate in pytorch3d as well (x = up x z = 0, the fallback
    # y x z is 0 too): finite, but not a rotation -- the reference never gets there (elev = -pi/6 or |elev| < pi/2)
    R4, _ = p3d.look_at_view_transform(eye=((0.0, 5.0, 0.0),), at=((0.0, 0.0, 0.0),))
    assert torch.isfinite(R4).all()


def test_turntable_camera_through_the_raster_matrices():
    """A turntable cam_param built the reference's way goes through make_raster_matrices like any other camera: the
    look-at point projects to the image centre."""
    import math
    from exavatar_release_amd.camera import make_raster_matrices
    at = torch.tensor([0.0, 0.3, 3.0])
    cp = p3d.turntable_cam_param(2.0, -math.pi / 6, math.pi + 0.7, at, (1500.0, 1500.0), (512.0, 512.0))
    tanx, tany, view, proj, campos = make_raster_matrices(cp, (1024, 1024))
    h = torch.cat((at, torch.ones(1))) @ proj
    ndc = h[:2] / h[3]
    assert torch.allclose(ndc, torch.zeros(2), atol=1e-5)
    assert torch.allclose(campos, -cp['R'].t() @ cp['t'], atol=1e-5)


def test_save_obj(tmp_path):
    v = torch.tensor([[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.0, 1.5, -2.0]])
    f = torch.tensor([[0, 1, 2]])
    path = tmp_path / 'm.obj'
    p3d.save_obj(str(path), v, f)
    lines = path.read_text().strip().splitlines()
    assert lines[0].startswith('v 0.0') and lines[2].split() == ['v', '0.000000', '1.500000', '-2.000000']
    assert lines[3] == 'f 1 2 3'
