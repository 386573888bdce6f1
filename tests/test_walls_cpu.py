"""Host logic of the wall certificates (csrc/pt_host_scene.h: choose_walls), on the CPU: the planes pt_init keeps for ROTATED walls
(ptd::wallPlanesOriented, round 5) are what the certificate's argument needs -- checked in double precision against the cubes' corners.
The device side is swept in tests/test_gpu_parity.py::test_wall_planes_never_reject_a_hit."""
import numpy as np
import pytest


def _corners(oracle, g):
    T = np.asarray(g["transform"], np.float64).reshape(4, 4).T          # column-major 4x4 -> rows
    pts = np.array([[(q & 1) - 0.5, ((q >> 1) & 1) - 0.5, ((q >> 2) & 1) - 0.5, 1.0] for q in range(8)])
    return (pts @ T.T)[:, :3]


def _rooms(oracle):
    S = oracle.make_geom
    return {
        "cornell": ([S(1, 0, (0, 0, 0), (0, 0, 0), (10, .01, 10)), S(1, 0, (0, 10, 0), (0, 0, 90), (.01, 10, 10)), S(1, 0, (0, 5, -5), (0, 90, 0), (.01, 10, 10)),
                     S(1, 0, (-5, 5, 0), (0, 0, 0), (.01, 10, 10)), S(1, 0, (5, 5, 0), (0, 0, 0), (.01, 10, 10))], 5, 0),
        "room_tilted.txt": ([S(1, 0, (0, 0, 0), (3, 17, -2), (12, .05, 12)), S(1, 0, (0, 10, 0), (-4, 10, 2), (12, .05, 12)), S(1, 0, (0, 5, -5.5), (85, 3, 12), (12, .05, 11)),
                             S(1, 0, (-5.5, 5, 0), (0, 15, 88), (11, .05, 12)), S(1, 0, (5.5, 5, 0), (7, -12, 93), (11, .05, 12)),
                             S(1, 0, (0.3, 9.4, -0.5), (6, 20, -4), (3.5, .3, 3))], 0, 5),          # (the small tilted light keeps the slab test)
        "one wall turned": ([S(1, 0, (0, 0, 0), (0, 0, 0), (10, .01, 10)), S(1, 0, (0, 10, 0), (0, 0, 0), (10, .01, 10)), S(1, 0, (0, 5, -5), (80, 10, 0), (10, .05, 10)),
                             S(1, 0, (-5, 5, 0), (0, 0, 90), (10, .01, 10)), S(1, 0, (5, 5, 0), (0, 0, 90), (10, .01, 10))], 4, 1),
    }


def test_rotated_walls_get_the_plane_of_their_inner_face(pt, oracle):
    for name, (geoms, want_slots, want_planes) in _rooms(oracle).items():
        g = np.concatenate(geoms).view(pt.GEOM_DTYPE)
        planes, wall_geom, nslot, nwalls = pt.test_wall_planes(g)
        assert (nslot, len(planes), nwalls) == (want_slots, want_planes, len(geoms)), name
        assert sorted(wall_geom.tolist()) == list(range(len(geoms))), name
        cubes = [_corners(oracle, g[i]) for i in range(len(g))]
        allc = np.concatenate(cubes)
        middle = 0.5 * (allc.min(0) + allc.max(0))
        for w, (nx, ny, nz, th, far) in enumerate(planes.astype(np.float64)):
            n = np.array([nx, ny, nz])
            own = cubes[wall_geom[nslot + w]]
            assert abs(np.linalg.norm(n) - 1.0) < 1e-6, name
            # the wall's cube lies on the far side of its plane, with the inflation and the slack in between -- and not much more than that
            gap = th - (own @ n).max()
            assert 1e-5 < gap < 2e-3, (name, w, gap)
            # ... every wall's cube inside the half-space the segment is cut by
            assert (allc @ n).min() - far > 1e-5, (name, w)
            assert (allc @ n).min() - far < 5e-3, (name, w)
            # ... and the middle of the room on the side the certificate is issued for
            assert middle @ n > th + 1.0, (name, w)
            # the plane IS a face of the cube: four of its corners share the largest n . corner
            d = np.sort(own @ n)
            # (the normal is a float: the four agree to its rounding)
            assert d[-1] - d[-4] < 1e-5 and d[-4] - d[-5] > 1e-3, (name, w)


def test_axis_aligned_rooms_keep_their_slots(pt, oracle, monkeypatch):
    # the switch the experiments use: round 4's behaviour (world boxes and axis slots only)
    geoms, _, _ = _rooms(oracle)["room_tilted.txt"]
    g = np.concatenate(geoms).view(pt.GEOM_DTYPE)
    monkeypatch.setenv("PT_AMD_NO_ORIENTED_WALLS", "1")
    planes, wall_geom, nslot, nwalls = pt.test_wall_planes(g)
    assert len(planes) == 0 and nwalls == 6
