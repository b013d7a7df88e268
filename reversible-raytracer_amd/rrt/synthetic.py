"""Synthetic scenes for benchmarks and parity tests (NumPy only).

Each generator returns plain arrays in the layout of ``include/rrt.h``:

    w2o      float32 [S,3,4]  rows 0..2 of ``shape.w2o.m``
    material float32 [S,7]    r,g,b,ks,kd,ka,shininess
    light    float32 [6]      direction xyz, intensity rgb
    kinds    uint8   [S]      0 sphere, 1 square

Scene recipes follow SURVEY.md section 8(d); literal scenes quote the reference
drivers they restate.
"""
import numpy as np


def _w2o_translate_scale(c, rho):
    """inverse of translate(c)*scale(rho): scale(1/rho)*translate(-c)
    (transform.py:35-38, 60-93)."""
    c = np.asarray(c, dtype=np.float64)
    rho = np.broadcast_to(np.asarray(rho, dtype=np.float64), (3,))
    m = np.zeros((3, 4))
    m[:, :3] = np.diag(1.0 / rho)
    m[:, 3] = -c / rho
    return m


def random_spheres(S, seed=0):
    """S random spheres inside the +-0.5 view frustum (SURVEY 8(d) 'Synthetic
    inputs'): z~U(4,12), c_xy~U(-.4,.4)*z, rho~U(.15,.45); materials around
    optimize_brightness.py:24-25; light of optimize_brightness.py:35."""
    rs = np.random.RandomState(seed)
    w2o = np.zeros((S, 3, 4))
    mat = np.zeros((S, 7))
    for k in range(S):
        z = rs.uniform(4.0, 12.0)
        cx = rs.uniform(-0.4, 0.4) * z
        cy = rs.uniform(-0.4, 0.4) * z
        rho = rs.uniform(0.15, 0.45)
        w2o[k] = _w2o_translate_scale((cx, cy, z), rho)
        mat[k, 0:3] = rs.uniform(0.1, 0.9, size=3)
        mat[k, 3] = 0.3
        mat[k, 4] = rs.uniform(0.6, 0.9)
        mat[k, 5] = rs.uniform(0.3, 0.5)
        mat[k, 6] = 50.0
    light = np.array([-1.0, -1.0, 2.0, 0.961, 1.0, 0.87])
    return (w2o.astype(np.float32), mat.astype(np.float32), light.astype(np.float32),
            np.zeros(S, dtype=np.uint8))


def one_sphere():
    """BASELINE config 2: one unit sphere at (0,0,4), Phong."""
    w2o = _w2o_translate_scale((0.0, 0.0, 4.0), 1.0)[None]
    mat = np.array([[0.2, 0.9, 0.4, 0.3, 0.7, 0.5, 50.0]])
    light = np.array([-1.0, -1.0, 2.0, 0.961, 1.0, 0.87])
    return (w2o.astype(np.float32), mat.astype(np.float32), light.astype(np.float32),
            np.zeros(1, dtype=np.uint8))


def match_mirror(with_square=True):
    """The literal scene of match_mirror.py:16-29: two unit spheres at
    (-.5,-.5,4), (.5,.5,4) and a Square translate((0,0,3))*rotate(50,[0,1,0])."""
    m1 = [0.2, 0.9, 0.4, 0.3, 0.7, 0.5, 50.0]
    m2 = [0.87, 0.1, 0.507, 0.3, 0.9, 0.4, 50.0]
    w = [_w2o_translate_scale((-0.5, -0.5, 4.0), 1.0), _w2o_translate_scale((0.5, 0.5, 4.0), 1.0)]
    mats = [m1, m2]
    kinds = [0, 0]
    if with_square:
        a = np.deg2rad(50.0)
        s, c = np.sin(a), np.cos(a)
        R = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])   # rotate(50,(0,1,0)) transform.py:95-122
        m = np.zeros((3, 4))
        m[:, :3] = R.T                                       # (T R)^-1 = R^T T^-1
        m[:, 3] = -R.T @ np.array([0.0, 0.0, 3.0])
        w.append(m)
        mats.append(m2)
        kinds.append(1)
    light = np.array([-1.0, -1.0, 2.0, 1.0, 0.87, 0.961])
    return (np.stack(w).astype(np.float32), np.array(mats, dtype=np.float32),
            light.astype(np.float32), np.array(kinds, dtype=np.uint8))


def orbit_scene(center1=(3.83069295, -8.1440648, 32.0)):
    """orbit_experiments/planet_orbit.py:21-39: green sphere translate(c1)*scale(4),
    red sphere translate((0,0,48))*scale(6), light (0,0,1)/(1,1,1); orbit
    PhongShader (no specular); cameras translate((0,+-2.5,0))."""
    w = np.stack([_w2o_translate_scale(center1, 4.0), _w2o_translate_scale((0.0, 0.0, 48.0), 6.0)])
    mat = np.array([[0.0, 0.9, 0.0, 0.3, 0.7, 0.5, 50.0], [0.9, 0.0, 0.0, 0.3, 0.9, 0.4, 50.0]])
    light = np.array([0.0, 0.0, 1.0, 1.0, 1.0, 1.0])
    return (w.astype(np.float32), mat.astype(np.float32), light.astype(np.float32),
            np.zeros(2, dtype=np.uint8))


def orbit_camera(y):
    """Camera(x, y, translate((0,y,0)), [0,0,1]) -> cam[16] of rrt.h."""
    cam = np.array([1, 0, 0, 0, 0, 1, 0, y, 0, 0, 1, 0, 0, 0, 1, 0], dtype=np.float32)
    return cam
