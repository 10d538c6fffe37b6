"""2-D pose-graph optimisation (reference MapBuilder::OptimizeMap -> Ceres, src/optimization_2d/pose_graph_2d.cc,
include/optimization_2d/pose_graph_2d_error_term.h:62-95): the product's own Levenberg-Marquardt
(ni-slam_amd/csrc/kcc_posegraph.cpp, host code) against an independent restatement of the same least-squares problem
solved by scipy.optimize.least_squares.  Ceres itself is absent here (parity unpinned): the check is that both reach
the same minimum of the reference's cost, not Ceres' iterates."""
import math

import numpy as np
import pytest
import scipy.optimize as so

from kcc_helpers import nik
from ref_tracker import compute_absolute_pose, compute_relative_pose, normalize_angle


def residuals(flat, ids, fixed_pose, cons):
    """the reference's residual vector for free poses `flat` (every id except 0), restated with numpy"""
    poses = {0: fixed_pose}
    k = 0
    for i in ids:
        if i != 0:
            poses[i] = flat[3 * k: 3 * k + 3]; k += 1
    out = []
    for a, b, x, y, yaw, info in cons:
        pa, pb = poses[a], poses[b]
        c, s = math.cos(pa[2]), math.sin(pa[2])
        Rt = np.array([[c, s], [-s, c]])
        e = np.zeros(3)
        e[:2] = Rt @ (np.asarray(pb[:2]) - np.asarray(pa[:2])) - np.array([x, y])
        e[2] = normalize_angle((pb[2] - pa[2]) - yaw)
        out.append(np.linalg.cholesky(np.asarray(info, float).reshape(3, 3)) @ e)      # information.llt().matrixL()
    return np.concatenate(out)


def jacobian(flat, ids, fixed_pose, cons):
    """analytic Jacobian of `residuals` (dense), written from the error term, not from the product's code"""
    col = {i: 3 * k for k, i in enumerate(i for i in ids if i != 0)}
    poses = {0: fixed_pose}
    for i, c0 in col.items():
        poses[i] = flat[c0:c0 + 3]
    J = np.zeros((3 * len(cons), len(flat)))
    for e, (a, b, x, y, yaw, info) in enumerate(cons):
        pa, pb = poses[a], poses[b]
        c, s = math.cos(pa[2]), math.sin(pa[2])
        d = np.asarray(pb[:2]) - np.asarray(pa[:2])
        Rt = np.array([[c, s], [-s, c]]); dRt = np.array([[-s, c], [-c, -s]])
        Ea = np.zeros((3, 3)); Ea[:2, :2] = -Rt; Ea[:2, 2] = dRt @ d; Ea[2, 2] = -1
        Eb = np.zeros((3, 3)); Eb[:2, :2] = Rt; Eb[2, 2] = 1
        L = np.linalg.cholesky(np.asarray(info, float).reshape(3, 3))
        if a != 0:
            J[3 * e:3 * e + 3, col[a]:col[a] + 3] = L @ Ea
        if b != 0:
            J[3 * e:3 * e + 3, col[b]:col[b] + 3] = L @ Eb
    return J


def make_graph(n, seed, loops, noise=(0.02, 0.02, 0.01), info_scale=1.0):
    """a wandering trajectory with noisy odometry edges i -> i+1 and a few (noisy) loop edges; initial guess = odometry"""
    rng = np.random.default_rng(seed)
    truth = [np.zeros(3)]
    for i in range(1, n):
        truth.append(compute_absolute_pose(truth[-1], np.array([rng.uniform(0.2, 0.5), rng.uniform(-0.1, 0.1), rng.uniform(-0.4, 0.5)])))
    cons = []
    def info():
        A = rng.normal(size=(3, 3)) * 0.2
        return (np.diag([50.0, 60.0, 120.0]) + A @ A.T) * info_scale
    for i in range(n - 1):
        rel = compute_relative_pose(truth[i], truth[i + 1]) + rng.normal(0, noise)
        cons.append((i, i + 1, rel[0], rel[1], rel[2], info()))
    for a, b in loops:
        rel = compute_relative_pose(truth[a], truth[b]) + rng.normal(0, noise)
        cons.append((a, b, rel[0], rel[1], rel[2], info()))
    guess = [np.zeros(3)]
    for i in range(n - 1):
        guess.append(compute_absolute_pose(guess[-1], np.array(cons[i][2:5])))
    return list(range(n)), np.array(guess), cons


@pytest.mark.parametrize("n,loops,seed", [(12, [(0, 11), (3, 9)], 1), (60, [(0, 59), (10, 40), (25, 50), (5, 55)], 2),
                                          (300, [(0, 299), (20, 180), (100, 250), (50, 290), (130, 140)], 3)])
def test_matches_independent_least_squares(n, loops, seed):
    N = nik()
    ids, guess, cons = make_graph(n, seed, loops)
    got, sm = N.pose_graph_optimize(ids, guess, cons)
    assert sm["termination"] == 0, sm
    assert sm["final_cost"] < sm["initial_cost"]
    assert np.array_equal(got[0], guess[0])                                       # the base frame is constant
    x0 = guess[1:].reshape(-1)
    ref = so.least_squares(residuals, x0, jac=jacobian, args=(ids, guess[0], cons), method="trf", xtol=1e-14, ftol=1e-14, gtol=1e-12)
    assert 0.5 * float(np.sum(residuals(got[1:].reshape(-1), ids, guess[0], cons) ** 2)) == pytest.approx(sm["final_cost"], rel=1e-12)
    assert sm["final_cost"] == pytest.approx(ref.cost, rel=1e-5)
    want = ref.x.reshape(-1, 3)
    assert np.abs(got[1:, :2] - want[:, :2]).max() < 2e-3
    assert max(abs(normalize_angle(a - b)) for a, b in zip(got[1:, 2], want[:, 2])) < 1e-3
    assert all(-math.pi <= a < math.pi for a in got[1:, 2])                        # AngleLocalParameterization keeps yaw normalised


def test_known_answers_and_errors():
    N = nik()
    I = np.eye(3)
    # a consistent graph is already optimal: nothing moves, zero cost
    ids, guess, cons = make_graph(8, 5, [(0, 7)], noise=(0, 0, 0))
    got, sm = N.pose_graph_optimize(ids, guess, cons)
    assert sm["initial_cost"] < 1e-20 and np.allclose(got, guess, atol=1e-9)
    # two poses, one constraint: the free pose lands exactly on the measurement (ids need not be contiguous or sorted)
    got, sm = N.pose_graph_optimize([7, 0], [[5.0, 5.0, 1.0], [1.0, 2.0, math.pi / 2]], [(0, 7, 2.0, 0.0, 0.5, I)])
    assert sm["termination"] == 0 and sm["final_cost"] < 1e-12
    assert got[0] == pytest.approx([1.0, 4.0, normalize_angle(math.pi / 2 + 0.5)], abs=1e-6)
    # no constraints: no problem (pose_graph_2d.cc:58-61)
    got, sm = N.pose_graph_optimize([0, 1], [[0, 0, 0], [1, 0, 0]], [])
    assert sm["termination"] == 3 and np.array_equal(got, [[0, 0, 0], [1, 0, 0]])
    # the reference CHECK-fails on these; here they are reported
    with pytest.raises(N.NikError):
        N.pose_graph_optimize([1, 2], [[0, 0, 0], [1, 0, 0]], [(1, 2, 1, 0, 0, I)])            # no pose 0
    with pytest.raises(N.NikError):
        N.pose_graph_optimize([0, 1], [[0, 0, 0], [1, 0, 0]], [(0, 5, 1, 0, 0, I)])            # unknown pose id
    with pytest.raises(N.NikError):
        N.pose_graph_optimize([0, 1], [[0, 0, 0], [1, 0, 0]], [(0, 1, 1, 0, 0, -I)])           # information not positive definite
    # poses that no constraint touches are left alone
    got, sm = N.pose_graph_optimize([0, 1, 2], [[0, 0, 0], [1.2, 0.1, 0], [9, 9, 9]], [(0, 1, 1, 0, 0, I)])
    assert got[2] == pytest.approx([9, 9, 9]) and got[1] == pytest.approx([1, 0, 0], abs=1e-6)


# ---- the same problem linearised on the device (kcc_posegraph_dev.hip) ----------------------------------------------

def test_host_linearisation_matches_the_independent_jacobian():
    """nik_pose_graph_linearize on the host against the numpy residual / Jacobian written from the error term"""
    N = nik()
    ids, guess, cons = make_graph(40, 8, [(0, 39), (7, 30)])
    cost, g, d = N.pose_graph_linearize(ids, guess, cons, device=-1)
    r = residuals(guess[1:].reshape(-1), ids, guess[0], cons)
    J = jacobian(guess[1:].reshape(-1), ids, guess[0], cons)
    assert cost == pytest.approx(0.5 * float(r @ r), rel=1e-13)
    assert np.allclose(g[1:].reshape(-1), J.T @ r, rtol=1e-11, atol=1e-12) and np.all(g[0] == 0)
    H = J.T @ J
    for k in range(1, 40):
        assert np.allclose(d[k], H[3 * (k - 1):3 * k, 3 * (k - 1):3 * k], rtol=1e-11, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("n,loops,seed", [(12, [(0, 11), (3, 9)], 1), (300, [(0, 299), (20, 180), (100, 250), (50, 290), (130, 140)], 3),
                                          (1500, [(0, 1499), (100, 900), (400, 1300), (50, 1450), (700, 720), (10, 600)], 7)])
def test_device_linearisation_and_solve_match_the_host(n, loops, seed):
    """device = 0: residuals, normal equations AND the damped solve (block-Jacobi PCG in one workgroup, k_pg_pcg) on the GPU;
    device = -1: all on the host (dense Cholesky up to 256 free poses, PCG beyond)"""
    N = nik()
    ids, guess, cons = make_graph(n, seed, loops)
    ch, gh, dh = N.pose_graph_linearize(ids, guess, cons, device=-1)
    cd, gd, dd = N.pose_graph_linearize(ids, guess, cons, device=0)
    # the DEVICE linearisation against the independent numpy residual / Jacobian written from the reference's error term
    # (pose_graph_2d_error_term.h:62-116) -- not only against the product's own host code
    r = residuals(guess[1:].reshape(-1), ids, guess[0], cons)
    J = jacobian(guess[1:].reshape(-1), ids, guess[0], cons)
    assert cd == pytest.approx(0.5 * float(r @ r), rel=1e-12)
    assert np.allclose(gd[1:].reshape(-1), J.T @ r, rtol=1e-10, atol=1e-11) and np.all(gd[0] == 0)
    Hn = J.T @ J
    for k in range(1, n):
        assert np.allclose(dd[k], Hn[3 * (k - 1):3 * k, 3 * (k - 1):3 * k], rtol=1e-10, atol=1e-11)
    # same double arithmetic; only the order of the sums over a pose's constraints / over the constraints may differ
    assert cd == pytest.approx(ch, rel=1e-13)
    assert np.allclose(gd, gh, rtol=1e-12, atol=1e-13) and np.allclose(dd, dh, rtol=1e-12, atol=1e-13)
    got_h, sm_h = N.pose_graph_optimize(ids, guess, cons)
    got_d, sm_d = N.pose_graph_optimize(ids, guess, cons, device=0)
    assert sm_d["termination"] == sm_h["termination"] == 0 and sm_d["iterations"] == sm_h["iterations"]
    assert sm_d["final_cost"] == pytest.approx(sm_h["final_cost"], rel=1e-9)
    assert np.allclose(got_d, got_h, rtol=0, atol=1e-8)
    # reproducible: the device sums have a fixed order
    cd2, gd2, dd2 = N.pose_graph_linearize(ids, guess, cons, device=0)
    assert cd2 == cd and np.array_equal(gd2, gd) and np.array_equal(dd2, dd)


@pytest.mark.gpu
def test_sharded_cost_through_the_group():
    """constraints sharded like the frame pairs that produced them: each shard's cost is reduced on its device and the group
    adds one double per member (RCCL between GPUs; here one GPU, with and without the RCCL call path)"""
    import os
    import subprocess
    import sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, "tests")
from kcc_helpers import SMALL, nik
from test_posegraph import make_graph
N = nik()
ids, guess, cons = make_graph(120, 4, [(0, 119), (10, 80), (33, 99)])
whole, _, _ = N.pose_graph_linearize(ids, guess, cons, device=-1)
g = SMALL
grp = N.Group.local(N.default_config(rotation_divisor=g["PD"], rotation_channel=g["PC"]), g["H"], g["W"], max_batch=2, max_frames=4, devices=(0,))
total = 0.0
for b, e in (N.Group.shard(len(cons), 3, r) for r in range(3)):          # three shards, evaluated one after the other on the one GPU
    sh = N.PgShard(0, ids, guess, cons[b:e])
    total += grp.pose_graph_cost([sh])
    sh.close()
assert abs(total - whole) <= 1e-12 * whole, (total, whole)
moved = guess.copy(); moved[5] += [0.01, -0.02, 0.003]
sh = N.PgShard(0, ids, guess, cons)
c2 = grp.pose_graph_cost([sh], moved)
want, _, _ = N.pose_graph_linearize(ids, moved, cons, device=-1)
assert abs(c2 - want) <= 1e-12 * want and c2 != whole
sh.close(); grp.close()
print("PG-GROUP-OK", grp.__class__.__name__)
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for force in ("0", "1"):
        env = dict(os.environ)
        if force == "1":
            env["NIK_GROUP_FORCE_RCCL"] = "1"
        p = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0 and "PG-GROUP-OK" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]
