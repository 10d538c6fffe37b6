// kcc_posegraph.cpp -- 2-D pose-graph optimisation: the problem the reference hands to Ceres in
// MapBuilder::OptimizeMap (src/map_builder.cc:195-271) through BuildOptimizationProblem /
// SolveOptimizationProblem (src/optimization_2d/pose_graph_2d.cc:53-109,187-200), solved here by an own
// Levenberg-Marquardt (no Ceres in this stack).
//
//   residual(a -> b) = L * [ R(yaw_a)^T (p_b - p_a) - p_ab ;  Normalize(yaw_b - yaw_a - yaw_ab) ],   L = chol_lower(information)
//                                                    (include/optimization_2d/pose_graph_2d_error_term.h:62-95)
//   yaw updates go through NormalizeAngle (angle_local_parameterization.h:42-52); the pose with id 0 is constant
//   (pose_graph_2d.cc:103-108); no loss function; at most max_iterations (300 in the reference) iterations.
//
// Same trust-region scheme as Ceres' default LM (Jacobi-scaled damping D = sqrt(diag(J^T J)) clamped to
// [1e-6, 1e32], initial radius 1e4, step accepted if rho > 1e-3, radius /= max(1/3, 1 - (2 rho - 1)^3) on success
// and halved with a doubling factor on failure, function / gradient / parameter tolerances 1e-6 / 1e-10 / 1e-8),
// so it converges to the same minimum; iterates are not Ceres' bit for bit.  The normal equations are solved by
// dense Cholesky for small graphs and block-Jacobi preconditioned conjugate gradients for large ones.
// Per keyframe insertion with >= 2 loop matches, not per frame: host code, double precision.
#include "../../include/nislam_kcc.h"
#include "kcc_posegraph_dev.h"
#include "kcc_tune.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

namespace {

inline double normalize_angle(double a) { const double two_pi = 2.0 * M_PI; return a - two_pi * std::floor((a + M_PI) / two_pi); }

struct Edge { int a, b; double m[3]; double L[9]; };            // pose indices, measurement, sqrt information (lower)

// 3x3 lower Cholesky factor (Eigen's information.llt().matrixL()); false if not positive definite
bool chol3(const double A[9], double L[9]) {
    for (int i = 0; i < 9; ++i) L[i] = 0;
    for (int j = 0; j < 3; ++j) {
        double d = A[3 * j + j];
        for (int k = 0; k < j; ++k) d -= L[3 * j + k] * L[3 * j + k];
        if (!(d > 0)) return false;
        L[3 * j + j] = std::sqrt(d);
        for (int i = j + 1; i < 3; ++i) {
            double v = A[3 * i + j];
            for (int k = 0; k < j; ++k) v -= L[3 * i + k] * L[3 * j + k];
            L[3 * i + j] = v / L[3 * j + j];
        }
    }
    return true;
}

struct Problem {
    int n = 0;                       // poses
    std::vector<double> x;           // [n][3]
    std::vector<char> fixed;         // [n]
    std::vector<Edge> edges;
    std::vector<int> col;            // pose -> first column in the reduced system (-1 fixed)
    int dim = 0;

    // residuals (3 per edge) and, optionally, Jacobian blocks Ja, Jb (3x3 each, row-major) for the state xs
    void eval(const std::vector<double>& xs, std::vector<double>& r, std::vector<double>* J) const {
        r.resize(3 * edges.size());
        if (J) J->resize(18 * edges.size());
        for (size_t e = 0; e < edges.size(); ++e) {
            const Edge& E = edges[e];
            const double* pa = &xs[3 * E.a]; const double* pb = &xs[3 * E.b];
            const double c = std::cos(pa[2]), s = std::sin(pa[2]);
            const double dx = pb[0] - pa[0], dy = pb[1] - pa[1];
            const double err[3] = { c * dx + s * dy - E.m[0], -s * dx + c * dy - E.m[1], normalize_angle((pb[2] - pa[2]) - E.m[2]) };
            for (int i = 0; i < 3; ++i) r[3 * e + i] = E.L[3 * i] * err[0] + E.L[3 * i + 1] * err[1] + E.L[3 * i + 2] * err[2];
            if (J) {
                // d err / d (xa, ya, yawa) and d err / d (xb, yb, yawb)
                const double Ea[9] = { -c, -s, -s * dx + c * dy,   s, -c, -c * dx - s * dy,   0, 0, -1 };
                const double Eb[9] = { c, s, 0,   -s, c, 0,   0, 0, 1 };
                double* Ja = &(*J)[18 * e]; double* Jb = Ja + 9;
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j < 3; ++j) {
                        double va = 0, vb = 0;
                        for (int k = 0; k < 3; ++k) { va += E.L[3 * i + k] * Ea[3 * k + j]; vb += E.L[3 * i + k] * Eb[3 * k + j]; }
                        Ja[3 * i + j] = va; Jb[3 * i + j] = vb;
                    }
            }
        }
    }
    static double cost(const std::vector<double>& r) { double s = 0; for (double v : r) s += v * v; return 0.5 * s; }
};

// Gauss-Newton system in 3x3 blocks: diagonal blocks per free pose, off-diagonal blocks per edge between free poses
struct Normal {
    int dim;
    std::vector<double> diag;                     // [free pose][9]
    std::vector<double> off;                      // [edge][9]: block (col(a), col(b)) = Ja^T Jb
    std::vector<double> g;                        // J^T r
};

void build_normal(const Problem& P, const std::vector<double>& r, const std::vector<double>& J, Normal& N) {
    N.dim = P.dim;
    N.diag.assign((size_t)P.dim * 3, 0.0); N.diag.resize((size_t)(P.dim / 3) * 9, 0.0);
    std::fill(N.diag.begin(), N.diag.end(), 0.0);
    N.off.assign(P.edges.size() * 9, 0.0); N.g.assign(P.dim, 0.0);
    for (size_t e = 0; e < P.edges.size(); ++e) {
        const Edge& E = P.edges[e];
        const double* Ja = &J[18 * e]; const double* Jb = Ja + 9; const double* re = &r[3 * e];
        const int ca = P.col[E.a], cb = P.col[E.b];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double aa = 0, bb = 0, ab = 0;
                for (int k = 0; k < 3; ++k) { aa += Ja[3 * k + i] * Ja[3 * k + j]; bb += Jb[3 * k + i] * Jb[3 * k + j]; ab += Ja[3 * k + i] * Jb[3 * k + j]; }
                if (ca >= 0) N.diag[(size_t)(ca / 3) * 9 + 3 * i + j] += aa;
                if (cb >= 0) N.diag[(size_t)(cb / 3) * 9 + 3 * i + j] += bb;
                if (ca >= 0 && cb >= 0) N.off[9 * e + 3 * i + j] = ab;
            }
        for (int i = 0; i < 3; ++i) {
            double ga = 0, gb = 0;
            for (int k = 0; k < 3; ++k) { ga += Ja[3 * k + i] * re[k]; gb += Jb[3 * k + i] * re[k]; }
            if (ca >= 0) N.g[ca + i] += ga;
            if (cb >= 0) N.g[cb + i] += gb;
        }
    }
}

// y = (H + diag(damp)) v
void apply(const Problem& P, const Normal& N, const std::vector<double>& damp, const std::vector<double>& v, std::vector<double>& y) {
    y.assign(N.dim, 0.0);
    for (int b = 0; b < N.dim / 3; ++b)
        for (int i = 0; i < 3; ++i) {
            double s = damp[3 * b + i] * v[3 * b + i];
            for (int j = 0; j < 3; ++j) s += N.diag[(size_t)b * 9 + 3 * i + j] * v[3 * b + j];
            y[3 * b + i] = s;
        }
    for (size_t e = 0; e < P.edges.size(); ++e) {
        const int ca = P.col[P.edges[e].a], cb = P.col[P.edges[e].b];
        if (ca < 0 || cb < 0) continue;
        const double* B = &N.off[9 * e];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) { y[ca + i] += B[3 * i + j] * v[cb + j]; y[cb + j] += B[3 * i + j] * v[ca + i]; }
    }
}

bool inv3(const double A[9], double I[9]) {
    const double d = A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
    if (d == 0 || !std::isfinite(d)) return false;
    const double id = 1.0 / d;
    I[0] = (A[4] * A[8] - A[5] * A[7]) * id; I[1] = (A[2] * A[7] - A[1] * A[8]) * id; I[2] = (A[1] * A[5] - A[2] * A[4]) * id;
    I[3] = (A[5] * A[6] - A[3] * A[8]) * id; I[4] = (A[0] * A[8] - A[2] * A[6]) * id; I[5] = (A[2] * A[3] - A[0] * A[5]) * id;
    I[6] = (A[3] * A[7] - A[4] * A[6]) * id; I[7] = (A[1] * A[6] - A[0] * A[7]) * id; I[8] = (A[0] * A[4] - A[1] * A[3]) * id;
    return true;
}

// (H + diag(damp)) step = -g : dense Cholesky (small systems) or block-Jacobi PCG
bool solve(const Problem& P, const Normal& N, const std::vector<double>& damp, std::vector<double>& step) {
    const int n = N.dim;
    step.assign(n, 0.0);
    if (n == 0) return true;
    if (n <= 768) {
        std::vector<double> A((size_t)n * n, 0.0);
        for (int b = 0; b < n / 3; ++b)
            for (int i = 0; i < 3; ++i) {
                for (int j = 0; j < 3; ++j) A[(size_t)(3 * b + i) * n + 3 * b + j] = N.diag[(size_t)b * 9 + 3 * i + j];
                A[(size_t)(3 * b + i) * n + 3 * b + i] += damp[3 * b + i];
            }
        for (size_t e = 0; e < P.edges.size(); ++e) {
            const int ca = P.col[P.edges[e].a], cb = P.col[P.edges[e].b];
            if (ca < 0 || cb < 0) continue;
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) { A[(size_t)(ca + i) * n + cb + j] += N.off[9 * e + 3 * i + j]; A[(size_t)(cb + j) * n + ca + i] += N.off[9 * e + 3 * i + j]; }
        }
        // in-place lower Cholesky
        for (int j = 0; j < n; ++j) {
            double d = A[(size_t)j * n + j];
            for (int k = 0; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
            if (!(d > 0)) return false;
            const double l = std::sqrt(d);
            A[(size_t)j * n + j] = l;
            for (int i = j + 1; i < n; ++i) {
                double v = A[(size_t)i * n + j];
                for (int k = 0; k < j; ++k) v -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
                A[(size_t)i * n + j] = v / l;
            }
        }
        for (int i = 0; i < n; ++i) { double v = -N.g[i]; for (int k = 0; k < i; ++k) v -= A[(size_t)i * n + k] * step[k]; step[i] = v / A[(size_t)i * n + i]; }
        for (int i = n - 1; i >= 0; --i) { double v = step[i]; for (int k = i + 1; k < n; ++k) v -= A[(size_t)k * n + i] * step[k]; step[i] = v / A[(size_t)i * n + i]; }
        return true;
    }
    // preconditioner: inverse of the damped diagonal blocks
    std::vector<double> Minv((size_t)(n / 3) * 9);
    for (int b = 0; b < n / 3; ++b) {
        double A[9];
        for (int i = 0; i < 9; ++i) A[i] = N.diag[(size_t)b * 9 + i];
        for (int i = 0; i < 3; ++i) A[4 * i] += damp[3 * b + i];
        if (!inv3(A, &Minv[(size_t)b * 9])) return false;
    }
    auto precond = [&](const std::vector<double>& r, std::vector<double>& z) {
        z.resize(n);
        for (int b = 0; b < n / 3; ++b)
            for (int i = 0; i < 3; ++i) z[3 * b + i] = Minv[(size_t)b * 9 + 3 * i] * r[3 * b] + Minv[(size_t)b * 9 + 3 * i + 1] * r[3 * b + 1] + Minv[(size_t)b * 9 + 3 * i + 2] * r[3 * b + 2];
    };
    std::vector<double> r(n), z, p, Ap;
    double bnorm = 0;
    for (int i = 0; i < n; ++i) { r[i] = -N.g[i]; bnorm += r[i] * r[i]; }
    if (bnorm == 0) return true;
    precond(r, z); p = z;
    double rz = 0; for (int i = 0; i < n; ++i) rz += r[i] * z[i];
    for (int it = 0; it < 20 * n; ++it) {
        apply(P, N, damp, p, Ap);
        double pAp = 0; for (int i = 0; i < n; ++i) pAp += p[i] * Ap[i];
        if (!(pAp > 0)) return false;
        const double alpha = rz / pAp;
        double rn = 0;
        for (int i = 0; i < n; ++i) { step[i] += alpha * p[i]; r[i] -= alpha * Ap[i]; rn += r[i] * r[i]; }
        if (rn <= 1e-24 * bnorm) break;
        precond(r, z);
        double rz2 = 0; for (int i = 0; i < n; ++i) rz2 += r[i] * z[i];
        const double beta = rz2 / rz; rz = rz2;
        for (int i = 0; i < n; ++i) p[i] = z[i] + beta * p[i];
    }
    return true;
}

}  // namespace

// the problem BuildOptimizationProblem hands to Ceres (pose_graph_2d.cc:53-109), from the C ABI's arrays
static int setup(int n_poses, const int32_t* ids, const double* poses, int n_constraints, const nik_pg_constraint* cons, Problem& P) {
    std::map<int, int> index;
    for (int i = 0; i < n_poses; ++i) if (!index.emplace(ids[i], i).second) return NIK_ERR_INVALID_ARG;
    if (!index.count(0)) return NIK_ERR_INVALID_ARG;                               // CHECK(baseframe_pose_iter != poses->end())  (:104)
    P.n = n_poses; P.x.assign(poses, poses + (size_t)3 * n_poses); P.fixed.assign(n_poses, 0);
    P.fixed[index[0]] = 1;
    P.edges.resize(n_constraints);
    std::vector<char> used(n_poses, 0);
    for (int e = 0; e < n_constraints; ++e) {
        const auto ia = index.find(cons[e].id_begin), ib = index.find(cons[e].id_end);
        if (ia == index.end() || ib == index.end()) return NIK_ERR_INVALID_ARG;    // CHECK "Pose with ID ... not found" (:74-79)
        Edge& E = P.edges[e];
        E.a = ia->second; E.b = ib->second; E.m[0] = cons[e].x; E.m[1] = cons[e].y; E.m[2] = cons[e].yaw_radians;
        if (!chol3(cons[e].information, E.L)) return NIK_ERR_INVALID_ARG;
        used[E.a] = used[E.b] = 1;
    }
    // poses that no constraint touches are not part of the problem (Ceres never sees them)
    P.col.assign(n_poses, -1);
    P.dim = 0;
    for (int i = 0; i < n_poses; ++i) if (used[i] && !P.fixed[i]) { P.col[i] = P.dim; P.dim += 3; }
    return NIK_OK;
}

// the problem's device twin (kcc_posegraph_dev.hip): residuals, J^T J blocks and J^T r evaluated on `device`
static kcc_pg::DevProblem* to_device(const Problem& P, int device, std::string& err) {
    std::vector<kcc_pg::DevEdge> de(P.edges.size());
    for (size_t e = 0; e < P.edges.size(); ++e) {
        const Edge& E = P.edges[e];
        de[e].a = E.a; de[e].b = E.b; de[e].ca = P.col[E.a]; de[e].cb = P.col[E.b];
        for (int k = 0; k < 3; ++k) de[e].m[k] = E.m[k];
        for (int k = 0; k < 9; ++k) de[e].L[k] = E.L[k];
    }
    return kcc_pg::dev_create(device, P.n, P.dim, de, P.col, err);
}

// device < 0: residuals, normal equations and the damped solve on the host; else all three on that HIP device
// ($NIK_PG_HOST_SOLVE=1: only the linearisation on the device, the solve on the host as in round 3)
static int optimize(int device, int n_poses, const int32_t* ids, double* poses, int n_constraints,
                    const nik_pg_constraint* cons, int max_iterations, nik_pg_summary* summary) {
    if (n_poses < 0 || n_constraints < 0 || (n_poses > 0 && (!ids || !poses)) || (n_constraints > 0 && !cons)) return NIK_ERR_INVALID_ARG;
    nik_pg_summary sm{}; sm.termination = NIK_PG_NO_CONSTRAINTS;
    if (n_constraints == 0) { if (summary) *summary = sm; return NIK_OK; }        // "No constraints, no problem to optimize." (pose_graph_2d.cc:58-61)
    Problem P;
    int rc = setup(n_poses, ids, poses, n_constraints, cons, P);
    if (rc) return rc;
    kcc_pg::DevProblem* dev = nullptr;
    if (device >= 0) {
        std::string err;
        dev = to_device(P, device, err);
        if (!dev) return NIK_ERR_HIP;
    }
    struct Guard { kcc_pg::DevProblem* d; ~Guard() { kcc_pg::dev_destroy(d); } } guard{ dev };
    const bool host_solve = kcc::tune_env("NIK_PG_HOST_SOLVE") && atoi(kcc::tune_env("NIK_PG_HOST_SOLVE")) != 0;
    // linearise at xs: residual cost, and (want_normal) the Gauss-Newton blocks into N
    std::vector<double> r_, J_;
    auto linearize = [&](const std::vector<double>& xs, double& cost_out, Normal* N) -> bool {
        if (dev) {
            if (N) {
                N->dim = P.dim; N->diag.assign((size_t)(P.dim / 3) * 9, 0.0); N->off.assign(P.edges.size() * 9, 0.0); N->g.assign(P.dim, 0.0);
                return kcc_pg::dev_linearize(dev, xs.data(), &cost_out, nullptr, N->diag.data(), N->off.data(), N->g.data()) == 0;
            }
            return kcc_pg::dev_linearize(dev, xs.data(), &cost_out, nullptr, nullptr, nullptr, nullptr) == 0;
        }
        P.eval(xs, r_, N ? &J_ : nullptr);
        cost_out = Problem::cost(r_);
        if (N) build_normal(P, r_, J_, *N);
        return true;
    };

    std::vector<double> step, damp(P.dim), xn;
    Normal N;
    double cost = 0;
    if (!linearize(P.x, cost, &N)) return NIK_ERR_HIP;
    sm.initial_cost = cost;
    double radius = 1e4, decrease = 2.0;
    sm.termination = NIK_PG_NO_CONVERGENCE;
    const int max_it = max_iterations > 0 ? max_iterations : 300;
    int it = 0;
    for (; it < max_it; ++it) {
        double gmax = 0; for (double v : N.g) gmax = std::max(gmax, std::fabs(v));
        if (gmax <= 1e-10) { sm.termination = NIK_PG_CONVERGENCE; break; }          // gradient_tolerance
        for (int b = 0; b < P.dim / 3; ++b)
            for (int i = 0; i < 3; ++i) {
                const double d2 = std::min(std::max(N.diag[(size_t)b * 9 + 4 * i], 1e-6 * 1e-6), 1e32 * 1e32);   // D^2, D clamped to [1e-6, 1e32]
                damp[3 * b + i] = d2 / radius;
            }
        bool solved;
        if (dev && !host_solve) {
            // the damped step on the device too (k_pg_pcg: the blocks of the current point are still there)
            step.assign(P.dim, 0.0);
            const int sr = kcc_pg::dev_solve(dev, damp.data(), step.data(), nullptr);
            if (sr < 0) return NIK_ERR_HIP;
            solved = sr == 0 || sr == 2;                  // 2: iteration limit before tolerance -- tried on its merits, and counted
            if (sr == 2) sm.inexact_solves += 1;
        } else {
            solved = solve(P, N, damp, step);
        }
        if (!solved) { radius /= decrease; decrease *= 2; if (radius < 1e-32) { sm.termination = NIK_PG_FAILURE; break; } continue; }
        // model decrease: -(g^T s + 0.5 s^T H s) with H = J^T J
        std::vector<double> Hs, zero(P.dim, 0.0);
        apply(P, N, zero, step, Hs);
        double gs = 0, sHs = 0, snorm = 0, xnorm = 0;
        for (int i = 0; i < P.dim; ++i) { gs += N.g[i] * step[i]; sHs += step[i] * Hs[i]; snorm += step[i] * step[i]; }
        const double model = -(gs + 0.5 * sHs);
        xn = P.x;
        for (int i = 0; i < P.n; ++i) {
            if (P.col[i] < 0) continue;
            const double* s = &step[P.col[i]];
            xnorm += xn[3 * i] * xn[3 * i] + xn[3 * i + 1] * xn[3 * i + 1] + xn[3 * i + 2] * xn[3 * i + 2];
            xn[3 * i] += s[0]; xn[3 * i + 1] += s[1]; xn[3 * i + 2] = normalize_angle(xn[3 * i + 2] + s[2]);   // AngleLocalParameterization
        }
        if (std::sqrt(snorm) <= 1e-8 * (std::sqrt(xnorm) + 1e-8)) { sm.termination = NIK_PG_CONVERGENCE; break; }   // parameter_tolerance
        double cn = 0;
        if (!linearize(xn, cn, nullptr)) return NIK_ERR_HIP;
        const double rho = model > 0 ? (cost - cn) / model : -1.0;
        if (rho > 1e-3) {                                                         // min_relative_decrease
            const double dc = cost - cn;
            P.x = xn;
            if (!linearize(P.x, cost, &N)) return NIK_ERR_HIP;
            const double t = 2.0 * rho - 1.0;
            radius = std::min(radius / std::max(1.0 / 3.0, 1.0 - t * t * t), 1e16);
            decrease = 2.0;
            ++sm.successful_steps;
            if (std::fabs(dc) <= 1e-6 * cost) { sm.termination = NIK_PG_CONVERGENCE; ++it; break; }   // function_tolerance
        } else {
            radius /= decrease; decrease *= 2;
            if (radius < 1e-32) { sm.termination = NIK_PG_FAILURE; break; }
        }
    }
    sm.iterations = it; sm.final_cost = cost;
    for (size_t i = 0; i < (size_t)3 * n_poses; ++i) poses[i] = P.x[i];
    if (summary) *summary = sm;
    return NIK_OK;
}

extern "C" {

int nik_pose_graph_optimize(int n_poses, const int32_t* ids, double* poses, int n_constraints,
                            const nik_pg_constraint* cons, int max_iterations, nik_pg_summary* summary) {
    return optimize(-1, n_poses, ids, poses, n_constraints, cons, max_iterations, summary);
}

int nik_pose_graph_optimize_dev(int device, int n_poses, const int32_t* ids, double* poses, int n_constraints,
                                const nik_pg_constraint* cons, int max_iterations, nik_pg_summary* summary) {
    if (device < 0) return NIK_ERR_INVALID_ARG;
    return optimize(device, n_poses, ids, poses, n_constraints, cons, max_iterations, summary);
}

// cost, gradient and J^T J diagonal blocks at `poses`, by pose (zero rows for the constant pose and for poses no constraint touches)
int nik_pose_graph_linearize(int device, int n_poses, const int32_t* ids, const double* poses, int n_constraints,
                             const nik_pg_constraint* cons, double* cost, double* gradient /*[n_poses][3]*/, double* jtj_diag /*[n_poses][9]*/) {
    if (n_poses <= 0 || n_constraints < 0 || !ids || !poses || (n_constraints > 0 && !cons)) return NIK_ERR_INVALID_ARG;
    Problem P;
    int rc = setup(n_poses, ids, poses, n_constraints, cons, P);
    if (rc) return rc;
    Normal N;
    double c = 0;
    if (device >= 0) {
        std::string err;
        kcc_pg::DevProblem* dev = to_device(P, device, err);
        if (!dev) return NIK_ERR_HIP;
        N.dim = P.dim; N.diag.assign((size_t)(P.dim / 3) * 9, 0.0); N.off.assign(P.edges.size() * 9, 0.0); N.g.assign(P.dim, 0.0);
        const int hr = kcc_pg::dev_linearize(dev, P.x.data(), &c, nullptr, N.diag.data(), N.off.data(), N.g.data());
        kcc_pg::dev_destroy(dev);
        if (hr) return NIK_ERR_HIP;
    } else {
        std::vector<double> r, J;
        P.eval(P.x, r, &J);
        c = Problem::cost(r);
        build_normal(P, r, J, N);
    }
    if (cost) *cost = c;
    for (int i = 0; i < n_poses; ++i) {
        const int col = P.col[i];
        if (gradient) for (int k = 0; k < 3; ++k) gradient[3 * i + k] = col >= 0 ? N.g[col + k] : 0.0;
        if (jtj_diag) for (int k = 0; k < 9; ++k) jtj_diag[9 * i + k] = col >= 0 ? N.diag[(size_t)(col / 3) * 9 + k] : 0.0;
    }
    return NIK_OK;
}

// a SHARD of the constraints evaluated on `device`, its cost left on the device: what nik_group_pose_graph_cost all-reduces
struct nik_pg_shard { kcc_pg::DevProblem* dev; std::vector<double> x; };
int nik_pg_shard_create(int device, int n_poses, const int32_t* ids, const double* poses, int n_constraints, const nik_pg_constraint* cons, nik_pg_shard** out) {
    if (!out || device < 0 || n_poses <= 0 || !ids || !poses || n_constraints < 0 || (n_constraints > 0 && !cons)) return NIK_ERR_INVALID_ARG;
    *out = nullptr;
    Problem P;
    int rc = setup(n_poses, ids, poses, n_constraints, cons, P);
    if (rc) return rc;
    std::string err;
    kcc_pg::DevProblem* dev = to_device(P, device, err);
    if (!dev) return NIK_ERR_HIP;
    *out = new nik_pg_shard{ dev, P.x };
    return NIK_OK;
}
void nik_pg_shard_destroy(nik_pg_shard* s) { if (s) { kcc_pg::dev_destroy(s->dev); delete s; } }
int nik_pg_shard_device(const nik_pg_shard* s) { return s ? kcc_pg::dev_device(s->dev) : NIK_ERR_INVALID_ARG; }
int nik_pg_shard_cost_dev(nik_pg_shard* s, const double* poses, double** d_cost, void** stream) {
    if (!s || !d_cost) return NIK_ERR_INVALID_ARG;
    if (poses) s->x.assign(poses, poses + s->x.size());
    return kcc_pg::dev_cost_async(s->dev, s->x.data(), d_cost, stream) ? NIK_ERR_HIP : NIK_OK;
}

}  // extern "C"
