// kcc_posegraph_dev.hip -- the pose graph's residuals and normal equations on the device.
//
// Reference: the cost BuildOptimizationProblem assembles for Ceres (src/optimization_2d/pose_graph_2d.cc:53-109) from
// PoseGraph2dErrorTerm (include/optimization_2d/pose_graph_2d_error_term.h:62-116):
//   r(a -> b) = L [ R(yaw_a)^T (p_b - p_a) - p_ab ;  NormalizeAngle(yaw_b - yaw_a - yaw_ab) ],  L = chol_lower(information).
// This is the one place of the pipeline where a "residual sum" is a real reduction (north star: "weighted-least-squares pose
// normal-equations reduction ... wavefront shuffles for the residual/JTJ reduction ... RCCL all-reduce only for the final
// pose-graph residual sum"): the constraints shard over GPUs like the frame pairs that produced them, every GPU reduces
// its shard here, and nik_group_pose_graph_cost all-reduces one double.
//
//   k_pg_edges   one thread per constraint: residual, the two 3x3 Jacobian blocks, Ja^T Jb, 0.5 |r|^2
//   k_pg_poses   one thread per free pose: J^T J diagonal block and J^T r over the pose's incident constraints, walked in
//                constraint order through a CSR incidence list (no atomics: the sums are reproducible)
//   k_pg_cost    one wave: per-lane partial sums in index order, then a fixed xor-shuffle butterfly
// All arithmetic is double, as in the reference (Ceres) and in the host solver.
#include "kcc_posegraph_dev.h"

#include <hip/hip_runtime.h>

#include <algorithm>

namespace kcc_pg {

namespace {

__device__ __forceinline__ double normalize_angle(double a) {        // optimization_2d/normalize_angle.h:42-46
    const double two_pi = 2.0 * M_PI;
    return a - two_pi * floor((a + M_PI) / two_pi);
}

__global__ __launch_bounds__(64) void k_pg_edges(int n_edges, const DevEdge* __restrict__ edges, const double* __restrict__ x,
                                                 double* __restrict__ r, double* __restrict__ J /*[E][18]*/, double* __restrict__ off /*[E][9]*/,
                                                 double* __restrict__ ecost, int want_normal) {
#pragma clang fp contract(off)
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (e >= n_edges) return;
    const DevEdge E = edges[e];
    const double* pa = x + 3 * E.a; const double* pb = x + 3 * E.b;
    const double c = cos(pa[2]), s = sin(pa[2]);
    const double dx = pb[0] - pa[0], dy = pb[1] - pa[1];
    const double err[3] = { c * dx + s * dy - E.m[0], -s * dx + c * dy - E.m[1], normalize_angle((pb[2] - pa[2]) - E.m[2]) };
    double re[3], cost = 0.0;
    for (int i = 0; i < 3; ++i) {
        re[i] = E.L[3 * i] * err[0] + E.L[3 * i + 1] * err[1] + E.L[3 * i + 2] * err[2];
        r[3 * e + i] = re[i];
    }
    for (int i = 0; i < 3; ++i) cost += re[i] * re[i];
    ecost[e] = cost;                                                   // (0.5 applied once, on the sum)
    // cost-only evaluations (the trial points of the Levenberg-Marquardt) must leave J and J^T J of the CURRENT point alone: the
    // damped solve on the device (k_pg_pcg) reads them again when a step is rejected
    if (!want_normal) return;
    const double Ea[9] = { -c, -s, -s * dx + c * dy,   s, -c, -c * dx - s * dy,   0, 0, -1 };
    const double Eb[9] = { c, s, 0,   -s, c, 0,   0, 0, 1 };
    double Ja[9], Jb[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double va = 0, vb = 0;
            for (int k = 0; k < 3; ++k) { va += E.L[3 * i + k] * Ea[3 * k + j]; vb += E.L[3 * i + k] * Eb[3 * k + j]; }
            Ja[3 * i + j] = va; Jb[3 * i + j] = vb;
            J[18 * e + 3 * i + j] = va; J[18 * e + 9 + 3 * i + j] = vb;
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double ab = 0;
            for (int k = 0; k < 3; ++k) ab += Ja[3 * k + i] * Jb[3 * k + j];
            off[9 * e + 3 * i + j] = (E.ca >= 0 && E.cb >= 0) ? ab : 0.0;
        }
}

// incidence list of free pose b: entries inc[first[b] .. first[b+1]) = 2 * edge + side (0: the pose is the edge's a, 1: its b)
__global__ __launch_bounds__(64) void k_pg_poses(int n_free, const int* __restrict__ first, const int* __restrict__ inc,
                                                 const double* __restrict__ r, const double* __restrict__ J,
                                                 double* __restrict__ diag /*[n_free][9]*/, double* __restrict__ g /*[3 n_free]*/) {
#pragma clang fp contract(off)
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= n_free) return;
    double D[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 }, G[3] = { 0, 0, 0 };
    for (int t = first[b]; t < first[b + 1]; ++t) {
        const int e = inc[t] >> 1, side = inc[t] & 1;
        const double* Jx = J + 18 * e + 9 * side; const double* re = r + 3 * e;
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) {
                double v = 0;
                for (int k = 0; k < 3; ++k) v += Jx[3 * k + i] * Jx[3 * k + j];
                D[3 * i + j] += v;
            }
            double gv = 0;
            for (int k = 0; k < 3; ++k) gv += Jx[3 * k + i] * re[k];
            G[i] += gv;
        }
    }
    for (int i = 0; i < 9; ++i) diag[9 * b + i] = D[i];
    for (int i = 0; i < 3; ++i) g[3 * b + i] = G[i];
}

// cost = 0.5 * sum ecost: lane l folds entries l, l + 64, ... in index order, then a fixed butterfly -> reproducible
__global__ __launch_bounds__(64) void k_pg_cost(int n_edges, const double* __restrict__ ecost, double* __restrict__ cost) {
    double s = 0.0;
    for (int e = threadIdx.x; e < n_edges; e += 64) s += ecost[e];
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if (threadIdx.x == 0) cost[0] = 0.5 * s;
}


// ---- the damped Gauss-Newton step on the device: (J^T J + diag(damp)) step = -g ---------------------------------------
// Block-Jacobi preconditioned conjugate gradients in ONE workgroup (a pose graph is a few hundred to a few thousand 3-vectors:
// the reference hands it to Ceres' SPARSE_NORMAL_CHOLESKY, pose_graph_2d.cc:187-200).  The matrix is never assembled: pose b's
// row is its damped diagonal block plus, over its incident constraints (the CSR list k_pg_poses walks), Ja^T Jb or its transpose
// times the other pose's vector -- a gather, no atomics.  Every sum is a fixed-order reduction: reproducible.
__device__ __forceinline__ double block_sum(double v, double* red) {
    const int t = threadIdx.x;
    red[t] = v;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) { if (t < s) red[t] += red[t + s]; __syncthreads(); }
    const double r = red[0];
    __syncthreads();
    return r;
}
__device__ __forceinline__ bool inv3_dev(const double A[9], double I[9]) {
    const double d = A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
    if (d == 0 || !isfinite(d)) return false;
    const double id = 1.0 / d;
    I[0] = (A[4] * A[8] - A[5] * A[7]) * id; I[1] = (A[2] * A[7] - A[1] * A[8]) * id; I[2] = (A[1] * A[5] - A[2] * A[4]) * id;
    I[3] = (A[5] * A[6] - A[3] * A[8]) * id; I[4] = (A[0] * A[8] - A[2] * A[6]) * id; I[5] = (A[2] * A[3] - A[0] * A[5]) * id;
    I[6] = (A[3] * A[7] - A[4] * A[6]) * id; I[7] = (A[1] * A[6] - A[0] * A[7]) * id; I[8] = (A[0] * A[4] - A[1] * A[3]) * id;
    return true;
}
// work: [x | r | z | p | Ap], dim doubles each; minv: [n_free][9]; status: [0] 0 ok / 1 not positive definite, [1] iterations
__global__ __launch_bounds__(256) void k_pg_pcg(int n_free, const int* __restrict__ first, const int* __restrict__ inc, const DevEdge* __restrict__ edges,
                                                const double* __restrict__ diag, const double* __restrict__ off, const double* __restrict__ g,
                                                const double* __restrict__ damp, double* __restrict__ work, double* __restrict__ minv,
                                                int max_iter, int* __restrict__ status) {
#pragma clang fp contract(off)
    __shared__ double red[256];
    __shared__ int s_bad;
    const int tid = threadIdx.x, dim = 3 * n_free;
    double* x = work; double* r = x + dim; double* z = r + dim; double* p = z + dim; double* Ap = p + dim;
    if (tid == 0) s_bad = 0;
    __syncthreads();
    for (int b = tid; b < n_free; b += 256) {
        double A[9], I[9];
        for (int i = 0; i < 9; ++i) A[i] = diag[9 * b + i];
        for (int i = 0; i < 3; ++i) A[4 * i] += damp[3 * b + i];
        if (!inv3_dev(A, I)) s_bad = 1;
        for (int i = 0; i < 9; ++i) minv[9 * b + i] = I[i];
    }
    double acc = 0;
    for (int i = tid; i < dim; i += 256) { const double v = -g[i]; r[i] = v; x[i] = 0.0; acc += v * v; }
    __syncthreads();
    if (s_bad) { if (tid == 0) { status[0] = 1; status[1] = 0; } return; }
    const double bnorm = block_sum(acc, red);
    if (bnorm == 0) { if (tid == 0) { status[0] = 0; status[1] = 0; } return; }
    auto precond = [&]() {                                  // z = Minv r; returns this thread's share of r . z
        double a = 0;
        for (int b = tid; b < n_free; b += 256)
            for (int i = 0; i < 3; ++i) {
                const double v = minv[9 * b + 3 * i] * r[3 * b] + minv[9 * b + 3 * i + 1] * r[3 * b + 1] + minv[9 * b + 3 * i + 2] * r[3 * b + 2];
                z[3 * b + i] = v; a += r[3 * b + i] * v;
            }
        return a;
    };
    double rz = block_sum(precond(), red);
    for (int i = tid; i < dim; i += 256) p[i] = z[i];
    __syncthreads();
    int it = 0, code = 0, done = 0;
    for (; it < max_iter; ++it) {
        // Ap = (H + diag(damp)) p
        double a = 0;
        for (int b = tid; b < n_free; b += 256) {
            double y[3];
            for (int i = 0; i < 3; ++i) {
                double sacc = damp[3 * b + i] * p[3 * b + i];
                for (int j = 0; j < 3; ++j) sacc += diag[9 * b + 3 * i + j] * p[3 * b + j];
                y[i] = sacc;
            }
            for (int t = first[b]; t < first[b + 1]; ++t) {
                const int e = inc[t] >> 1, side = inc[t] & 1;
                const int co = side ? edges[e].ca : edges[e].cb;          // the OTHER pose's first column (-1: constant)
                if (co < 0) continue;
                const double* B = off + 9 * e;                             // Ja^T Jb
                if (side == 0) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) y[i] += B[3 * i + j] * p[co + j]; }
                else           { for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) y[j] += B[3 * i + j] * p[co + i]; }
            }
            for (int i = 0; i < 3; ++i) { Ap[3 * b + i] = y[i]; a += p[3 * b + i] * y[i]; }
        }
        const double pAp = block_sum(a, red);
        if (!(pAp > 0)) { code = 1; break; }
        const double alpha = rz / pAp;
        double rn = 0;
        for (int i = tid; i < dim; i += 256) { x[i] += alpha * p[i]; const double v = r[i] - alpha * Ap[i]; r[i] = v; rn += v * v; }
        __syncthreads();
        rn = block_sum(rn, red);
        if (rn <= 1e-24 * bnorm) { ++it; done = 1; break; }
        const double rz2 = block_sum(precond(), red);
        const double beta = rz2 / rz; rz = rz2;
        for (int i = tid; i < dim; i += 256) p[i] = z[i] + beta * p[i];
        __syncthreads();
    }
    if (tid == 0) { status[0] = code ? code : (done ? 0 : 2); status[1] = it; }
}

}  // namespace

struct DevProblem {
    int device = 0, n_poses = 0, dim = 0, n_edges = 0, n_free = 0;
    DevEdge* d_edges = nullptr; int* d_first = nullptr; int* d_inc = nullptr;
    double *d_x = nullptr, *d_r = nullptr, *d_J = nullptr, *d_off = nullptr, *d_ecost = nullptr, *d_diag = nullptr, *d_g = nullptr, *d_cost = nullptr;
    double *d_damp = nullptr, *d_work = nullptr, *d_minv = nullptr; int* d_status = nullptr;   // k_pg_pcg: damping, [x r z p Ap], block inverses, [code, iterations]
    double* h_pin = nullptr;            // pinned staging: x in, then [cost | r | diag | off | g] out
    size_t pin_doubles = 0;
    hipStream_t stream = nullptr;
    std::string err;
};

#define PG_TRY(p, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { (p)->err = std::string(#expr) + ": " + hipGetErrorString(e_); return (int)e_; } } while (0)

static int dev_init(DevProblem* p, const std::vector<DevEdge>& edges, const std::vector<int>& col) {
    PG_TRY(p, hipSetDevice(p->device));
    PG_TRY(p, hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
    const size_t E = edges.size(), nf = (size_t)p->n_free;
    // CSR incidence of the free poses, entries in constraint order
    std::vector<int> first(nf + 1, 0), inc;
    for (const DevEdge& e : edges) { if (e.ca >= 0) first[e.ca / 3 + 1] += 1; if (e.cb >= 0) first[e.cb / 3 + 1] += 1; }
    for (size_t b = 0; b < nf; ++b) first[b + 1] += first[b];
    inc.resize(first[nf]);
    std::vector<int> fill(first.begin(), first.end() - 1);
    for (size_t e = 0; e < E; ++e) {
        if (edges[e].ca >= 0) inc[fill[edges[e].ca / 3]++] = 2 * (int)e;
        if (edges[e].cb >= 0) inc[fill[edges[e].cb / 3]++] = 2 * (int)e + 1;
    }
    (void)col;
    PG_TRY(p, hipMalloc(&p->d_edges, sizeof(DevEdge) * std::max<size_t>(E, 1)));
    PG_TRY(p, hipMalloc(&p->d_first, sizeof(int) * (nf + 1)));
    PG_TRY(p, hipMalloc(&p->d_inc, sizeof(int) * std::max<size_t>(inc.size(), 1)));
    PG_TRY(p, hipMemcpy(p->d_edges, edges.data(), sizeof(DevEdge) * E, hipMemcpyHostToDevice));
    PG_TRY(p, hipMemcpy(p->d_first, first.data(), sizeof(int) * (nf + 1), hipMemcpyHostToDevice));
    PG_TRY(p, hipMemcpy(p->d_inc, inc.data(), sizeof(int) * inc.size(), hipMemcpyHostToDevice));
    PG_TRY(p, hipMalloc(&p->d_x, sizeof(double) * 3 * std::max(p->n_poses, 1)));
    PG_TRY(p, hipMalloc(&p->d_r, sizeof(double) * 3 * std::max<size_t>(E, 1)));
    PG_TRY(p, hipMalloc(&p->d_J, sizeof(double) * 18 * std::max<size_t>(E, 1)));
    PG_TRY(p, hipMalloc(&p->d_off, sizeof(double) * 9 * std::max<size_t>(E, 1)));
    PG_TRY(p, hipMalloc(&p->d_ecost, sizeof(double) * std::max<size_t>(E, 1)));
    PG_TRY(p, hipMalloc(&p->d_diag, sizeof(double) * 9 * std::max<size_t>(nf, 1)));
    PG_TRY(p, hipMalloc(&p->d_g, sizeof(double) * 3 * std::max<size_t>(nf, 1)));
    PG_TRY(p, hipMalloc(&p->d_cost, sizeof(double)));
    PG_TRY(p, hipMalloc(&p->d_damp, sizeof(double) * 3 * std::max<size_t>(nf, 1)));
    PG_TRY(p, hipMalloc(&p->d_work, sizeof(double) * 15 * std::max<size_t>(nf, 1)));
    PG_TRY(p, hipMalloc(&p->d_minv, sizeof(double) * 9 * std::max<size_t>(nf, 1)));
    PG_TRY(p, hipMalloc(&p->d_status, sizeof(int) * 2));
    p->pin_doubles = 3 * (size_t)p->n_poses + 1 + 3 * E + 9 * nf + 9 * E + 3 * nf;
    PG_TRY(p, hipHostMalloc(&p->h_pin, sizeof(double) * p->pin_doubles));
    return 0;
}

DevProblem* dev_create(int device, int n_poses, int dim, const std::vector<DevEdge>& edges, const std::vector<int>& col, std::string& err) {
    DevProblem* p = new DevProblem();
    p->device = device; p->n_poses = n_poses; p->dim = dim; p->n_edges = (int)edges.size(); p->n_free = dim / 3;
    if (dev_init(p, edges, col)) { err = p->err; dev_destroy(p); return nullptr; }
    return p;
}

void dev_destroy(DevProblem* p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    (void)hipFree(p->d_edges); (void)hipFree(p->d_first); (void)hipFree(p->d_inc); (void)hipFree(p->d_x); (void)hipFree(p->d_r); (void)hipFree(p->d_J);
    (void)hipFree(p->d_off); (void)hipFree(p->d_ecost); (void)hipFree(p->d_diag); (void)hipFree(p->d_g); (void)hipFree(p->d_cost);
    (void)hipFree(p->d_damp); (void)hipFree(p->d_work); (void)hipFree(p->d_minv); (void)hipFree(p->d_status);
    if (p->h_pin) (void)hipHostFree(p->h_pin);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
}

const char* dev_error(const DevProblem* p) { return p ? p->err.c_str() : ""; }
int dev_device(const DevProblem* p) { return p ? p->device : -1; }

static int enqueue(DevProblem* p, const double* x, bool normal) {
    PG_TRY(p, hipSetDevice(p->device));
    // the ONE pinned staging buffer may still be the source of an earlier call's asynchronous upload (dev_cost_async returns
    // without synchronising): wait for that stream work before the buffer is overwritten
    PG_TRY(p, hipStreamSynchronize(p->stream));
    std::copy(x, x + 3 * (size_t)p->n_poses, p->h_pin);
    PG_TRY(p, hipMemcpyAsync(p->d_x, p->h_pin, sizeof(double) * 3 * p->n_poses, hipMemcpyHostToDevice, p->stream));
    if (p->n_edges > 0)
        hipLaunchKernelGGL(k_pg_edges, dim3((p->n_edges + 63) / 64), dim3(64), 0, p->stream, p->n_edges, p->d_edges, p->d_x, p->d_r, p->d_J, p->d_off, p->d_ecost, normal ? 1 : 0);
    if (normal && p->n_free > 0)
        hipLaunchKernelGGL(k_pg_poses, dim3((p->n_free + 63) / 64), dim3(64), 0, p->stream, p->n_free, p->d_first, p->d_inc, p->d_r, p->d_J, p->d_diag, p->d_g);
    hipLaunchKernelGGL(k_pg_cost, dim3(1), dim3(64), 0, p->stream, p->n_edges, p->d_ecost, p->d_cost);
    PG_TRY(p, hipGetLastError());
    return 0;
}

int dev_linearize(DevProblem* p, const double* x, double* cost, double* r, double* diag, double* off, double* g) {
    const bool normal = diag || off || g;
    int rc = enqueue(p, x, normal);
    if (rc) return rc;
    const size_t E = (size_t)p->n_edges, nf = (size_t)p->n_free;
    double* o = p->h_pin + 3 * (size_t)p->n_poses;
    double* h_cost = o; double* h_r = o + 1; double* h_diag = h_r + 3 * E; double* h_off = h_diag + 9 * nf; double* h_g = h_off + 9 * E;
    PG_TRY(p, hipMemcpyAsync(h_cost, p->d_cost, sizeof(double), hipMemcpyDeviceToHost, p->stream));
    if (r && E) PG_TRY(p, hipMemcpyAsync(h_r, p->d_r, sizeof(double) * 3 * E, hipMemcpyDeviceToHost, p->stream));
    if (diag && nf) PG_TRY(p, hipMemcpyAsync(h_diag, p->d_diag, sizeof(double) * 9 * nf, hipMemcpyDeviceToHost, p->stream));
    if (off && E) PG_TRY(p, hipMemcpyAsync(h_off, p->d_off, sizeof(double) * 9 * E, hipMemcpyDeviceToHost, p->stream));
    if (g && nf) PG_TRY(p, hipMemcpyAsync(h_g, p->d_g, sizeof(double) * 3 * nf, hipMemcpyDeviceToHost, p->stream));
    PG_TRY(p, hipStreamSynchronize(p->stream));
    if (cost) *cost = *h_cost;
    if (r) std::copy(h_r, h_r + 3 * E, r);
    if (diag) std::copy(h_diag, h_diag + 9 * nf, diag);
    if (off) std::copy(h_off, h_off + 9 * E, off);
    if (g) std::copy(h_g, h_g + 3 * nf, g);
    return 0;
}

// (J^T J + diag(damp)) step = -g for the point of the latest FULL linearisation (its blocks are still on the device; cost-only
// evaluations in between leave them alone).  damp, step: host, dim doubles.  Returns 0 (solved), 2 (solved INEXACTLY: the PCG
// iteration limit came before its tolerance; the step is the best iterate), 1 (not positive definite: the caller shrinks the
// trust region, as after a failed host solve) or a negative hipError_t.
// (dev_solve's contract is "negative on a HIP error": PG_TRY returns the positive hipError_t, which optimize() would read as
// "not positive definite" and answer by shrinking the trust region until NIK_PG_FAILURE -- ADVICE r4)
#define PG_TRYN(p, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { (p)->err = std::string(#expr) + ": " + hipGetErrorString(e_); return -(int)e_; } } while (0)
int dev_solve(DevProblem* p, const double* damp, double* step, int* iterations) {
    if (p->n_free == 0) return 0;
    PG_TRYN(p, hipSetDevice(p->device));
    PG_TRYN(p, hipStreamSynchronize(p->stream));               // (h_pin may still feed an earlier upload)
    const size_t dim = 3 * (size_t)p->n_free;
    std::copy(damp, damp + dim, p->h_pin);
    PG_TRYN(p, hipMemcpyAsync(p->d_damp, p->h_pin, sizeof(double) * dim, hipMemcpyHostToDevice, p->stream));
    hipLaunchKernelGGL(k_pg_pcg, dim3(1), dim3(256), 0, p->stream, p->n_free, p->d_first, p->d_inc, p->d_edges, p->d_diag, p->d_off, p->d_g, p->d_damp,
                       p->d_work, p->d_minv, (int)(20 * dim), p->d_status);
    PG_TRYN(p, hipGetLastError());
    int st[2] = { 0, 0 };
    PG_TRYN(p, hipMemcpyAsync(p->h_pin, p->d_work, sizeof(double) * dim, hipMemcpyDeviceToHost, p->stream));
    PG_TRYN(p, hipMemcpyAsync(st, p->d_status, sizeof(st), hipMemcpyDeviceToHost, p->stream));
    PG_TRYN(p, hipStreamSynchronize(p->stream));
    std::copy(p->h_pin, p->h_pin + dim, step);
    if (iterations) *iterations = st[1];
    // st[0]: 0 converged, 1 not positive definite, 2 iteration limit reached without convergence (the step is still the best
    // iterate: the trust-region test of the caller accepts or rejects it on its merits)
    return st[0] == 1 ? 1 : (st[0] == 2 ? 2 : 0);
}

int dev_cost_async(DevProblem* p, const double* x, double** d_cost, void** stream) {
    int rc = enqueue(p, x, false);
    if (rc) return rc;
    if (d_cost) *d_cost = p->d_cost;
    if (stream) *stream = (void*)p->stream;
    return 0;
}

}  // namespace kcc_pg
