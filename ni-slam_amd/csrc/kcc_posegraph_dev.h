// kcc_posegraph_dev.h -- device-side linearisation of the 2-D pose graph (kcc_posegraph_dev.hip), used by the
// Levenberg-Marquardt of kcc_posegraph.cpp and by nik_group_pose_graph_cost (kcc_group.cpp).  Plain types only.
#pragma once
#include <string>
#include <vector>

namespace kcc_pg {

struct DevEdge { int a, b, ca, cb; double m[3]; double L[9]; };      // pose indices, reduced columns (-1: constant / unused), measurement, sqrt information

struct DevProblem;                                                   // device-resident edges, incidence lists and work buffers

// n_poses poses, `dim` free parameters (3 per free pose); col[i] = first reduced column of pose i or -1
DevProblem* dev_create(int device, int n_poses, int dim, const std::vector<DevEdge>& edges, const std::vector<int>& col, std::string& err);
void dev_destroy(DevProblem* p);
// x: host poses [n_poses][3].  Outputs (host, each may be null): cost = 0.5 sum r^2; r [3 E]; diag [dim/3][9] (J^T J diagonal
// blocks); off [E][9] (Ja^T Jb); g [dim] (J^T r).  Deterministic (fixed summation orders).  Returns 0 or a hipError_t.
int dev_linearize(DevProblem* p, const double* x, double* cost, double* r, double* diag, double* off, double* g);
// the damped step (J^T J + diag(damp)) step = -g at the point of the latest full linearisation, by block-Jacobi PCG in one
// workgroup (k_pg_pcg).  damp, step: host [dim].  0 ok, 1 not positive definite, < 0 hipError_t.
int dev_solve(DevProblem* p, const double* damp, double* step, int* iterations);
// cost only, left ON THE DEVICE: *d_cost points at one double valid after work on *stream (a hipStream_t) has finished
int dev_cost_async(DevProblem* p, const double* x, double** d_cost, void** stream);
const char* dev_error(const DevProblem* p);
int dev_device(const DevProblem* p);                                 // the GPU the problem lives on (-1: null)

}  // namespace kcc_pg
