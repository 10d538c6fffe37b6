// kcc_tune.h -- laboratory switches live in the TUNING library only.
//
// The release library (libnislam_kcc_hip.so) reads five environment variables, all of them things a caller may need:
// NIK_STREAMS, NIK_KZZ_CACHE, NIK_GENERIC, NIK_GRAPH, NIK_GROUP_* (INIT_TIMEOUT, FORCE_RCCL).  Every other $NIK_* switch --
// ablation flags, LDS padding for occupancy experiments, the ring-form B kernels, alternative fusion / ordering choices,
// tracker look-ahead depths -- is an experiment knob: tune_env() returns it only in a library built with -DKCC_ABLATE
// (ni-slam_amd/build.py build_tuning() -> libnislam_kcc_hip_tune.so, loaded through $NIK_LIB by tools/ and by the tests that
// exercise a non-default form), and is a compile-time nullptr in the release library.
#pragma once
#ifdef KCC_ABLATE
#include <cstdlib>
#endif
namespace kcc {
#ifdef KCC_ABLATE
inline const char* tune_env(const char* name) { return getenv(name); }
inline bool tuning_build() { return true; }
#else
inline const char* tune_env(const char*) { return nullptr; }     // (folds away: not even the switch's name reaches the release binary)
inline bool tuning_build() { return false; }
#endif
}  // namespace kcc
