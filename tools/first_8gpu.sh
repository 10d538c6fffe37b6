#!/bin/bash
# usage: bash tools/first_8gpu.sh            -- on an 8-GPU node: bench.py --gpus {1,2,4,8} for pairs, hd, loop4096 with the
#                                               assertions of tools/first_8gpu.py (rccl_ranks == N, no fallback, rank balance)
#        bash tools/first_8gpu.sh --selftest -- on a 1-GPU box: N = 1 and the 2-rank one-device gloo hook (exercises the script)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && exec python tools/first_8gpu.py "$@"
