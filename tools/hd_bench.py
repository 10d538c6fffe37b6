"""BASELINE config 4 on one GPU: 1280x720 RGB frames -> gray (integer luma) -> ComputeIntermedium + ComputePose,
batch of pairs resident in HBM.  (The 8-GPU sharding of this config is bench.py's --gpus path: pairs split across
ranks, one 4-double all-reduce per step.)  Prints one JSON line; an extra measurement, not the headline."""
import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import numpy as np, torch, synth
from kcc_helpers import nik
N = nik()
H, W, B = 720, 1280, int(sys.argv[1]) if len(sys.argv) > 1 else 32
cf = N.CorrelationFlow(N.default_config(), H, W, max_batch=B, max_frames=2 * B)
U = 8
keys, curs, _ = synth.make_batch(U, H, W, seed0=11, max_shift=60, max_theta=8.0)
rep = (B + U - 1) // U
rgb_k = np.repeat(np.tile(keys, (rep, 1, 1))[:B, :, :, None], 3, axis=3).copy()
rgb_c = np.repeat(np.tile(curs, (rep, 1, 1))[:B, :, :, None], 3, axis=3).copy()
dk = torch.from_numpy(rgb_k).cuda(); dc = torch.from_numpy(rgb_c).cuda()
gk = torch.empty((B, H, W), dtype=torch.uint8, device="cuda"); gc = torch.empty_like(gk)
torch.cuda.synchronize()
cf.rgb_to_gray_dev(dk.data_ptr(), B, gk.data_ptr())
cf.intermedium_batch_dev(gk.data_ptr(), B, list(range(B))); cf.synchronize()
ring = [(N.NikPoseResult * B)() for _ in range(3)]
def step(k):
    cf.rgb_to_gray_dev(dc.data_ptr(), B, gc.data_ptr())          # colour conversion is part of the step (synchronous helper)
    return cf.track_batch_dev(gc.data_ptr(), list(range(B)), list(range(B, 2 * B)), True, sync=False, res=ring[k % 3])
for k in range(3): step(k)
cf.synchronize()
steps = 10
t0 = time.perf_counter()
for k in range(steps): step(k)
cf.synchronize(); torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
MB = 83.08e6
print(json.dumps({"config": "configs[3] on 1 GPU: 1280x720 RGB", "pairs_per_step": B, "ms_per_step": round(1e3 * dt, 3), "pairs_per_s": round(B / dt, 1),
                  "frac_of_8TBps_at_83.08MB": round(B / dt * MB / 8e12, 4), "pose0": ring[(steps - 1) % 3][0].as_dict()["pose"]}))
if os.environ.get("NIK_HD_PROFILE"):
    cf.set_streams(1); cf.profile_enable(True)
    for k in range(4): step(k)
    cf.synchronize()
    st = sorted(cf.profile_read(), key=lambda r: -r["ms"])
    for r in st:
        if r["launches"]:
            print("%-28s %.3f ms  %.0f GB/s" % (r["name"], r["ms"] / r["launches"], r["bytes"] / r["launches"] / (r["ms"] / r["launches"]) / 1e6))
