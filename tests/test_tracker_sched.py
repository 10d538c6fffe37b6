"""The tracker's batching on the CPU: kcc_tracker.cpp compiled against a stub of the C ABI underneath it (tests/cpp/tracker_sched_test.cpp)
-- asynchronous batches whose results arrive only when waited for, a deterministic ComputePose -- and driven over sequences with
regular, periodic and irregular keyframe gaps, PSR-gated insertions and lost frames.  1176 combinations of window size, look-ahead
depth, batch room, prefetching (none / one / two windows), ragged windows and the host-frame entry point must all give, bit for
bit, the outputs of one nik_tracker_push_u8 per frame: MapBuilder::AddNewInput's loop (src/map_builder.cc:30-70).  36 more runs
inject a failing batch: the push reports it, nothing stays in flight, and pushing again from the first undecided frame carries on
exactly."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_lookahead_batches_equal_frame_by_frame(tmp_path):
    exe = str(tmp_path / "tracker_sched")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-ffp-contract=off", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", exe,
                    os.path.join(ROOT, "tests", "cpp", "tracker_sched_test.cpp"), os.path.join(ROOT, "ni-slam_amd", "csrc", "kcc_tracker.cpp")],
                   check=True, timeout=600)
    r = subprocess.run([exe, "500"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith("OK: 1176 configurations, 0 differ"), last
