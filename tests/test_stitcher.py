"""MapStitcher on the device (reference src/map_stitcher.cc) against its literal CPU restatement (tests/ref_stitcher.py):
occupancy cells bit-exact after inserting rotated / translated key frames (several cells, negative cell indices, cells
seen once and many times) and after RecomputeOccupancy with updated poses."""
import numpy as np
import pytest

import synth
from kcc_helpers import SMALL, nik
from ref_stitcher import RefStitcher, cell_position


def test_cell_position_matches_cxx_division():
    cell, pos = cell_position(np.array([-130, -65, -64, -1, 0, 63, 64, 130]), 64)
    assert cell.tolist() == [-3, -2, -1, -1, 0, 0, 1, 2] and pos.tolist() == [62, 63, 0, 63, 0, 63, 0, 2]


@pytest.mark.gpu
def test_stitcher_matches_reference_logic():
    import torch
    N = nik()
    H, W = SMALL["H"], SMALL["W"]
    cf = N.CorrelationFlow(N.default_config(rotation_divisor=SMALL["PD"], rotation_channel=SMALL["PC"]), H, W, max_batch=2, max_frames=4)
    size = 64
    st = N.Stitcher(cf, size); ref = RefStitcher(H, W, size)
    rng = np.random.default_rng(4)
    cv = synth.canvas(8, H, W)
    poses = {}
    for fid in range(9):
        img = synth.window(cv, H, W, int(rng.integers(-5, 6)), int(rng.integers(-5, 6)))
        pose = (float(rng.uniform(-90, 70)), float(rng.uniform(-60, 80)), float(rng.uniform(-3.1, 3.1)))
        if fid == 0:
            pose = (0.0, 0.0, 0.0)
        d = torch.from_numpy(img).cuda()
        st.insert_dev(3 * fid, d.data_ptr(), pose); ref.insert(3 * fid, img, pose)
        poses[3 * fid] = pose

    def compare():
        assert sorted(st.cells()) == sorted(ref.cells)
        for key in ref.cells:
            d, w = st.read_cell(*key)
            assert np.array_equal(w, ref.cells[key][1]), key
            assert np.array_equal(d, ref.cells[key][0]), key
    compare()
    assert len(ref.cells) >= 6 and min(k[0] for k in ref.cells) < 0          # several cells, negative indices exercised
    assert max(int(w.max()) for _, w in ref.cells.values()) >= 4             # cells blended many times
    # pose-graph update: some poses move, an unknown id is ignored, then everything is replayed
    new = {fid: (p[0] + 7.3, p[1] - 4.1, p[2] + 0.2) for fid, p in poses.items() if fid % 2 == 0}
    new[999] = (0.0, 0.0, 0.0)
    st.recompute(list(new), [new[k] for k in new]); ref.recompute(new)
    compare()
    with pytest.raises(N.NikError):
        st.insert_dev(0, torch.zeros((H, W), dtype=torch.uint8).cuda().data_ptr(), (0, 0, 0))      # duplicate frame id
    st.close()
