"""Coarse-to-fine registration over an image pyramid: BASELINE config 3 ("640x480 stereo + 4-level correlation pyramid
with radius-4 lookup, batch 32").  The reference has no counterpart (SURVEY 8d marks it an extension): parity is against
this repository's own CPU restatement -- the oracle run per level with the same windows -- and the level-0 result must
equal the plain single-resolution KCC answer.  "Stereo" = two independent mono streams: just more pairs in the batch."""
import numpy as np
import pytest

import synth
from kcc_helpers import FULL, ang_diff, check_pose_parity, nik
from oracle import kcc_oracle as O

LEVEL_POLAR = [(720, 480), (480, 320), (240, 160), (120, 80)]


def predict(idx, n_from, n_to):
    off = (idx - n_from // 2) * n_to / n_from
    return (n_to // 2 + int(np.floor(abs(off) + 0.5)) * (1 if off >= 0 else -1)) % n_to      # lround: half away from zero


def oracle_pyramid(key, cur, levels, radius):
    """the definition, restated on the CPU oracle: returns per-level (pose, info, dbg), level 0 first"""
    ks, cs = [key], [cur]
    for _ in range(1, levels):
        ks.append(O.downsample_u8(ks[-1])); cs.append(O.downsample_u8(cs[-1]))
    out = [None] * levels
    for l in range(levels - 1, -1, -1):
        H, W = ks[l].shape
        PD, PC = LEVEL_POLAR[l]
        ora = O.Oracle(O.default_config(rotation_divisor=PD, rotation_channel=PC), H, W)
        if l < levels - 1:
            up = out[l + 1][2]
            Hu, Wu = ks[l + 1].shape
            PDu, PCu = LEVEL_POLAR[l + 1]
            ora.set_window(predict(up["rot_row"], PDu, PD), predict(up["rot_col"], PCu, PC),
                           predict(up["trans_row"][0], Hu, H), predict(up["trans_col"][0], Wu, W), radius)
        kf, kp = ora.intermedium(ora.normalize_u8(ks[l]))
        ci = ora.normalize_u8(cs[l])
        cf, cp = ora.intermedium(ci)
        out[l] = ora.compute_pose(kf, ci, kp, cp, True)
    return out


def test_oracle_building_blocks():
    img = synth.window(synth.canvas(3, 60, 80), 60, 80, 0, 0)
    d = O.downsample_u8(img)
    want = (img[0::2, 0::2].astype(int) + img[0::2, 1::2] + img[1::2, 0::2] + img[1::2, 1::2] + 2) >> 2
    assert np.array_equal(d, want)
    assert [predict(i, 60, 120) for i in (0, 29, 30, 31, 59)] == [0, 58, 60, 62, 118]
    assert predict(125, 480, 720) == 360 - 173 and predict(355, 480, 720) == 360 + 173      # 1.5x: halves round away from zero
    # a window around the true peak leaves the answer unchanged; a window elsewhere moves it inside that window
    H, W, PD, PC = 60, 80, 120, 80
    ora = O.Oracle(O.default_config(rotation_divisor=PD, rotation_channel=PC), H, W)
    key, cur = synth.make_pair(9, H, W, 4, -3, 0.0)
    kf, kp = ora.intermedium(ora.normalize_u8(key)); ci = ora.normalize_u8(cur); cf, cp = ora.intermedium(ci)
    pose, info, dbg = ora.compute_pose(kf, ci, kp, cp, True)
    ora.set_window(dbg["rot_row"], dbg["rot_col"], dbg["trans_row"][0], dbg["trans_col"][0], 4)
    pose2, info2, dbg2 = ora.compute_pose(kf, ci, kp, cp, True)
    assert list(pose2) == list(pose) and dbg2["trans_row"] == dbg["trans_row"]
    ora.set_window(dbg["rot_row"], dbg["rot_col"], (dbg["trans_row"][0] + 20) % H, dbg["trans_col"][0], 2)
    _, _, dbg3 = ora.compute_pose(kf, ci, kp, cp, True)
    assert min(abs(dbg3["trans_row"][0] - (dbg["trans_row"][0] + 20) % H), H - abs(dbg3["trans_row"][0] - (dbg["trans_row"][0] + 20) % H)) <= 2
    ora.set_window(0, 0, 0, 0, -1)
    assert list(ora.compute_pose(kf, ci, kp, cp, True)[0]) == list(pose)


@pytest.mark.gpu
def test_pyramid_matches_oracle_and_plain_kcc():
    import torch
    N = nik()
    H, W = FULL["H"], FULL["W"]
    levels, radius, n = 4, 4, 4
    pyr = N.Pyramid(N.default_config(), H, W, levels=levels, max_batch=n)
    assert pyr.dims == [[480, 640, 720, 480], [240, 320, 480, 320], [120, 160, 240, 160], [60, 80, 120, 80]]
    motions = [(16, -24, 0.0), (-33, 9, 4.0), (8, 40, -7.5), (-21, -14, 2.5)]        # "stereo": pairs 0/1 and 2/3 are two streams
    pairs = [synth.make_pair(300 + i, H, W, dy, dx, th) for i, (dy, dx, th) in enumerate(motions)]
    keys = np.stack([p[0] for p in pairs]); curs = np.stack([p[1] for p in pairs])
    dk = torch.from_numpy(keys).cuda(); dc = torch.from_numpy(curs).cuda(); torch.cuda.synchronize()
    got = pyr.track_dev(dk.data_ptr(), dc.data_ptr(), n, radius)
    # plain single-resolution KCC on the same pairs
    cf = N.CorrelationFlow(N.default_config(), H, W, max_batch=n, max_frames=2 * n)
    cf.intermedium_batch_dev(dk.data_ptr(), n, list(range(n)))
    plain = cf.track_batch_dev(dc.data_ptr(), list(range(n)), list(range(n, 2 * n)), True)
    for i in range(n):
        want = oracle_pyramid(keys[i], curs[i], levels, radius)
        for l in range(levels):
            pose, info, dbg = want[l]
            ok, _, msg = check_pose_parity(got[l][i], pose, info, dbg, LEVEL_POLAR[l][0], psr_rtol=5e-3)
            assert ok, (i, l, msg)
        # level 0 inside its window == the global answer of the plain path (and the known motion)
        p0 = plain[i].as_dict()
        assert got[0][i]["pose"][:2] == p0["pose"][:2]
        assert ang_diff(got[0][i]["pose"][2], p0["pose"][2]) < 1e-6
        assert got[0][i]["trans_row"][0] == p0["trans_row"][0] and got[0][i]["trans_col"][0] == p0["trans_col"][0]
    # the window API on its own: bad centres are rejected, a window around the plain peak reproduces the plain result
    with pytest.raises(N.NikError):
        cf.pose_batch_window([0], [n], [[720, 0, 0, 0]], 4)
    c0 = [[plain[i].rot_row, plain[i].rot_col, plain[i].trans_row[0], plain[i].trans_col[0]] for i in range(n)]
    again = cf.pose_batch_window(list(range(n)), list(range(n, 2 * n)), c0, 4)
    for i in range(n):
        assert again[i].as_dict()["pose"] == plain[i].as_dict()["pose"]
    pyr.close()


@pytest.mark.gpu
def test_pyramid_batches_in_flight_equal_single_batches():
    """nik_pyramid_track_dev_async: several DIFFERENT batches enqueued back to back (the levels of consecutive batches
    overlap, frame slots and box-filter buffers are recycled under stream order) give, after one synchronize, exactly
    what each batch gives alone."""
    import torch
    N = nik()
    H, W, levels, radius, n, nb = FULL["H"], FULL["W"], 4, 4, 6, 5
    pyr = N.Pyramid(N.default_config(), H, W, levels=levels, max_batch=n)
    batches = []
    for b in range(nb):
        keys, curs, _ = synth.make_unique_batch(n, H, W, seed0=700 + 31 * b, max_theta=6.0, max_shift=30)
        batches.append((torch.from_numpy(keys).cuda(), torch.from_numpy(curs).cuda()))
    torch.cuda.synchronize()
    alone = [pyr.track_dev(dk.data_ptr(), dc.data_ptr(), n, radius) for dk, dc in batches]
    raws = [pyr.track_dev_async(dk.data_ptr(), dc.data_ptr(), n, radius) for dk, dc in batches]
    pyr.synchronize()
    for b in range(nb):
        got = pyr.as_lists(raws[b], n)
        for l in range(levels):
            for i in range(n):
                assert got[l][i] == alone[b][l][i], (b, l, i)
    # distinct batches did give distinct answers (the comparison above is not vacuous)
    assert len({tuple(alone[b][0][0]["pose"]) for b in range(nb)}) > 1
    pyr.close()


@pytest.mark.gpu
def test_fused_downsample_equals_chained_box_filters():
    """nik_downsample_pyr_u8_stream: up to three levels per launch, two source buffers -- the integers of the chained 2x2 box
    filters (the oracle's downsample_u8), and a refusal (not a wrong answer) for unaligned pointers"""
    import torch
    N = nik()
    H, W, na, nb = 120, 160, 3, 2
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, (na, H, W), dtype=np.uint8); b = rng.integers(0, 256, (nb, H, W), dtype=np.uint8)
    cf = N.CorrelationFlow(N.default_config(rotation_divisor=240, rotation_channel=160), H, W, max_batch=4, max_frames=8)
    dev = torch.device("cuda:0")
    da, db = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    want = [np.concatenate([a, b])]
    for _ in range(3):
        want.append(np.stack([O.downsample_u8(f) for f in want[-1]]))
    for steps in (1, 2, 3):
        outs = [torch.zeros((na + nb, H >> d, W >> d), dtype=torch.uint8, device=dev) for d in range(1, steps + 1)]
        assert cf.downsample_pyr_u8(steps, na, da.data_ptr(), nb, db.data_ptr(), [o.data_ptr() for o in outs]) == 0
        torch.cuda.synchronize()
        for d in range(steps):
            assert np.array_equal(outs[d].cpu().numpy(), want[d + 1]), (steps, d)
    # one buffer only, and an unaligned source
    o1 = torch.zeros((na, H // 2, W // 2), dtype=torch.uint8, device=dev)
    assert cf.downsample_pyr_u8(1, na, da.data_ptr(), 0, 0, [o1.data_ptr()]) == 0
    torch.cuda.synchronize()
    assert np.array_equal(o1.cpu().numpy(), want[1][:na])
    big = torch.zeros(na * H * W + 8, dtype=torch.uint8, device=dev)
    assert cf.downsample_pyr_u8(3, na, big.data_ptr() + 1, 0, 0, [o.data_ptr() for o in outs]) == N.NIK_ERR_UNSUPPORTED_SIZE
    cf.close()
