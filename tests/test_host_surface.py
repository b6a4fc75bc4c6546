"""CPU tier: host-side pieces of the drop-in surface that need no kernel (I/O formats, zone pick, alignment,
Sim3 index pick, config marshalling) against the reference-generated goldens."""
import numpy as np
import pytest

from gps_optimize_slam_amd import _lib
from gps_optimize_slam_amd import ekfgpsslam as E


def test_config_matches_oracle_defaults():
    from oracle import oracle as orc
    for sec in ("ekf", "sim3_ransac", "time_alignment", "rts_decision"):
        assert E.CONFIG[sec] == orc.DEFAULT_CONFIG[sec]


def test_tum_io_roundtrip(tmp_path, golden):
    g = golden("kat_bundled.npz")
    p = tmp_path / "out_corrected_utm.txt"
    E.save_tum_utm(str(p), g["ts"], g["kat3_pos"], g["kat3_quat"])
    lines = p.read_text().splitlines()
    assert lines[0] == "timestamp x y z qx qy qz qw (UTM)"          # header without '#', SURVEY Q14
    assert len(lines) == 272
    first = lines[1].split()
    assert first[0] == "0.000000" and len(first[1].split(".")[1]) == 6 and len(first[4].split(".")[1]) == 8
    # a file with a header line cannot be np.loadtxt'ed (the reference's own output is not re-loadable): ValueError
    with pytest.raises(ValueError):
        E.load_slam_trajectory(str(p))
    q = tmp_path / "traj.txt"
    np.savetxt(q, np.column_stack((g["ts"], g["pos"], g["quat"])))
    d = E.load_slam_trajectory(str(q))
    np.testing.assert_array_equal(d["timestamps"], g["ts"])           # timestamps bit-exact through the text surface
    np.testing.assert_array_equal(d["positions"], g["pos"])
    with pytest.raises(ValueError):
        E.load_slam_trajectory(str(tmp_path / "missing.txt"))
    bad = tmp_path / "bad.txt"
    np.savetxt(bad, np.zeros((3, 7)))
    with pytest.raises(ValueError):
        E.load_slam_trajectory(str(bad))


def test_zone_pick():
    assert E.auto_utm_projection(np.array([8.39]), np.array([49.0])) == (32, "")
    assert E.auto_utm_projection(np.array([49.03]), np.array([8.39])) == (39, "")
    assert E.auto_utm_projection(np.array([-180.0]), np.array([-1.0])) == (1, " +south")
    assert E.auto_utm_projection(np.array([180.0]), np.array([1.0]))[0] == 61
    with pytest.raises(ValueError):
        E.auto_utm_projection(np.array([]), np.array([]))


def test_estimate_time_offset_is_zero(golden):
    g = golden("kat_bundled.npz")
    assert E.estimate_time_offset(g["ts"], g["ts"] + 0.37, 500) == 0.0 == float(g["kat5_offset"])


@pytest.mark.parametrize("tag", ["kitti04gps", "combined"])
def test_c1_host_stages(golden, tag):
    g, k = golden(f"c1_{tag}.npz"), golden("kat_bundled.npz")
    # (the RANSAC pre-filter runs on the GPU now: test_gps_ransac_filter_vs_reference_goldens / test_c1_dropin_* in the -m gpu tier)
    slam = {"timestamps": k["ts"], "positions": k["pos"], "quaternions": k["quat"]}
    np.testing.assert_array_equal(E.pick_sim3_indices(slam, g["valid"]), g["sim3_idx"])      # alignment itself: -m gpu tier


def test_ekf_config_marshalling():
    c = _lib.EkfConfig.from_config(E.CONFIG)
    assert list(c.process_noise_diag) == [0.1, 0.1, 0.7, 0.01, 0.01, 0.01, 0.01]
    assert c.sharp_turn_yaw_rate_threshold_deg_per_sec == 45.0 and c.default_ekf_transition_steps_on_sharp_turn == 0
    bad = {"ekf": dict(E.CONFIG["ekf"], initial_cov_diag=[1, 2, 3]), "rts_decision": E.CONFIG["rts_decision"]}
    with pytest.raises(ValueError):
        _lib.EkfConfig.from_config(bad)
