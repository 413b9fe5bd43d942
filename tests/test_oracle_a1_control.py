"""CPU: the oracle's restatement of the A1 control-side wrappers (ETG action path, reward shaping; oracle/a1.py)
against vectors recorded from the unmodified reference (tests/golden/a1_control.npz, oracle/gen_golden_a1_control.py)."""
import os

import numpy as np
import pytest

from oracle import a1 as oa

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "a1_control.npz")


@pytest.fixture(scope="module")
def g():
    return np.load(GOLDEN)


def etg_path(g, name):
    etg, T, T2, H, sig, amp, pose_mode, gallop, space, _dt = g[name + "/config"]
    return oa.EtgActionPath(g[name + "/w"], g[name + "/b"], bool(etg), T, T2, int(H), sig, amp, bool(pose_mode), bool(gallop), int(space))


@pytest.mark.parametrize("idx", range(5))
def test_action_path_matches_reference(g, idx):
    name = str(g["b_cases"][idx])
    p = etg_path(g, name)
    obs = p.reset(0.0)
    if p.enabled:
        assert np.array_equal(obs, g[name + "/reset_etg_obs"][0])
        assert np.array_equal(p.last_etg_act, g[name + "/reset_etg_act"][0])
    for k in range(len(g[name + "/action"])):
        cmd, obs = p.step(g[name + "/action"][k], g[name + "/t"][k])
        assert np.array_equal(cmd, g[name + "/command"][k]), "%s command, step %d" % (name, k)
        if p.enabled:
            assert np.array_equal(obs, g[name + "/etg_obs"][k])
            assert np.array_equal(p.last_etg_act, g[name + "/etg_act"][k])
    if name == "etg_traj_unreachable":
        assert p.retries > 0, "the fixture must exercise the IK retry loop (ETG_model.py:126-129)"


def reward_case(g, name):
    reward_p, vel_d, d_yaw = g[name + "/config"]
    r = oa.RewardShaping(g[name + "/param"], reward_p, vel_d, g[name + "/segments"], vel_mode=str(g[name + "/vel_mode"]))
    return r, d_yaw


@pytest.mark.parametrize("idx", range(8))
def test_reward_shaping_matches_reference(g, idx):
    name = str(g["c_cases"][idx])
    r, d_yaw = reward_case(g, name)
    # reset(): last_* come from the RESET info; get_foot_world of it is recorded
    r.steps = 0
    r.last_basepose = g[name + "/reset_base"].copy()
    r.last_foot = g[name + "/reset_foot_world"].copy()
    r.last_base10 = np.tile(g[name + "/reset_base"], (10, 1))
    for k in range(len(g[name + "/reward"])):
        terms, reward, done = r.step(g[name + "/base"][k], g[name + "/pose"][k], g[name + "/rot_mat"][k],
                                     g[name + "/footposition"][k], g[name + "/real_contact"][k], g[name + "/energy"][k],
                                     g[name + "/bad"][k], d_yaw)
        assert np.array_equal(terms, g[name + "/terms"][k]), "%s terms, step %d" % (name, k)
        assert reward == g[name + "/reward"][k]
        assert done == bool(g[name + "/done"][k])
        assert np.array_equal(r.last_foot, g[name + "/foot_world"][k])


def test_reward_goldens_cover_every_termination_rule(g):
    done = {n: g[n + "/done"] for n in map(str, g["c_cases"])}
    assert done["reward_tumble"].any() and done["reward_feet_up"].any() and done["reward_still"].any()
    assert not done["reward_walk"][:5].any()


@pytest.mark.parametrize("name", ["sensors_raw", "sensors_normalised", "sensors_noise_raw", "sensors_noise_normalised"])
def test_sensor_stack_matches_reference(name):
    """The noise cases (a1_sensors_noise.npz) are the reference's sensors built with noise=True (sensor_mode["noise"]), their
    np.random.normal draws recorded as inputs."""
    noisy = "noise" in name
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "a1_sensors_noise.npz" if noisy else "a1_sensors.npz"))
    normal, dt = g[name + "/config"]
    st = oa.SensorStack(int(normal), dt)
    for k in range(len(g[name + "/obs"])):
        obs = st.observe(g[name + "/in_base"][k], g[name + "/in_rpy"][k], g[name + "/in_drpy"][k], g[name + "/in_angles"][k],
                         g[name + "/in_contact"][k], g[name + "/kind"][k] == 0, noise=g[name + "/in_noise"][k] if noisy else None)
        assert np.array_equal(obs, g[name + "/obs"][k]), "%s observation %d" % (name, k)
    if noisy:
        assert (np.abs(g[name + "/in_noise"]).max(axis=0) > 0).all()          # every one of the 33 slots was drawn


@pytest.mark.parametrize("name", ["butter_default", "butter_bandpass", "exp"])
def test_action_filter_matches_reference(name):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "a1_filter.npz"))
    f = oa.ActionFilter(g[name + "/a"], g[name + "/b"])
    for k in range(len(g[name + "/x"])):
        kind, x = g[name + "/kind"][k], g[name + "/x"][k]
        if kind == 1:
            f.reset()
        elif kind == 2:
            f.reset(); f.init_history(x)
        else:
            assert np.array_equal(f.filter(x), g[name + "/y"][k]), "%s sample %d" % (name, k)


def test_butter_coefficients_are_scipys():
    """The product's ActionFilter.butter must hand the kernel the coefficients the reference's ActionFilterButter computes."""
    pytest.importorskip("scipy")
    from scipy.signal import butter
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "a1_filter.npz"))
    b, a = butter(2, [4.0 / (0.5 * (1 / (0.002 * 13)))], btype="low")
    assert np.array_equal(a / a[0], g["butter_default/a"][0]) and np.array_equal(b / a[0], g["butter_default/b"][0])
