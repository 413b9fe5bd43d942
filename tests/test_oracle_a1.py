"""CPU: the A1 actuation oracle (oracle/a1.py) against the vectors recorded from the unmodified reference
(tests/golden/a1_actuation.npz, oracle/gen_golden_a1.py): applied torques, control observation, sensor getters
bit-identical over every sub-step of all eight cases; energy within 1 ulp-scale of np.dot's association."""
import os

import numpy as np
import pytest

from oracle import a1 as oa

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "a1_actuation.npz")


@pytest.fixture(scope="module")
def g():
    return np.load(GOLDEN)


def replay(g, name, act):
    """Drive `act` (anything with the oracle's interface) with the recorded inputs; yield per-step outputs."""
    dt, repeat, clat, plat, interp, clip, mode, n_steps = g[name + "/config"]
    repeat, n_steps = int(repeat), int(n_steps)
    first = g[name + "/first_obs"][0]
    act.reset()
    # a1.A1.__init__ observes twice before the first Step (Reset -> _SettleDownForReset, then minitaur.py:226): the
    # history starts with two identical entries, and a blend of two equal values is not always that value bit for bit
    for _ in range(int(g[name + "/n_history_at_start"][0])):
        act.receive_observation(first[None, 0:12], first[None, 12:24], first[None, 36:40], first[None, 40:43])
    k = 0
    for s in range(n_steps):
        action = g[name + "/action"][s][None]
        for i in range(repeat):
            cmd = act.process_action(action, i)
            np.testing.assert_array_equal(cmd[0], g[name + "/command"][k], err_msg="%s processed command, sub-step %d" % (name, k))
            torque = act.apply_action(cmd)
            true = g[name + "/true_obs"][k]
            act.receive_observation(true[None, 0:12], true[None, 12:24], true[None, 36:40], true[None, 40:43])
            yield "sub", k, torque[0], act
            k += 1
        act.last_action = action
        yield "step", s, None, act


@pytest.mark.parametrize("idx", range(8))
def test_oracle_reproduces_reference_actuation(g, idx):
    name = str(g["cases"][idx])
    act = oa.from_golden(g, name)
    n_sub = 0
    for kind, k, torque, a in replay(g, name, act):
        if kind == "sub":
            assert np.array_equal(torque, g[name + "/torque"][k]), "%s torque, sub-step %d" % (name, k)
            assert np.array_equal(a.observed_torque[0], g[name + "/observed_torque"][k])
            assert np.array_equal(a.control_obs[0], g[name + "/control_obs"][k]), "%s control obs, sub-step %d" % (name, k)
            n_sub += 1
        else:
            ang, vel, tor, rate, energy = a.sensors()
            assert np.array_equal(ang[0], g[name + "/motor_angles"][k])
            assert np.array_equal(vel[0], g[name + "/motor_velocities"][k])
            assert np.array_equal(tor[0], g[name + "/motor_torques"][k])
            assert np.array_equal(rate[0], g[name + "/rpy_rate"][k])
            assert energy[0] == pytest.approx(g[name + "/energy"][k], rel=1e-14, abs=1e-300)
    assert n_sub == len(g[name + "/torque"])


def test_goldens_exercise_every_branch(g):
    """The fixture really contains: torque saturation, a latency longer than the history held so far (oldest entry
    returned), more than 100 sub-steps (the deque wraps), interpolated and clipped commands, all three modes."""
    assert np.any(np.abs(g["position_saturating_clip/torque"]) == 20.0)
    assert np.any(np.abs(g["hybrid_saturating/torque"]) == 33.5)
    assert len(g["position_history_wrap/torque"]) > 100
    assert not np.array_equal(g["position_latency_interp/command"][13], g["position_latency_interp/action"][1])   # lerp 1/13
    assert np.array_equal(g["position_latency_interp/command"][25], g["position_latency_interp/action"][1])      # lerp 1
    assert {int(g[str(c) + "/config"][6]) for c in g["cases"]} == {1, 2, 3}
    early = g["position_long_latency/control_obs"][:10]            # 0.045 s = 22 sub-steps of delay, history shorter
    assert np.array_equal(early[0], g["position_long_latency/first_obs"][0])
