"""Differential fuzzer: the A1 actuation kernels (mg_a1_apply_action / receive_observation / receive_and_apply, four lanes
per robot) against the CPU oracle (oracle/a1.py, itself pinned to the unmodified reference) — random batch sizes, motor
modes, per-robot latencies / gains, strength ratios, torque limits, command clip, action interpolation, history lengths,
masked resets. Every torque of every sub-step and every control observation must be bit-identical.

    python scripts/fuzz_a1.py --configs 60 --seed 1        (GPU box)"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from metagym_amd.quadrupedal import A1Actuators, MotorControlMode
from oracle import a1 as oa

DEV = "cuda:0"


def dev(*xs):
    return [torch.as_tensor(x, dtype=torch.float64, device=DEV) for x in xs]


def one(rs, idx):
    n = int(rs.choice([1, 3, 15, 16, 17, 63, 64, 65, 1000, 4133, int(rs.randint(2, 9000))]))
    mode = int(rs.choice([oa.POSITION, oa.HYBRID, oa.TORQUE]))
    repeat = int(rs.randint(1, 14))
    hist = int(rs.randint(2, 40))
    per_env = rs.rand() < 0.6
    clat = rs.uniform(0.0, 0.04, n) if per_env else float(rs.choice([0.0, 0.002, 0.0137, 0.05]))
    plat = (np.where(rs.rand(n) < 0.5, 0.0, rs.uniform(0.0, 0.008, n)) if per_env else float(rs.choice([0.0, 0.0, 0.003])))
    kp = rs.uniform(40, 120, (n, 12)) if per_env else rs.uniform(40, 120, 12)
    kd = rs.uniform(0.3, 4, (n, 12)) if per_env else rs.uniform(0.3, 4, 12)
    strength = rs.uniform(0.5, 1.0, 12)
    limit = float(rs.choice([0.0, 15.0, 33.5]))
    interp, clip = bool(rs.rand() < 0.5), bool(mode == oa.POSITION and rs.rand() < 0.5)
    o = oa.A1Actuation(n, 0.002, repeat, clat, plat, mode, kp, kd, strength, limit if limit else None, interpolate=interp, clip=clip,
                       history_len=hist)
    a = A1Actuators(n, DEV, action_repeat=repeat, motor_control_mode=MotorControlMode(mode), motor_torque_limits=limit if limit else None,
                    enable_action_interpolation=interp, enable_clip_motor_commands=clip, history_len=hist)
    a.SetControlLatency(torch.as_tensor(clat) if per_env else clat)
    a.SetPDLatency(torch.as_tensor(plat) if per_env else plat)
    a.SetMotorGains(torch.as_tensor(kp), torch.as_tensor(kd))
    a.SetMotorStrengthRatios(strength)

    def world():
        return rs.uniform(-1, 1, (n, 12)), rs.uniform(-8, 8, (n, 12)), rs.uniform(-1, 1, (n, 4)), rs.uniform(-2, 2, (n, 3))

    w = world()
    o.reset(); a.Reset()
    o.receive_observation(*w); a.ReceiveObservation(*dev(*w))
    cdim = 60 if mode == oa.HYBRID else 12
    steps = int(rs.randint(2, 7))
    checked = 0
    for s in range(steps):
        action = rs.uniform(-2, 2, (n, cdim)) if mode != oa.HYBRID else rs.uniform(0.2, 90, (n, cdim))
        seq = []

        def physics(torque, seq=seq):
            ww = world()
            seq.append((torque.cpu().numpy().copy(), ww))
            return dev(*ww)

        a.Step(dev(action)[0], physics)
        for i in range(repeat):
            t = o.apply_action(o.process_action(action, i))
            if not np.array_equal(t, seq[i][0]):
                return "torque differs: config %d step %d sub-step %d (n=%d mode=%d)" % (idx, s, i, n, mode), checked
            o.receive_observation(*seq[i][1])
            checked += n
        o.last_action = action
        if not np.array_equal(a.GetControlObservation().cpu().numpy(), o.control_obs):
            return "control observation differs: config %d step %d (n=%d mode=%d)" % (idx, s, n, mode), checked
        if rs.rand() < 0.3:          # Minitaur.Reset for a subset, then the first observation of the new episode
            clear = rs.rand(n) < 0.3
            ww = world()
            o.count[clear] = 0; o.observed_torque[clear] = 0.0
            a.Reset(mask=torch.as_tensor(clear)); a._last_action = torch.as_tensor(action.T.copy(), device=DEV)
            a._step_counter = 1
            o.receive_observation(*ww); a.ReceiveObservation(*dev(*ww))
            if not np.array_equal(a.GetControlObservation().cpu().numpy(), o.control_obs):
                return "control observation after a masked reset differs: config %d step %d" % (idx, s), checked
    return None, checked


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    rs = np.random.RandomState(args.seed)
    bad, total = 0, 0
    for c in range(args.configs):
        err, checked = one(rs, c)
        total += checked
        if err:
            bad += 1
            print(err, flush=True)
    print("A1 actuation, HIP vs oracle: %d / %d configs with a difference, %d robot-sub-steps compared bit for bit" % (bad, args.configs, total))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
