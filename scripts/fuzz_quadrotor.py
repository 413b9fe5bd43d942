"""Differential fuzzer: quadrotor_step_kernel (through the C ABI) against the CPU oracle over random
SIMULATOR CONFIGS (inertia with / without off-diagonals, drag, centre of gravity, thrust polynomial,
propeller geometry, precision -> 1..20 sub-steps, voltage range, fail limits), random states incl. ones
that trip the failure test, hovering_control and no_collision. Bit-exact comparison of the state,
reward, done and failure code (observation angles: 4 ulp, OCML vs glibc atan2f). GPU box only.

    python scripts/fuzz_quadrotor.py [--configs 80] [--seed 0]"""
import argparse
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def random_config(rs, stock_shape):
    f = lambda lo, hi: float(rs.uniform(lo, hi))
    off = (lambda: 0.0) if stock_shape else (lambda: f(-8e-4, 8e-4))
    prec = float(rs.choice([0.0005, 0.001, 0.002, 0.005, 0.01]))
    return {
        "precision": prec, "quality": float(rs.choice([0.5, 0.62, 1.0, 0.35])),
        "inertia": {"xx": f(0.01, 0.02), "xy": off(), "xz": off(), "yy": f(0.01, 0.02), "yz": off(), "zz": f(0.02, 0.03)},
        "drag": {"m_xx": f(0.05, 0.09), "m_yy": f(0.05, 0.09), "m_zz": f(0.04, 0.06), "f_xx": f(0.1, 0.14),
                 "f_yy": f(0.1, 0.14), "f_zz": f(0.08, 0.12)},
        "gravity_center": {"x": 0.0 if stock_shape else f(-5e-3, 5e-3), "y": 0.0 if stock_shape else f(-5e-3, 5e-3),
                           "z": 0.0 if stock_shape else f(-5e-3, 5e-3)},
        "thrust": {"CT": ["%.6e" % f(1.4e-5, 1.7e-5), "%.6e" % f(-3e-4, -2e-4), "0.0" if stock_shape else "%.3e" % f(0, 5e-4)],
                   "Mm": "%.4f" % f(0.008, 0.012), "Jm": "%.4e" % f(2.3e-4, 2.8e-4), "RA": "%.4f" % f(0.19, 0.22),
                   "phi": "%.8f" % f(0.016, 0.018)},
        "propeller": [{"x": sx * f(0.16, 0.2), "y": sy * f(0.16, 0.2), "z": 0.0 if stock_shape else f(-0.01, 0.01)}
                      for sx, sy in ((1, 1), (-1, 1), (-1, -1), (1, -1))],
        "fail": {"velocity": float(rs.choice([100.0, 6.0])), "w": float(rs.choice([1000.0, 8.0])),
                 "range": float(rs.choice([1000.0, 40.0]))},
        "electric": {"min_voltage": f(0.05, 0.5), "max_voltage": f(10.0, 16.0)},
        "init_velocity": {"x": 0, "y": 0, "z": 0, "noisy": 2.0},
        "init_angular_velocity": {"x": 0, "y": 0, "z": 0, "noisy": 5.0},
    }


def main():
    import torch
    import metagym_amd
    from oracle import quadrotor as qo
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", type=int, default=80)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    rs = np.random.RandomState(args.seed)
    n, T = 512, 4
    fails = 0
    nonang = [i for i in range(16) if i not in (12, 13, 14)]
    for c in range(args.configs):
        stock_shape = bool(rs.rand() < 0.4)          # 40 %: structure of the stock config -> SIMPLE kernel
        cfg = random_config(rs, stock_shape)
        task = "hovering_control" if rs.rand() < 0.6 else "no_collision"
        nt = int(rs.choice([3, 1000]))
        with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
            json.dump(cfg, f)
            path = f.name
        env = metagym_amd.make("quadrotor-v0", num_envs=n, device="cuda:0", task=task, nt=nt, simulator_conf=path)
        os.unlink(path)
        oc = qo.consts_from_config(cfg, nt=nt, task=qo.TASK_HOVERING if task == "hovering_control" else qo.TASK_NO_COLLISION)
        pos = (rs.uniform(-30, 30, (n, 3)) * [1, 1, 0.15]).astype(np.float32)
        vel = rs.uniform(-5, 5, (n, 3))
        omega = rs.uniform(-6, 6, (n, 3))
        propw = rs.uniform(0, 600, (n, 4)).astype(np.float32)
        R = np.tile(np.eye(3, dtype=np.float32).reshape(9), (n, 1)) + rs.uniform(-0.05, 0.05, (n, 9)).astype(np.float32)
        sd = dict(pos=torch.as_tensor(np.ascontiguousarray(pos.T)), vel=torch.as_tensor(np.ascontiguousarray(vel.T)),
                  omega=torch.as_tensor(np.ascontiguousarray(omega.T)), propw=torch.as_tensor(np.ascontiguousarray(propw.T)),
                  rot=torch.as_tensor(np.ascontiguousarray(R.T)), ct=torch.zeros(n, dtype=torch.int32))
        env.load_state_dict(sd)
        st = qo.make_states(pos, vel, omega, propw, R)
        ct = np.zeros(n, np.int32)
        bad = []
        n_failed = 0
        for t in range(T):
            a = rs.uniform(-0.5, 16.5, (n, 4)).astype(np.float32)
            obs, rew, done, info = env.step(torch.as_tensor(a))
            o_obs, o_rew, o_done, o_failed = qo.batch_env_step(oc, st, ct, a)
            o = qo.states_to_arrays(st)
            g = env.state_dict()
            for k, gk in (("pos", "pos"), ("vel", "vel"), ("omega", "omega"), ("propw", "propw"), ("R", "rot")):
                if not np.array_equal(g[gk].T.cpu().numpy(), o[k]):
                    bad.append((t, k))
            if not np.array_equal(g["ct"].cpu().numpy(), ct): bad.append((t, "ct"))
            if not np.array_equal(env.reward64.cpu().numpy(), o_rew): bad.append((t, "reward"))
            if not np.array_equal(done.cpu().numpy(), o_done.astype(bool)): bad.append((t, "done"))
            if not np.array_equal(info["failed"].cpu().numpy(), o_failed.astype(np.uint8)): bad.append((t, "failed"))
            go = obs.cpu().numpy()
            if not np.array_equal(go[:, nonang], o_obs[:, nonang]): bad.append((t, "obs"))
            if np.max(np.abs(go[:, 12:15] - o_obs[:, 12:15])) > 4 * np.spacing(np.float32(np.pi)): bad.append((t, "angles"))
            n_failed += int((o_failed != 0).sum())
        fails += 1 if bad else 0
        print("cfg %3d %-16s %s precision=%.4f quality=%.2f nt=%4d sim-failures %4d  %s"
              % (c, task, "stock-shape" if stock_shape else "general    ", cfg["precision"], cfg["quality"], nt, n_failed,
                 "ok" if not bad else "MISMATCH %r" % (bad[:4],)), flush=True)
        del env
    print("configs with a mismatch: %d / %d" % (fails, args.configs))
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
