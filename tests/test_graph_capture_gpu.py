"""hipGraph capture of the step launches ("K-steps-per-launch / hipGraph capture").

The C ABI only enqueues kernels on the caller's stream — no allocation, no synchronisation, no host
read-back — so a whole step (or several) can be captured into one hipGraph through torch's
`torch.cuda.graph` and replayed with new actions written into the captured action buffer. Replays
must be bit-identical to eager stepping. That includes the fused auto-reset: its noise is keyed by
(seed, global env id, per-env episode counter held in device memory), so no launch argument changes
from step to step and a captured launch replays correctly (ABI 1 passed a host step counter by value,
which capture froze)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _capture(fn):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):          # warm-up outside capture (first-launch lazy module load)
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g


def test_quadrotor_step_graph_replay_is_bit_identical():
    import metagym_amd
    n, K, T = 4096, 4, 5
    mk = lambda: metagym_amd.make("quadrotor-v0", num_envs=n, device="cuda:0", task="hovering_control")
    eager, graphed = mk(), mk()
    eager.reset(seed=3)
    graphed.reset(seed=3)
    sd0 = graphed.state_dict()
    static_a = torch.zeros(K, n, 4, dtype=torch.float32, device="cuda:0")
    outs = []

    def k_steps():
        outs.clear()
        for k in range(K):
            obs, rew, done, info = graphed.step(static_a[k])
            outs.append((obs.clone(), rew.clone(), done.clone()))

    g = _capture(k_steps)
    graphed.load_state_dict(sd0)           # the warm-up and the capture pass advanced the state
    rs = np.random.RandomState(0)
    for t in range(T):
        a = torch.as_tensor(rs.uniform(0.1, 15.0, (K, n, 4)).astype(np.float32)).cuda()
        static_a.copy_(a)
        g.replay()
        for k in range(K):
            obs, rew, done, _ = eager.step(a[k])
            assert torch.equal(outs[k][0], obs) and torch.equal(outs[k][1], rew) and torch.equal(outs[k][2], done)
    se, sg = eager.state_dict(), graphed.state_dict()
    for k in se:
        if torch.is_tensor(se[k]):
            assert torch.equal(se[k], sg[k]), k


def test_quadrotor_autoreset_graph_replay_is_bit_identical():
    """The launch bench.py times, captured: K fused-auto-reset steps per graph, replayed T times with fresh
    actions, episodes ending (ct == nt, nt = 7) inside and across replays."""
    import metagym_amd
    n, K, T, nt = 4096 + 5, 5, 6, 7
    mk = lambda: metagym_amd.make("quadrotor-v0", num_envs=n, device="cuda:0", task="hovering_control", nt=nt,
                                  auto_reset=True, seed=21)
    eager, graphed = mk(), mk()
    eager.reset(seed=3)
    graphed.reset(seed=3)
    sd0 = graphed.state_dict()
    static_a = torch.zeros(K, n, 4, dtype=torch.float32, device="cuda:0")
    outs = []

    def k_steps():
        outs.clear()
        for k in range(K):
            obs, rew, done, info = graphed.step(static_a[k])
            outs.append((obs.clone(), rew.clone(), done.clone()))

    g = _capture(k_steps)
    graphed.load_state_dict(sd0)           # the warm-up and the capture pass advanced state and episode counters
    rs = np.random.RandomState(0)
    n_done = 0
    for t in range(T):
        a = torch.as_tensor(rs.uniform(0.1, 15.0, (K, n, 4)).astype(np.float32)).cuda()
        static_a.copy_(a)
        g.replay()
        for k in range(K):
            obs, rew, done, _ = eager.step(a[k])
            assert torch.equal(outs[k][0], obs) and torch.equal(outs[k][1], rew) and torch.equal(outs[k][2], done)
            n_done += int(done.sum())
    assert n_done == n * (K * T // nt)
    se, sg = eager.state_dict(), graphed.state_dict()
    for k in se:
        if torch.is_tensor(se[k]):
            assert torch.equal(se[k], sg[k]), k
    assert int(se["episode"].min()) == K * T // nt


def test_maze2d_single_env_graph_replay_matches_eager_and_golden():
    """C1 (one 15x15 env) is launch-bound: a run of steps captured as one graph, checked against eager
    stepping and against the rewards / dones of the golden trajectory recorded from the reference."""
    import metagym_amd
    from test_maze_gpu import _task_from_golden
    g = np.load(os.path.join(GOLDEN, "maze2d_escape_s0.npz"))
    mk = lambda: metagym_amd.make("meta-maze-2D-v0", num_envs=1, device="cuda:0", max_steps=int(g["max_steps"]),
                                  view_grid=int(g["view_grid"]), task_type="ESCAPE")
    eager, env = mk(), mk()
    for e in (eager, env):
        e.set_task(_task_from_golden(g))
        e.reset()
    # the stretch of the golden episode before its first reset
    stops = [t for t in range(1, len(g["actions"])) if g["reset_before"][t]]
    T = min(40, stops[0] if stops else len(g["actions"]))
    static_a = torch.zeros(T, 1, dtype=torch.int32, device="cuda:0")
    rec = []

    def run():
        rec.clear()
        for t in range(T):
            obs, rew, done, _ = env.step(static_a[t])
            rec.append((obs.clone(), env.reward64.clone(), done.clone()))

    sd0 = env.state_dict()
    graph = _capture(run)
    env.load_state_dict(sd0)
    env.need_reset = False
    acts = torch.as_tensor(np.asarray(g["actions"][:T]).astype(np.int32)).cuda()[:, None]
    static_a.copy_(acts)
    graph.replay()
    torch.cuda.synchronize()
    for t in range(T):
        obs, rew, done, _ = eager.step(acts[t])
        assert torch.equal(rec[t][0], obs), t
        assert float(rec[t][1][0]) == float(g["reward"][t]) and bool(rec[t][2][0]) == bool(g["done"][t]), t
    se, sg = eager.state_dict(), env.state_dict()
    for k in se:
        assert torch.equal(se[k], sg[k]), k
