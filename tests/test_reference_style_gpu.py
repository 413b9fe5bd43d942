"""The reference's own smoke tests, re-run against the batched engine with the same call sequence
(metagym/quadrotor/tests/test_env.py:19-52, metagym/metamaze/test.py:9-75): make -> [set_task] -> reset ->
step until done, with `action_space.sample()` actions. num_envs = 1 is the drop-in case; a second pass runs
the same loops with a batch. GPU box only."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _act(env, n):
    return torch.as_tensor(np.asarray(env.action_space.sample(n)))


@pytest.mark.parametrize("n", [1, 257])
@pytest.mark.parametrize("task", ["no_collision", "hovering_control"])
def test_quadrotor_tasks_run_to_episode_end(task, n):
    import metagym_amd
    env = metagym_amd.make("quadrotor-v0", num_envs=n, device="cuda:0", task=task)
    state = env.reset()
    assert tuple(state.shape) == (n, 16)
    ended = torch.zeros(n, dtype=torch.bool, device="cuda:0")
    steps = 0
    while not bool(ended.all()):
        state, reward, reset, info = env.step(_act(env, n).to(torch.float32))
        ended |= reset
        steps += 1
        assert steps <= env.nt
    assert steps >= 2


def test_quadrotor_velocity_control_runs_exactly_nt_steps():
    """test_env.py:30-41: constant action, `next_target_g_v_x` in info, the episode ends after nt steps."""
    import metagym_amd
    env = metagym_amd.make("quadrotor-v0", num_envs=1, device="cuda:0", task="velocity_control", nt=120)
    env.reset()
    reset, step = False, 0
    while not reset:
        state, reward, reset_t, info = env.step(torch.ones(1, 4))
        assert "next_target_g_v_x" in info
        reset = bool(reset_t[0])
        step += 1
    assert step == env.nt


@pytest.mark.parametrize("name,kw", [("meta-maze-2D-v0", dict(view_grid=1)),
                                     ("meta-maze-discrete-3D-v0", dict(resolution=(64, 64))),
                                     ("meta-maze-continuous-3D-v0", dict(resolution=(64, 64)))])
@pytest.mark.parametrize("task_type", ["ESCAPE", "SURVIVAL"])
def test_maze_envs_run_episodes_with_growing_mazes(name, kw, task_type):
    """metamaze/test.py: sample a task, set it, play an episode to `done`, then increase n by 2 and repeat."""
    import metagym_amd
    from metagym_amd.metamaze import MazeTaskSampler
    n_envs = 3
    env = metagym_amd.make(name, num_envs=n_envs, device="cuda:0", max_steps=60, task_type=task_type, **kw)
    n = 9
    for iteration in range(3):
        env.set_task(MazeTaskSampler(n=n, step_reward=-0.01, goal_reward=1.0, allow_loops=False, seed=iteration))
        env.reset()
        ended = torch.zeros(n_envs, dtype=torch.bool, device="cuda:0")
        sum_reward = torch.zeros(n_envs, device="cuda:0")
        guard = 0
        while not bool(ended.all()):
            a = env.action_space.sample(n_envs)
            state, reward, done, _ = env.step(torch.as_tensor(np.asarray(a)))
            sum_reward += reward * (~ended)
            ended |= done
            if bool(done.any()) and not bool(ended.all()):
                env.reset(mask=done)            # the reference raises on step-after-done; here: masked reset
            guard += 1
            assert guard <= 61
        assert torch.isfinite(sum_reward).all()
        n += 2


@pytest.mark.parametrize("env_id,prefix", [("meta-humanoid-v0", "humanoid"), ("meta-ant-v0", "ant")])
def test_locomotion_envs_run_tasks_to_done(env_id, prefix):
    """metalocomotion/test.py:4-28: for several tasks: set_task -> reset -> step(action_space.sample()) until done,
    with max_steps=2. The body variants come from the parsed-model fixture (the MJCF files live in the reference)."""
    import metagym_amd
    from walker_fixtures import load_models
    models = load_models()
    variants = [k for k in sorted(models) if k.startswith(prefix)][:4]
    n = 5
    env = metagym_amd.make(env_id, num_envs=n, device="cuda:0", max_steps=2)
    for name in variants:
        env.set_task(models[name])
        obs = env.reset()
        assert tuple(obs.shape) == (n, env.obs_dim)
        done = torch.zeros(n, dtype=torch.bool, device="cuda:0")
        steps = 0
        while not bool(done.all()):
            obs, r, d, info = env.step(torch.as_tensor(np.asarray(env.action_space.sample(n))))
            done |= d
            steps += 1
            assert steps <= 2
        assert torch.isfinite(obs).all() and torch.isfinite(r).all()
    env.close()


def test_every_env_family_takes_the_index_less_device_spelling():
    """`device="cuda"` (what a user types; tensors then report `cuda:0`) must mean the same as `device="cuda:0"`: the envs
    canonicalise the spelling once (`_lib.canonical_device`). Round 5's final validation found `quadrupedal-v0` refusing its own
    physics' tensors ("SoA tensor must be ... on cuda, got ... on cuda:0") and Quadrotor.step taking its conversion path on every
    call. One reset + a few steps per family, with device tensors as actions, and the fast-path precondition of Quadrotor.step."""
    import metagym_amd
    from metagym_amd.metamaze import MazeTaskSampler
    from urdf_fixture import a1_like_urdf
    n = 64
    q = metagym_amd.make("quadrotor-v0", num_envs=n, device="cuda", task="hovering_control", auto_reset=True)
    assert q.device == torch.device("cuda", torch.cuda.current_device()) and q.reset(seed=0).device == q.device
    a = torch.rand(n, 4, device="cuda") * 10 + 1
    assert a.device == q.device                                    # the zero-copy path of step() is taken
    q.step(a)
    for name in ("meta-maze-2D-v0", "meta-maze-discrete-3D-v0", "meta-maze-continuous-3D-v0"):
        kw = {} if "2D" in name else dict(resolution=(32, 32))
        m = metagym_amd.make(name, num_envs=n, device="cuda", max_steps=10, task_type="ESCAPE", **kw)
        m.set_task([MazeTaskSampler(n=9, allow_loops=False, seed=s_) for s_ in range(3)])
        m.reset()
        act = torch.rand(n, 2, device="cuda") * 2 - 1 if "continuous" in name else torch.randint(0, 4, (n,), device="cuda")
        obs = m.step(act)[0]
        assert obs.device == m.device
    for ident in ("meta-humanoid-v0", "meta-ant-v0"):
        w = metagym_amd.make(ident, num_envs=n, device="cuda", auto_reset=True)
        w.set_task([w.sample_task("TRAIN") for _ in range(4)])
        w.reset(seed=0)
        assert torch.isfinite(w.step(torch.rand(n, w.n_joints, device="cuda") * 2 - 1)[0]).all()
    a1 = metagym_amd.make("quadrupedal-v0", num_envs=n, device="cuda", urdf=a1_like_urdf(), task="slopestair", auto_reset=True)
    obs, info = a1.reset()
    for _ in range(3):
        obs, reward, done, info = a1.step(torch.zeros(n, 12, dtype=torch.float64, device="cuda"))
    assert torch.isfinite(obs).all() and obs.device == a1.device
