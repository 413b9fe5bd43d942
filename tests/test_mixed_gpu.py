"""BASELINE configs[4] / SURVEY.md §8(d) C5 and the north-star batch size under test (GPU box, -m gpu):
  * the mixed two-stream step bench.py times (`bench.MixedStep`: quadrotor launch + maze3d launch on two HIP streams)
    is, bit for bit, each family stepped alone — with episodes ending (fused auto-reset) inside the compared steps;
  * Quadrotor hovering_control at 2^20 envs on one GPU (north_star's total batch; 2^17 is its per-GPU share) and at
    2^17: 256 sampled envs of the fused-auto-reset run against the CPU oracle, bit-exact state and counters."""
import numpy as np
import pytest
import torch

import bench
from oracle import quadrotor as qo

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _maze_state(env):
    return {k: v.clone() for k, v in env.state_dict().items() if torch.is_tensor(v)}


def test_mixed_two_stream_step_equals_each_family_alone():
    dev = torch.device(DEV)
    n, T = 4096, 40
    plan = bench.shard_plan(0, 1, n, "mixed")
    # short episodes so both families restart envs inside the compared steps (quadrotor: clock nt = 7 staggered by
    # env id + floor hits; maze: max_steps = 9 and SURVIVAL deaths)
    mk_q = lambda: bench.QuadrotorShard(dev, plan, n, preroll=0, nt=7)
    mk_m = lambda: bench.MazeShard(dev, plan, n, res=64, max_steps=9)
    quad, maze = mk_q(), mk_m()
    quad_alone, maze_alone = mk_q(), mk_m()
    torch.cuda.synchronize()
    both = bench.MixedStep(dev, quad, maze)
    q_ends = m_ends = 0
    for t in range(T):
        both(t)
        quad_alone.step(t)
        maze_alone.step(t)
        torch.cuda.synchronize()
        for a, b in ((quad.env._obs, quad_alone.env._obs), (quad.env._reward, quad_alone.env._reward),
                     (quad.env._reward64, quad_alone.env._reward64), (quad.env._done, quad_alone.env._done),
                     (quad.env._failed, quad_alone.env._failed),
                     (maze.env._obs, maze_alone.env._obs), (maze.env._reward, maze_alone.env._reward),
                     (maze.env._done, maze_alone.env._done)):
            assert torch.equal(a, b), t
        q_ends += int(quad.env._done.sum())
        m_ends += int(maze.env._done.sum())
    sa, sb = quad.env.state_dict(), quad_alone.env.state_dict()
    for k in ("pos", "vel", "omega", "propw", "rot", "ct", "episode"):
        assert torch.equal(sa[k], sb[k]), k
    ma, mb = _maze_state(maze.env), _maze_state(maze_alone.env)
    assert ma.keys() == mb.keys()
    for k in ma:
        assert torch.equal(ma[k], mb[k]), k
    assert q_ends >= 4 * n and m_ends >= 3 * n        # every env restarted several times inside the compared steps
    assert int(quad.env.episode.min()) >= 4


@pytest.mark.parametrize("n", [1 << 17, 1 << 20])
def test_north_star_batch_sampled_against_oracle(n):
    """The north-star batch (2^20 quadrotors; 2^17 = one GPU's share of it on 8 GPUs) through the launch bench.py
    times — fused auto-reset, clocks staggered by global env id — 256 sampled envs replayed on the CPU oracle:
    final state, clock and episode counter bit-exact, with clock wraps and floor hits inside."""
    import metagym_amd
    nt, T, seed, base = 13, 30, 1000, 3 << 20
    env = metagym_amd.make("quadrotor-v0", num_envs=n, device=DEV, task="hovering_control", nt=nt, auto_reset=True,
                           seed=seed, env_id_base=base)
    env.reset(seed=5)
    ids = torch.arange(n, device=DEV, dtype=torch.int64) + base
    sd = env.state_dict()
    sd["ct"] = ((ids * 977) % nt).to(torch.int32)
    sd["pos"][2, ::5] = -4.98                        # a fifth start 2 cm above the floor: collision episode ends
    env.load_state_dict(sd)
    idx = np.sort(np.random.RandomState(n & 0xFFFF).choice(n, 256, replace=False))
    idx[0], idx[-1] = 0, n - 1                       # both ends of the batch (last wave, last lane)
    idx_t = torch.as_tensor(idx, device=DEV)
    st0 = {k: sd[k][..., idx_t].cpu().numpy() for k in ("pos", "vel", "omega", "propw", "rot", "ct")}
    g = torch.Generator(device=DEV)
    g.manual_seed(11)
    acts_s = np.zeros((T, 256, 4), np.float32)
    dones = torch.zeros(n, dtype=torch.int64, device=DEV)
    for t in range(T):
        a = torch.rand(n, 4, device=DEV, generator=g) * 14.9 + 0.1
        acts_s[t] = a[idx_t].cpu().numpy()
        obs, rew, done, info = env.step(a)
        dones += done
    assert int(dones.min()) >= 2 and int(info["failed"].max()) == 0
    fin = env.state_dict()
    c = qo.default_consts(nt=nt)
    collided = 0
    for j, e in enumerate(idx):
        st = qo.make_states(st0["pos"].T[[j]], st0["vel"].T[[j]], st0["omega"].T[[j]], st0["propw"].T[[j]], st0["rot"].T[[j]])
        ct, epo = st0["ct"][[j]].astype(np.int32), np.zeros(1, np.uint32)
        ar = qo.default_autoreset(seed=seed, env_id_base=base + int(e))
        for t in range(T):
            before = int(ct[0])
            out = qo.batch_env_step_autoreset(c, ar, st, ct, epo, acts_s[t][[j]])
            collided += int(out[2][0] != 0 and before + 1 != nt)
        o = qo.states_to_arrays(st)
        for k, kk in (("pos", "pos"), ("vel", "vel"), ("omega", "omega"), ("propw", "propw"), ("rot", "R")):
            assert np.array_equal(fin[k][..., e].cpu().numpy(), o[kk][0]), (k, e)
        assert int(fin["ct"][e]) == int(ct[0]) and int(fin["episode"][e]) == int(epo[0])
    assert collided > 0
