"""Pin oracle/maze_oracle.c against the golden vectors recorded from the unmodified reference
(oracle/gen_golden_maze.py -> tests/golden/maze*.npz). CPU-only."""
import glob
import os

import numpy as np
import pytest

from oracle import maze as mo

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _files(pattern):
    return sorted(glob.glob(os.path.join(GOLDEN, pattern)))


def _task_type(path):
    return mo.SURVIVAL if "survival" in os.path.basename(path) else mo.ESCAPE


@pytest.mark.parametrize("path", _files("maze2d_*.npz"))
def test_maze2d_bit_exact(path):
    """MetaMaze2D (config C1): grid, reward (f64), done, steps, life and the float32 window are
    bit-identical to the reference for 300 steps incl. resets."""
    g = np.load(path)
    tt = _task_type(path)
    task = mo.Task.from_golden(g)
    st = mo.State(task)
    vg, max_steps = int(g["view_grid"]), int(g["max_steps"])
    mo.reset(task, tt, st)
    assert np.array_equal(mo.observe_2d(task, tt, st, vg), g["obs0"])
    oi = 0
    for t, a in enumerate(g["actions"]):
        if g["reset_before"][t]:
            mo.reset(task, tt, st)
        r, d = mo.step_2d(task, tt, max_steps, st, a)
        assert list(st.c.grid) == list(g["grid"][t]), t
        assert r == g["reward"][t] and d == bool(g["done"][t]) and st.c.steps == g["steps"][t], t
        if tt == mo.SURVIVAL:
            assert st.c.life == g["life"][t], t
        if oi < len(g["obs_step"]) and g["obs_step"][oi] == t:
            assert np.array_equal(mo.observe_2d(task, tt, st, vg), g["obs"][oi]), t
            oi += 1
    assert oi == len(g["obs_step"])


def _view(g):
    tex = np.load(os.path.join(GOLDEN, "maze_textures.npz"))
    H, V = (int(x) for x in g["resolution"])
    kw = {}
    if "max_vision" in g.files:      # goldens recorded with non-default renderer parameters
        kw = dict(max_vision=float(g["max_vision"]), fov=float(g["fol_angle"]))
    return mo.View(tex["grounds"], tex["ceil"], H, V, **kw)


@pytest.mark.parametrize("path", _files("maze3d_disc_*.npz"))
def test_maze3d_discrete_bit_exact(path):
    """MetaMazeDiscrete3D (config C3): transitions bit-exact AND every pixel of every recorded
    int32 frame identical (floor, ceiling, walls, translucent overlays, life bar)."""
    g = np.load(path)
    tt = _task_type(path)
    task = mo.Task.from_golden(g)
    st = mo.State(task)
    view = _view(g)
    max_steps = int(g["max_steps"])
    mo.reset(task, tt, st)
    assert np.array_equal(mo.observe_3d(task, tt, view, st, 0), g["obs0"])
    oi = 0
    for t, a in enumerate(g["actions"]):
        if g["reset_before"][t]:
            mo.reset(task, tt, st)
        r, d = mo.step_disc3d(task, tt, max_steps, st, a)
        assert list(st.c.grid) == list(g["grid"][t]) and st.c.ori_idx == g["ori_idx"][t], t
        assert r == g["reward"][t] and d == bool(g["done"][t]) and st.c.steps == g["steps"][t], t
        if tt == mo.SURVIVAL:
            assert st.c.life == g["life"][t], t
        if oi < len(g["obs_step"]) and g["obs_step"][oi] == t:
            img = mo.observe_3d(task, tt, view, st, 0)
            bad = int((img != g["obs"][oi]).sum())
            assert bad == 0, "step %d: %d differing values, max |d| %d" % (
                t, bad, int(np.abs(img - g["obs"][oi]).max()))
            oi += 1
    assert oi == len(g["obs_step"])


@pytest.mark.parametrize("path", _files("maze3d_cont_*.npz"))
def test_maze3d_continuous(path):
    """MetaMazeContinuous3D: location/heading within 1e-5 (float dynamics; sin/cos come from libm
    here and from numpy in the reference), grid/reward/done exact, frames >= 99.9 % identical."""
    g = np.load(path)
    tt = _task_type(path)
    task = mo.Task.from_golden(g)
    st = mo.State(task)
    view = _view(g)
    max_steps = int(g["max_steps"])
    mo.reset(task, tt, st)
    assert np.array_equal(mo.observe_3d(task, tt, view, st, 1), g["obs0"])
    oi, total, bad, worst = 0, 0, 0, 0
    for t, a in enumerate(g["actions"]):
        if g["reset_before"][t]:
            mo.reset(task, tt, st)
        r, d = mo.step_cont3d(task, tt, max_steps, st, float(a[0]), float(a[1]))
        assert list(st.c.grid) == list(g["grid"][t]), t
        assert r == g["reward"][t] and d == bool(g["done"][t]), t
        assert np.allclose(list(st.c.loc), g["loc"][t], rtol=1e-5, atol=1e-5), t
        assert abs(st.c.ori - g["ori"][t]) <= 1e-5 * max(1.0, abs(g["ori"][t])), t
        if oi < len(g["obs_step"]) and g["obs_step"][oi] == t:
            img = mo.observe_3d(task, tt, view, st, 1)
            diff = img != g["obs"][oi]
            bad += int(diff.sum())
            total += img.size
            if diff.any():
                worst = max(worst, int(np.abs(img - g["obs"][oi]).max()))
            oi += 1
    print(os.path.basename(path), "mismatching values %d / %d (%.5f %%), max |d| %d" % (bad, total, 100.0 * bad / total, worst))
    assert bad <= 1e-3 * total
