"""oracle/make_ref.py — the recipe that byte-compiles the unmodified reference into oracle/_ref for bench.py's `cpu_baseline`
leg (north_star: "the reference CPU path timed on the same box's host cores in the same run")."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HAVE_REFERENCE = os.path.isdir("/root/reference/metagym/quadrotor")


@pytest.mark.skipif(not HAVE_REFERENCE, reason="needs the reference tree (build container only)")
def test_recipe_builds_sourceless_reference_that_steps(tmp_path):
    from oracle import make_ref
    out = str(tmp_path / "_ref")
    m = make_ref.build(out=out, quiet=True)
    assert make_ref.usable(out)
    # nothing but byte-code, the data files the modules read, and the manifest: no reference .py travels
    for dirpath, _d, files in os.walk(out):
        for f in files:
            assert not f.endswith(".py"), f
    assert "metagym/quadrotor/env.py" in m["files"] and len(m["files"]["metagym/quadrotor/env.py"]["sha256_source"]) == 64
    code = ("import sys, numpy as np\n"
            "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "np.int = int\n"
            "import gym\n"
            "from metagym.quadrotor.env import Quadrotor\n"
            "import metagym.quadrotor.env as e\n"
            "assert e.__file__.endswith('env.pyc'), e.__file__\n"
            "np.random.seed(3)\n"
            "env = Quadrotor(task='hovering_control', nt=1000); env.reset()\n"
            "o, r, d, i = env.step(np.array([2.0, 2.1, 2.2, 2.3], np.float32))\n"
            "print(repr(float(r)))\n") % (os.path.join(ROOT, "oracle", "refstubs"), out)
    got = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(tmp_path))
    assert got.returncode == 0, got.stderr
    # the same call against the sources where they lie: identical reward
    code_src = code.replace(repr(out), repr("/root/reference")).replace("assert e.__file__.endswith('env.pyc'), e.__file__\n", "")
    want = subprocess.run([sys.executable, "-c", code_src], capture_output=True, text=True, cwd=str(tmp_path))
    assert want.returncode == 0, want.stderr
    assert got.stdout == want.stdout


def test_bench_reference_root_falls_back_to_the_built_copy(monkeypatch):
    import bench
    monkeypatch.delenv("METAGYM_REFERENCE", raising=False)
    assert bench.reference_root() == os.path.join(ROOT, "oracle", "_ref")
    monkeypatch.setenv("METAGYM_REFERENCE", "/somewhere/else")
    assert bench.reference_root() == "/somewhere/else"
    # an absent tree is reported, not an exception (the C port then stays the baseline, labelled as such)
    res, why = bench.cpu_reference(seconds=0.1)
    assert res is None and "no importable reference" in why


def test_bench_never_reads_the_reference_tree_at_run_time():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"/root/reference"' not in src and "'/root/reference'" not in src


def test_ref_dir_is_ignored_by_git_but_ships_to_the_gpu_box():
    assert "oracle/_ref/" in open(os.path.join(ROOT, ".gitignore")).read().split()
    p = os.path.join(ROOT, ".gpurunignore")
    assert not os.path.exists(p) or "_ref" not in open(p).read()
