"""CPU-side checks of the drop-in boundary: the C-ABI library builds/loads without a GPU and exports
every symbol include/metagym_hip.h declares; the binding table matches the header; the product
refuses to run without a GPU instead of falling back."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "metagym_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mg_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from metagym_amd import _lib
    lib = _lib.load()
    syms = _header_symbols()
    assert len(syms) >= 7
    for s in syms:
        assert hasattr(lib, s), "libmetagym_hip.so does not export %s" % s
    assert sorted(_lib.SIGNATURES) == syms, "ctypes binding table and header disagree"
    assert lib.mg_abi_version() == _lib.ABI_VERSION
    assert lib.mg_target_arch() == b"gfx950"


def test_default_config_matches_reference_constants():
    """mg_quadrotor_default_config == metagym/quadrotor/config.json (values restated in the package)."""
    from metagym_amd import _lib
    from metagym_amd.quadrotor.env import DEFAULT_SIM_CONFIG, _fill_config
    lib = _lib.load()
    a = _lib.QuadrotorConfig()
    assert lib.mg_quadrotor_default_config(a) == 0
    b = _lib.QuadrotorConfig()
    _fill_config(b, DEFAULT_SIM_CONFIG, 0.01, 1000, "no_collision", 1.0)
    for name, _t in _lib.QuadrotorConfig._fields_:
        if name in ("map_d", "map_h", "map_w", "x_offset", "y_offset", "z_offset"):
            continue
        va, vb = getattr(a, name), getattr(b, name)
        if hasattr(va, "__len__"):
            assert list(va) == list(vb), name
        else:
            assert va == vb, name


def test_argument_errors_are_codes_not_crashes():
    from metagym_amd import _lib
    lib = _lib.load()
    cfg = _lib.QuadrotorConfig()
    lib.mg_quadrotor_default_config(cfg)
    st = _lib.QuadrotorState()
    rc = lib.mg_quadrotor_step(cfg, 0, st, None, None, None, None, None, None, None)
    assert rc == -1001 and b"NULL" in lib.mg_last_error()
    assert lib.mg_quadrotor_default_config(None) == -1001
    # a plan is validated when it is made and when it is used
    plan = _lib.QuadrotorPlan()
    assert lib.mg_quadrotor_plan_step(plan, 1, None, None, None, None, None, None, None) == -1003   # not initialised
    assert lib.mg_quadrotor_plan_init(plan, cfg, None, 4, st) == -1001                              # NULL state arrays
    import ctypes as C
    fake = C.create_string_buffer(64)
    for name, _ in _lib.QuadrotorState._fields_:
        setattr(st, name, C.addressof(fake) if name != "episode" else None)
    assert lib.mg_quadrotor_plan_init(plan, cfg, None, 4, st) == 0            # host-only: folds cfg, launches nothing
    ar = _lib.QuadrotorAutoReset()
    assert lib.mg_quadrotor_plan_init(plan, cfg, ar, 4, st) == -1001 and b"episode" in lib.mg_last_error()
    cfg.precision = 1.0                                                        # > dt (quadrotorsim.py:299-300)
    assert lib.mg_quadrotor_plan_init(plan, cfg, None, 4, st) == -1003


def test_no_cpu_fallback():
    import metagym_amd
    from metagym_amd._lib import MetaGymHipError
    with pytest.raises(MetaGymHipError):
        metagym_amd.make("quadrotor-v0", num_envs=2, device="cpu")


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under metagym_amd/ may reference it."""
    pkg = os.path.join(ROOT, "metagym_amd")
    for dirpath, _d, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "libmetagym_oracle" not in src, f
                # oracle/_ref (the byte-compiled reference, oracle/make_ref.py) is for bench.py's cpu_baseline leg only
                assert "_ref" + os.sep not in src and "oracle/_ref" not in src and "make_ref" not in src, f
                assert not re.search(r"^\s*(from|import)\s+metagym\b", src, flags=re.M), f


def test_integration_stub_structs_match_the_abi():
    """The ctypes structs of the binding stub in INTEGRATION.md (what a maintainer of the reference would
    paste) must have the layout of the real ones."""
    import ctypes as C
    import re
    from metagym_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    m = re.search(r"(class Cfg\(C\.Structure\):.*?)\nclass QuadrotorHIP", text, re.S)
    assert m, "stub structs not found in INTEGRATION.md"
    ns = {"C": C}
    exec(m.group(1), ns)
    assert C.sizeof(ns["Cfg"]) == C.sizeof(_lib.QuadrotorConfig)
    assert [f[0] for f in ns["Cfg"]._fields_] == [f[0] for f in _lib.QuadrotorConfig._fields_]
    for name, _ in ns["Cfg"]._fields_:
        assert getattr(ns["Cfg"], name).offset == getattr(_lib.QuadrotorConfig, name).offset, name
    assert C.sizeof(ns["State"]) == C.sizeof(_lib.QuadrotorState)
    assert [f[0] for f in ns["State"]._fields_] == [f[0] for f in _lib.QuadrotorState._fields_]
    assert "mg_abi_version() == %d" % _lib.ABI_VERSION in text


def test_walker_wave_mapping_rejects_topologies_its_lds_scratch_cannot_hold():
    """Argument validation happens before any launch, so it can be exercised without a GPU: 16 bodies welded
    together (no joints: 6 generalized coordinates) need more mass-matrix assembly scratch (16 doubles per body + 12
    per coordinate) than the wave mapping's 36 x 6 constraint block offers -> MG_ERR_UNSUPPORTED with a message,
    not a silent LDS overrun."""
    import ctypes as C
    from metagym_amd import _lib
    lib = _lib.load()
    tp = _lib.WalkerTopology()
    tp.n_bodies, tp.n_joints, tp.n_spheres, tp.n_feet = 16, 0, 0, 0
    for b in range(16):
        tp.body_parent[b] = b - 1
    ms = _lib.WalkerModels()
    fake = C.create_string_buffer(8)
    ms.table, ms.n_tasks, ms.model_stride = C.addressof(fake), 1, 25 * 16
    prm = _lib.WalkerParams()
    prm.time_step, prm.frame_skip, prm.solver_iterations, prm.mapping = 0.005, 4, 5, 1
    st = _lib.WalkerState()
    for name, _ in _lib.WalkerState._fields_:
        setattr(st, name, C.addressof(fake))
    p = C.c_void_p(C.addressof(fake))
    rc = lib.mg_walker_step(tp, ms, prm, 4, st, p, p, p, None, p, None)
    assert rc == -1004, rc                                       # MG_ERR_UNSUPPORTED
    assert b"mapping = lane" in lib.mg_last_error()


def test_walker_wave_mapping_rejects_a_body_listed_before_its_parent():
    """The kinematics scans (and the lane-parallel topology tables) walk body_parent[] upwards and rely on parents
    coming first, like the reference's MJCF / URDF traversal orders produce them: a topology that breaks the rule is
    refused by the host-side validation (no GPU needed), not run into an endless walk on the device."""
    import ctypes as C
    from metagym_amd import _lib
    lib = _lib.load()
    tp = _lib.WalkerTopology()
    tp.n_bodies, tp.n_joints, tp.n_spheres, tp.n_feet = 3, 2, 0, 0
    tp.body_parent[0], tp.body_parent[1], tp.body_parent[2] = -1, 2, 0          # body 1 hangs off body 2
    tp.joint_body[0], tp.joint_body[1] = 1, 2
    ms = _lib.WalkerModels()
    fake = C.create_string_buffer(8)
    ms.table, ms.n_tasks, ms.model_stride = C.addressof(fake), 1, 25 * 3 + 12 * 2
    prm = _lib.WalkerParams()
    prm.time_step, prm.frame_skip, prm.solver_iterations, prm.mapping = 0.005, 4, 5, 1
    st = _lib.WalkerState()
    for name, _ in _lib.WalkerState._fields_:
        setattr(st, name, C.addressof(fake))
    p = C.c_void_p(C.addressof(fake))
    rc = lib.mg_walker_step(tp, ms, prm, 4, st, p, p, p, None, p, None)
    assert rc == -1003, rc                                       # MG_ERR_BAD_CONFIG
    assert b"parents come first" in lib.mg_last_error()


def test_public_header_is_plain_c99(tmp_path):
    """The boundary is a C ABI: include/metagym_hip.h must compile as C99 (no C++-isms, no HIP or torch types)
    and a C program must be able to link the shared library's symbols."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "use_header.c"
    src.write_text('#include "metagym_hip.h"\n'
                   'int probe(void) { mg_quadrotor_config c; mg_maze_tasks t; mg_walker_params p;\n'
                   '  (void)t; (void)p; return mg_quadrotor_default_config(&c) + mg_abi_version(); }\n')
    out = subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"),
                          "-c", str(src), "-o", str(tmp_path / "use_header.o")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr


def test_new_round2_entry_points_validate_before_they_launch():
    """mg_a1_observation_extras and the walker's terrain parameters: wrong arguments are error codes with a message, decided
    on the host before any device call (so this runs without a GPU)."""
    import ctypes as C
    from metagym_amd import _lib
    lib = _lib.load()
    fake = C.create_string_buffer(64)
    p = C.c_void_p(C.addressof(fake))
    assert lib.mg_a1_observation_extras(0, 1, 0, 20, p, p, p, None, p, None) == -1002                  # n_envs
    assert lib.mg_a1_observation_extras(4, 0, 0, 20, p, p, p, None, p, None) == -1003 and b"flags" in lib.mg_last_error()
    assert lib.mg_a1_observation_extras(4, 8, 0, 20, p, p, p, None, p, None) == -1003                  # unknown flag bit
    assert lib.mg_a1_observation_extras(4, _lib.A1_EXTRA_ETG_OBS, 0, 99, p, p, p, None, p, None) == -1002 and b"etg_h" in lib.mg_last_error()
    assert lib.mg_a1_observation_extras(4, _lib.A1_EXTRA_ETG, 0, 20, None, p, p, None, p, None) == -1001 and b"etg_act" in lib.mg_last_error()
    assert lib.mg_a1_observation_extras(4, _lib.A1_EXTRA_YAW, 0, 20, None, None, None, None, p, None) == -1001 and b"pose" in lib.mg_last_error()
    # terrain boxes: a negative count, a NULL table, the lane mapping
    tp = _lib.WalkerTopology()
    tp.n_bodies, tp.n_joints, tp.n_spheres, tp.n_feet = 2, 1, 1, 0
    tp.body_parent[0], tp.body_parent[1] = -1, 0
    # ABI 4's sphere_foot is validated: the zero-initialised table of a caller that predates it would report every proxy to
    # foot 0 (here: a robot without feet), and a foot whose body does not carry the proxy is a mix-up
    empty = (_lib.WalkerModels(), _lib.WalkerParams(), 4, _lib.WalkerState(), p, p, p, None, p, None)
    assert lib.mg_walker_step(tp, *empty) == -1003 and b"sphere_foot" in lib.mg_last_error()
    tp.n_feet, tp.foot_body[0], tp.sphere_body[0], tp.sphere_foot[0] = 1, 1, 0, 0
    assert lib.mg_walker_step(tp, *empty) == -1003 and b"reports to foot" in lib.mg_last_error()
    tp.n_feet, tp.sphere_foot[0] = 0, -1
    ms = _lib.WalkerModels()
    ms.table, ms.n_tasks, ms.model_stride = C.addressof(fake), 1, 25 * 2 + 12 + 4
    prm = _lib.WalkerParams()
    prm.time_step, prm.frame_skip, prm.solver_iterations, prm.mapping = 0.005, 4, 5, 1
    st = _lib.WalkerState()
    for name, _ in _lib.WalkerState._fields_:
        setattr(st, name, C.addressof(fake))
    prm.n_terrain_boxes = -1
    assert lib.mg_walker_step(tp, ms, prm, 4, st, p, p, p, None, p, None) == -1002 and b"terrain" in lib.mg_last_error()
    prm.n_terrain_boxes, prm.terrain = 3, None
    assert lib.mg_walker_step(tp, ms, prm, 4, st, p, p, p, None, p, None) == -1001
    prm.terrain, prm.mapping = C.addressof(fake), 0
    assert lib.mg_walker_step(tp, ms, prm, 4, st, p, p, p, None, p, None) == -1004 and b"wave mapping" in lib.mg_last_error()


def test_uniform_cell_size_promise_is_checked_not_trusted():
    """include/metagym_hip.h mg_maze_view.uniform_cell_size: the caller's "every task has this cell size" is
    compared with the table's cell_size column (TaskConfig.cell_size, maze_task.py:15-17). A wrong value is MG_ERR_BAD_CONFIG with
    the offending task named — never wrong pixels. Host table here (no GPU needed); the device route is in test_maze_gpu.py."""
    import numpy as np
    from metagym_amd import _lib
    lib = _lib.load()
    sc = np.zeros((5, 8))
    sc[:, 0] = 2.0
    t = _lib.MazeTasks()
    t.n, t.n_tasks, t.scalars = 9, 5, sc.ctypes.data
    assert lib.mg_maze_check_uniform_cell_size(t, 2.0, None) == 0
    assert lib.mg_maze_check_uniform_cell_size(t, 1.0, None) == -1003 and b"task 0 of 5 has cell_size 2" in lib.mg_last_error()
    sc[3, 0] = 2.0000000000000004                                         # one ulp off in one task
    assert lib.mg_maze_check_uniform_cell_size(t, 2.0, None) == -1003 and b"task 3" in lib.mg_last_error()
    assert lib.mg_maze_check_uniform_cell_size(t, 0.0, None) == -1003     # 0 means "unknown": nothing to check
    assert lib.mg_maze_check_uniform_cell_size(t, float("nan"), None) == -1003
    assert lib.mg_maze_check_uniform_cell_size(None, 2.0, None) == -1001
    t.scalars = None
    assert lib.mg_maze_check_uniform_cell_size(t, 2.0, None) == -1001
    t.scalars, t.n_tasks = sc.ctypes.data, 0
    assert lib.mg_maze_check_uniform_cell_size(t, 2.0, None) == -1002
    # (ABI 7) mg_maze_forget_tasks: host-only, MG_OK for a known, an unknown and a scalar-less table; NULL is an error code
    t.n_tasks = 5
    sc[3, 0] = 2.0
    assert lib.mg_maze_check_uniform_cell_size(t, 2.0, None) == 0
    assert lib.mg_maze_forget_tasks(t) == 0 and lib.mg_maze_forget_tasks(t) == 0
    t.scalars = None
    assert lib.mg_maze_forget_tasks(t) == 0
    assert lib.mg_maze_forget_tasks(None) == -1001
