"""oracle/make_ref.py — build `oracle/_ref/`: the UNMODIFIED reference Quadrotor (and MetaMaze) packages, byte-compiled from the
sources where they lie under /root/reference, so that the bench box (which has no /root/reference) can time "the reference CPU
path on the same box's host cores in the same run" (BASELINE.json north_star).

TEST / MEASUREMENT INFRASTRUCTURE — not the product. Nothing under metagym_amd/ may import or read oracle/_ref
(tests/test_abi.py::test_product_never_imports_the_oracle); only bench.py's `cpu_baseline` leg does.

What it writes (all under oracle/_ref/, which is git-ignored — no reference source enters the history — but NOT gpurun-ignored,
so it travels to the GPU box like the built .so files):

  metagym/__init__.pyc, metagym/quadrotor/{__init__,env,quadrotorsim}.pyc      sourceless CPython byte-code (py_compile of
  metagym/metamaze/__init__.pyc, metagym/metamaze/envs/*.pyc                    the reference .py, nothing edited)
  metagym/quadrotor/config.json                                                 the simulator constants the reference reads at
                                                                                run time (env.py:59-61) — parsed and re-serialised
  metagym/metamaze/envs/img/*.png                                               the nine 64x64 textures maze_task.py:19-36 loads
                                                                                when the module is imported — decoded and re-encoded
  MANIFEST.json                                                                 sha256 of every source it was built from, the
                                                                                interpreter's byte-code magic, numpy version

`metagym/quadrotor/render.py` is deliberately left out: its import fails in the reference itself without a display
(env.py:23-27 → NO_DISPLAY), which is the configuration every timing here uses.

Run:  python oracle/make_ref.py            (called by __graft_entry__.build() when /root/reference exists)
"""
import hashlib
import importlib.util
import json
import os
import py_compile
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")

# (reference-relative source, kind)
MODULES = [
    "metagym/__init__.py",
    "metagym/quadrotor/__init__.py",
    "metagym/quadrotor/env.py",            # Quadrotor.step  env.py:127-165
    "metagym/quadrotor/quadrotorsim.py",   # QuadrotorSim.step / _run_internal  quadrotorsim.py:122-304
    "metagym/metamaze/__init__.py",
    "metagym/metamaze/envs/__init__.py",
    "metagym/metamaze/envs/dynamics.py",
    "metagym/metamaze/envs/maze_2d.py",
    "metagym/metamaze/envs/maze_base.py",
    "metagym/metamaze/envs/maze_continuous_3d.py",
    "metagym/metamaze/envs/maze_discrete_3d.py",
    "metagym/metamaze/envs/maze_env.py",
    "metagym/metamaze/envs/maze_task.py",
    "metagym/metamaze/envs/ray_caster_utils.py",
]
DATA = ["metagym/quadrotor/config.json"] + ["metagym/metamaze/envs/img/" + f for f in (
    # MazeTaskManager.__init__ (maze_task.py:19-36) lists this directory at import time: the module does not import without it
    "arrow.png", "ceil_wood.png", "ground_greystone.png", "wall_aoi_stone.png", "wall_bluestone.png", "wall_colorstone.png",
    "wall_mossy.png", "wall_red_brick_1.png", "wall_red_brick_2.png")]


def _sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def build(reference="/root/reference", out=OUT, quiet=False):
    """Byte-compile the reference modules into `out`. Returns the manifest dict; raises FileNotFoundError when the reference
    tree is absent (the GPU box: there the prebuilt oracle/_ref that travelled with the snapshot is used as it is)."""
    if not os.path.isdir(os.path.join(reference, "metagym", "quadrotor")):
        raise FileNotFoundError("no reference tree at %s" % reference)
    if os.path.isdir(out):
        shutil.rmtree(out)
    manifest = {"reference": reference, "python": sys.version.split()[0],
                "bytecode_magic": importlib.util.MAGIC_NUMBER.hex(), "files": {}}
    for rel in MODULES:
        src = os.path.join(reference, rel)
        dst = os.path.join(out, rel + "c")                 # foo.py -> foo.pyc beside where foo.py would be: a sourceless import
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        # dfile = what tracebacks show; UNCHECKED_HASH: the .pyc is valid without its source file being present
        py_compile.compile(src, cfile=dst, dfile="<reference>/" + rel, doraise=True,
                           invalidation_mode=py_compile.PycInvalidationMode.UNCHECKED_HASH)
        manifest["files"][rel] = {"sha256_source": _sha(src), "built": os.path.relpath(dst, out)}
    for rel in DATA:
        src, dst = os.path.join(reference, rel), os.path.join(out, rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        # data is BUILT too, not copied byte for byte: the JSON is parsed and re-serialised (same values), the textures are decoded
        # and re-encoded (same pixels, checked) — what the modules read is identical, the files are this recipe's output
        if rel.endswith(".json"):
            with open(src) as f:
                obj = json.load(f)
            with open(dst, "w") as f:
                json.dump(obj, f, sort_keys=True, separators=(",", ":"))
            assert json.load(open(dst)) == obj
        elif rel.endswith(".png"):
            from PIL import Image
            import numpy
            im = Image.open(src)
            im.load()
            im.save(dst, format="PNG", optimize=True)
            assert numpy.array_equal(numpy.asarray(Image.open(dst).convert("RGB")), numpy.asarray(im.convert("RGB"))), rel
        else:
            shutil.copyfile(src, dst)
        manifest["files"][rel] = {"sha256_source": _sha(src), "built": rel}
    try:
        import numpy
        manifest["numpy_at_build"] = numpy.__version__
    except ImportError:
        pass
    with open(os.path.join(out, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    if not quiet:
        print("oracle/_ref: %d modules byte-compiled + %d data files from %s" % (len(MODULES), len(DATA), reference))
    return manifest


def usable(out=OUT):
    """True when `out` holds a build this interpreter can import (same byte-code magic)."""
    try:
        m = json.load(open(os.path.join(out, "MANIFEST.json")))
    except (OSError, ValueError):
        return False
    return m.get("bytecode_magic") == importlib.util.MAGIC_NUMBER.hex() and \
        os.path.exists(os.path.join(out, "metagym", "quadrotor", "env.pyc"))


if __name__ == "__main__":
    build(*(sys.argv[1:2] or ["/root/reference"]))
