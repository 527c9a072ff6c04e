"""The fixtures under tests/golden/ are outputs of the reference itself: where the reference is mounted (the build container; never
the GPU box) the base family is generated again into a scratch directory by tools/gen_golden.py and must come out byte for byte as
committed.  Keeps the generator runnable from a fresh clone (it writes nothing outside its output and temporary directories)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BASE_FILES = ("adversarial", "cfg1_500", "cfg2_small", "eval_tri", "merge_dedup", "metacell", "metacell_inputs", "simulated_elastic",
              "simulated_st", "synthetic_example", "unpack_merge")


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="the reference is not mounted here")
def test_base_fixtures_regenerate_byte_for_byte(tmp_path):
    out = tmp_path / "golden"
    env = dict(os.environ, SAME_GOLDEN_OUT=str(out))
    done = subprocess.run([sys.executable, "-B", os.path.join(ROOT, "tools", "gen_golden.py")], cwd=str(tmp_path), env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert done.returncode == 0, done.stdout[-2000:]
    assert sorted(f[:-4] for f in os.listdir(out)) == sorted(BASE_FILES)
    for name in BASE_FILES:
        with open(out / f"{name}.npz", "rb") as a, open(os.path.join(ROOT, "tests", "golden", f"{name}.npz"), "rb") as b:
            assert a.read() == b.read(), f"{name}.npz differs from the committed fixture"


def test_generator_keeps_its_scratch_out_of_the_tree():
    """gpurun_out/ is git-ignored: a fresh clone does not have it, so the generator must not need it (round 3: FileNotFoundError)."""
    src = open(os.path.join(ROOT, "tools", "gen_golden.py"), encoding="utf-8").read()
    assert "gpurun_out" not in src and "tempfile.mkdtemp(prefix=" in src
