"""The planning half of same_dev_alloc_spread (same_amd/csrc/spread_plan.h: which labelled chunks, in which order) is plain
C++; compile its test with g++ and run it on the CPU."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_spread_plan_native(tmp_path):
    exe = tmp_path / "spread_plan_test"
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-Wall", "-Werror", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I", os.path.join(ROOT, "same_amd", "csrc"),
                    os.path.join(ROOT, "tests", "native", "spread_plan_test.cpp"), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr
