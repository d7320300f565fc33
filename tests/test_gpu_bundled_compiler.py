"""The scene-specialised kernels under the OTHER compiler: conftest.py makes every test of this suite compile them with the image's
libamd_comgr (as bench.py does); a host that simply imports torch first gets the older one PyTorch bundles.  One child process does that and checks
the property the specialised kernels stand on - the same image as the generic kernel, bit for bit, and the same ray counts - for 02, 03 and 04."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
def test_specialised_kernels_match_generic_under_the_bundled_compiler(ctx):
    env = dict(os.environ, RT_MI355_SYSTEM_COMGR="0")
    env.pop("RT_MI355_JIT", None)
    r = subprocess.run([sys.executable, os.path.join(HERE, "bundled_compiler_child.py")], capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert "PyTorch" in out["compiler"], out["compiler"]
    for name, c in out["cases"].items():
        assert c["kernels"] == ([1, 2] if name == "02" else [2, 1]), (name, c)               # 03/04: specialised, then generic; 02: generic, then specialised
        assert c["rays"][0] == c["rays"][1] and c["rgb_equal"], (name, c)
        assert c["mean"] > 0.01


@pytest.mark.gpu
def test_jit_cache_on_disk(ctx, tmp_path):
    """RT_MI355_JIT_CACHE=<dir> (rt_jit.hip): the first process compiles the scene's kernel and leaves its code object there, the second loads it
    instead of compiling - the same frame bit for bit, both through the specialised kernel, the second start-up well under the compile time."""
    env = dict(os.environ, RT_MI355_JIT_CACHE=str(tmp_path))
    env.pop("RT_MI355_JIT", None)
    runs = []
    for _ in range(2):
        r = subprocess.run([sys.executable, os.path.join(HERE, "jit_cache_child.py")], capture_output=True, text=True, timeout=280, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        runs.append(json.loads(r.stdout.strip().splitlines()[-1]))
        files = [f for f in os.listdir(tmp_path) if f.endswith(".co")]
        assert len(files) == 1, files                                  # written once, reused
    assert runs[0]["kernel"] == runs[1]["kernel"] == 2
    assert runs[0]["sha"] == runs[1]["sha"] and runs[0]["rays"] == runs[1]["rays"]
    assert runs[1]["first_render_s"] < 0.5 * runs[0]["first_render_s"], runs


@pytest.mark.gpu
def test_jit_table_replaces_stale_kernels(ctx):
    """The table of compiled scenes (rt_jit.hip: 256 per process, here 2) replaces its least recently used kernel when that one is stale, and leaves newcomers
    on the generic kernel while every resident kernel is in use - a long-lived process keeps specialising new scenes, a cycle through too many does not
    recompile per render.  tests/jit_lru_child.py renders three scenes in a fixed order and reports the kernel kind of every render."""
    env = dict(os.environ, RT_MI355_JIT_MAX_SCENES="2")
    env.pop("RT_MI355_JIT", None); env.pop("RT_MI355_JIT_CACHE", None)
    r = subprocess.run([sys.executable, os.path.join(HERE, "jit_lru_child.py")], capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    kinds = json.loads(r.stdout.strip().splitlines()[-1])["kinds"]
    assert kinds == [2, 2, 1] + [2] * 8 + [2, 1, 2, 2], kinds
