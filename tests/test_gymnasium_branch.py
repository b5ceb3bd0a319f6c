"""The branch of rsoccer_amd/gymshim.py that runs when the real ``gymnasium`` package is installed (it is not in this image): a package
with gymnasium's public shape (tests/gymnasium_stub.py) is put on the path of a child process, and the registry / make() / wrapper /
seeding behaviour the reference relies on (rsoccer_gym/__init__.py:1-30, README.md:116-133) is checked THROUGH gymnasium's own entry
points.  CPU only: the simulator behind the adapters is the oracle-backed stand-in of tests/fake_robosim.py."""
import json
import os
import subprocess
import sys
import textwrap

import gymnasium_stub

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

CHILD = r'''
import json, os, sys
import numpy as np
import gymnasium
assert gymnasium.__version__.endswith("+stub")
import rsoccer_amd
from rsoccer_amd import gymshim
import fake_robosim

# the shim stepped aside: gymnasium's own objects are what the package exposes
assert gymshim.HAVE_GYMNASIUM
assert gymshim.Env is gymnasium.Env and gymshim.spaces is gymnasium.spaces
assert rsoccer_amd.registry is gymnasium.registry and rsoccer_amd.make is gymnasium.make
assert gymshim.TimeLimit is gymnasium.wrappers.TimeLimit

# the five ids live in ITS registry with the reference's step limits / kwargs / class names
ref = json.load(open(os.path.join(sys.argv[1], "golden", "registry.json")))
assert set(gymnasium.registry) == set(ref), sorted(gymnasium.registry)
for env_id, want in ref.items():
    spec = gymnasium.spec(env_id)
    assert spec.max_episode_steps == want["max_episode_steps"], env_id
    assert spec.kwargs == want["kwargs"], env_id
    assert spec.entry_point.split(":")[1] == want["entry_point"].split(":")[1], env_id

out = {}
for env_id, od, ad in (("VSS-v0", 40, 2), ("SSLStaticDefenders-v0", 24, 5)):
    fake_robosim.arm()
    env = gymnasium.make(env_id, sim_backend=fake_robosim)
    assert isinstance(env, gymnasium.wrappers.TimeLimit)                 # README.md:116-133: make() hands out the wrapped env
    assert isinstance(env.unwrapped, gymnasium.Env) and not isinstance(env.unwrapped, gymnasium.Wrapper)
    assert env.unwrapped.spec.id == env_id and env.spec.max_episode_steps == ref[env_id]["max_episode_steps"]
    assert isinstance(env.action_space, gymnasium.spaces.Box) and env.action_space.shape == (ad,)
    assert isinstance(env.observation_space, gymnasium.spaces.Box) and env.observation_space.shape == (od,)
    try:
        env.step(env.action_space.sample())
        raise SystemExit("step() before reset() passed gymnasium's order enforcing")
    except RuntimeError:
        pass
    obs, info = env.reset(seed=11)
    assert obs.shape == (od,) and obs.dtype == np.float32 and info == {}
    assert env.unwrapped._np_random_seed == 11                           # Env.reset(seed=) reached gymnasium's base class
    first = env.np_random.integers(1 << 30)
    env.reset(seed=11)
    assert env.np_random.integers(1 << 30) == first                      # ... and seeds np_random reproducibly
    n, term, trunc = 0, False, False
    limit = ref[env_id]["max_episode_steps"]
    while not (term or trunc):
        res = env.step(np.zeros(ad, dtype=np.float32))
        assert isinstance(res, tuple) and len(res) == 5
        obs, rew, term, trunc, info = res
        assert obs.shape == (od,) and np.isscalar(rew) or np.ndim(rew) == 0
        assert isinstance(info, dict)
        n += 1
        assert n <= limit
    assert term or n == limit                                            # the TimeLimit wrapper cut the episode at the registered length
    obs2, _ = env.reset()                                                # reset after the end works, the wrapper's clock restarts
    assert env._elapsed_steps == 0
    env.close()
    out[env_id] = [n, bool(term), bool(trunc)]

# a vector env of the fused layer describes its batch (gymnasium.vector convention): checked on the class attributes, no GPU needed
from rsoccer_amd.vec.fused import batched_space
b = batched_space(gymnasium.spaces.Box(low=-1, high=1, shape=(2,), dtype=np.float32), 7)
assert isinstance(b, gymnasium.spaces.Box) and b.shape == (7, 2) and b.low.min() == -1 and b.high.max() == 1
print("GYMNASIUM_BRANCH_OK", json.dumps(out))
'''


def test_real_gymnasium_branch_registers_makes_wraps_and_seeds(tmp_path, oracle_mod):
    stub_root = gymnasium_stub.write(str(tmp_path / "site"))
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([stub_root, ROOT, HERE] + ([env["PYTHONPATH"]] if env.get("PYTHONPATH") else []))
    r = subprocess.run([sys.executable, "-c", CHILD, HERE], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("GYMNASIUM_BRANCH_OK")]
    assert line, r.stdout[-2000:]
    res = json.loads(line[0].split(" ", 1)[1])
    # zero actions: the VSS episode runs into the registered TimeLimit (1200 steps) unless a goal ends it earlier
    assert res["VSS-v0"][0] <= 1200 and (res["VSS-v0"][1] or res["VSS-v0"][0] == 1200)


def test_the_shim_is_what_runs_without_gymnasium():
    """in this image (no gymnasium) the package's own minimal surface is used: the other side of the branch"""
    import rsoccer_amd
    from rsoccer_amd import gymshim
    try:
        import gymnasium  # noqa: F401
        have = True
    except Exception:
        have = False
    assert gymshim.HAVE_GYMNASIUM == have
    if not have:
        assert rsoccer_amd.registry is gymshim._REGISTRY and rsoccer_amd.make is gymshim._make
