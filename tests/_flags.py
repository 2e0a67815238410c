"""The library's test switches live in ONE environment variable, DEXGPU_TEST (dextractor_amd/csrc/dx_env.h): keys and key=value
pairs, comma separated.  set_flag / test_env keep that string for the tests."""
import os


def _parse(s):
    d = {}
    for tok in (s or "").replace(" ", ",").split(","):
        if tok:
            k, _, v = tok.partition("=")
            d[k] = v
    return d


def _join(d):
    return ",".join(k if v == "" else f"{k}={v}" for k, v in d.items())


def set_flag(monkeypatch, key, value="1"):
    """key[=value] into DEXGPU_TEST for the rest of the test (value None: the key taken out again)."""
    d = _parse(os.environ.get("DEXGPU_TEST"))
    if value is None:
        d.pop(key, None)
    else:
        d[key] = str(value)
    if d:
        monkeypatch.setenv("DEXGPU_TEST", _join(d))
    else:
        monkeypatch.delenv("DEXGPU_TEST", raising=False)


def test_env(**kv):
    """The DEXGPU_TEST string of these keys on top of what the environment holds (for a child process's env)."""
    d = _parse(os.environ.get("DEXGPU_TEST"))
    d.update({k: str(v) for k, v in kv.items()})
    return _join(d)


test_env.__test__ = False          # (not a test, whatever its name says to pytest)
