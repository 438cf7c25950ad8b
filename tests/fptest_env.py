"""The library's test knobs travel in ONE environment variable, FP_TEST="key=value,key=value" (fp_internal.h fp_test_opt;
the keys and the tests that use them: INTEGRATION.md).  Helpers to compose and to read it."""
import os


def with_test_opts(env=None, **opts):
    """a copy of `env` (default: os.environ) whose FP_TEST carries `opts` on top of what it already holds"""
    out = dict(os.environ if env is None else env)
    cur = dict(kv.split("=", 1) for kv in out.get("FP_TEST", "").split(",") if "=" in kv)
    cur.update({k: str(v) for k, v in opts.items() if v not in (None, "")})
    if cur:
        out["FP_TEST"] = ",".join(f"{k}={v}" for k, v in cur.items())
    return out


def test_opt(key, default=None):
    for kv in os.environ.get("FP_TEST", "").split(","):
        if kv.startswith(key + "="):
            return kv.split("=", 1)[1]
    return default
