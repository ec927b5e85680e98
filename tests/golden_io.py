"""Loader for tests/golden/*.json (written by oracle/gen_golden.js)."""
import json
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_TAGS = {"__inf": float("inf"), "__-inf": float("-inf"), "__nan": float("nan"), "__-0": -0.0}


def _untag(o):
    if isinstance(o, str):
        return _TAGS.get(o, o)
    if isinstance(o, list):
        return [_untag(v) for v in o]
    if isinstance(o, dict):
        return {k: _untag(v) for k, v in o.items()}
    return o


def load(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        return _untag(json.load(f))
