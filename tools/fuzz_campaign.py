"""Translator fuzz campaign (no GPU): random closures from tests/js/fuzz_translate_cli.js, the generated text compiled for the host,
every derived quantity and the return value compared bit for bit with what V8 returned at 40 random states, plus the lane-split
orders (2, 4, 64 lanes: derived quantities identical, sums equal to rounding).

    python tools/fuzz_campaign.py FIRST_SEED LAST_SEED [MODELS_PER_SEED]

tests/test_translate.py runs two fixed seeds of the same check; this is the open-ended version (about 15 s per model).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "bayes.js_amd"))
import numpy as np  # noqa: E402
import user_host  # noqa: E402


def f64(h):
    return float(np.frombuffer(bytes.fromhex(h), dtype=">f8")[0])


def same(a, b):
    return (a != a and b != b) or np.float64(a).tobytes() == np.float64(b).tobytes()


def check(seed, count):
    bad = 0
    for name in user_host.fuzz_models(seed, count):
        m = user_host.host_model(name)
        pts = user_host.stepper_states(name)
        src = open(os.path.join(user_host.workdir(), name + ".js")).read().split("\n")
        for t, pt in enumerate(pts):
            st = [f64(h) for h in pt["state"]]
            got, dv = m.eval(st, 1, derived=True)
            want = [f64(h) for h in pt["derived"]] + [f64(pt["lp"])]
            for j, (a, b) in enumerate(zip(dv + [got], want)):
                if not same(a, b):
                    key = m.meta["derived"][j] if j < len(m.meta["derived"]) else "return"
                    line = [ln for ln in src if ("s." + key + " =") in ln][:1]
                    print("MISMATCH", name, key, "got", a, "want", b, "state", st, "\n   ", (line[0][:400] if line else ""))
                    bad += 1
                    break
            if m.meta["parallel"] and t < 20:
                for lanes in (2, 4, 64):
                    v, dvl = m.eval(st, lanes, derived=True)
                    okd = all(same(a, b) for a, b in zip(dvl, dv))
                    okv = same(v, got) or (np.isfinite(got) and abs(v - got) <= 1e-9 * max(1.0, abs(got)))
                    if not (okd and okv):
                        print("LANE MISMATCH", name, lanes, v, got, okd, st)
                        bad += 1
                        break
        print(name, "checked", len(pts), "states,", len(m.meta["derived"]), "derived, parallel", m.meta["parallel"])
    return bad


if __name__ == "__main__":
    first, last = int(sys.argv[1]), int(sys.argv[2])
    per = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    total = 0
    for seed in range(first, last + 1):
        try:
            total += check(seed, per)
        except AssertionError as e:      # the generator or the translator refused a program: report and go on
            print("seed", seed, "ERROR", str(e)[-600:])
            total += 1
    print("mismatches:", total)
    sys.exit(1 if total else 0)
