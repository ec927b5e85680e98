"""TEST INFRASTRUCTURE: translate the closures of tests/js/user_models.js with the product's translator (node), build
the generated HIP text for the HOST (tests/host/user_eval_host.cpp, g++ -ffp-contract=off) and evaluate it."""
import ctypes as C
import json
import os
import shutil
import struct
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODE = shutil.which("node")
_dir = None
_cache = {}


def workdir():
    global _dir
    if _dir is None:
        _dir = tempfile.mkdtemp(prefix="amwg_user_")
        p = subprocess.run([NODE, os.path.join(ROOT, "tests", "js", "translate_cli.js"), _dir], cwd=ROOT, capture_output=True, text=True, timeout=120)
        assert p.returncode == 0, p.stdout + "\n" + p.stderr
    return _dir


_stepper_names = None


def stepper_models():
    """Translates the zero-argument log_post closures of tests/js/stepper_cases.js (stand-alone steppers) into the same
    work directory; -> their names (stepper_<case>_<k>)."""
    global _stepper_names
    if _stepper_names is None:
        d = workdir()
        p = subprocess.run([NODE, os.path.join(ROOT, "tests", "js", "translate_steppers_cli.js"), d], cwd=ROOT, capture_output=True, text=True, timeout=120)
        assert p.returncode == 0, p.stdout + "\n" + p.stderr
        _stepper_names = json.load(open(os.path.join(d, "steppers.index.json")))
    return _stepper_names


def fuzz_models(seed, count, n_derived=48):
    """Random closures (tests/js/fuzz_translate_cli.js) translated into the work directory; -> their names."""
    d = workdir()
    p = subprocess.run([NODE, os.path.join(ROOT, "tests", "js", "fuzz_translate_cli.js"), d, str(seed), str(count), str(n_derived)], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + "\n" + p.stderr
    return p.stdout.split()


def stepper_states(name):
    return json.load(open(os.path.join(workdir(), name + ".states.json")))


def translate_extra(name):
    """Translates one closure that is not in user_models.names (the full-size bench_* closures)."""
    d = workdir()
    if not os.path.exists(os.path.join(d, name + ".hip")):
        p = subprocess.run([NODE, os.path.join(ROOT, "tests", "js", "translate_cli.js"), d, name], cwd=ROOT, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stdout + "\n" + p.stderr
    return d


def translated(name):
    """-> (source, arrays, meta) without building the host harness"""
    d = translate_extra(name)
    return (open(os.path.join(d, name + ".hip")).read(), read_arrays(os.path.join(d, name + ".arrays.bin")),
            json.load(open(os.path.join(d, name + ".meta.json"))))


def user_spec_part(source, arrays, meta):
    """The `user` entry of an amwg_ctypes.Sampler spec."""
    return {"source": source, "arrays": arrays, "array_types": meta["array_types"], "n_derived": len(meta["derived"]),
            "lds_bytes": meta["lds_bytes"], "lds_bytes_one_lane": meta.get("lds_bytes_one_lane", meta["lds_bytes"]), "parallel": meta["parallel"], "max_threads": meta["max_threads"],
            "work_per_eval": meta.get("work_per_eval", 0.0), "work_one_lane": meta.get("work_one_lane", 0.0),
            "rows_n_obs": meta.get("rows_n_obs", 0), "rows_groups": meta.get("rows_groups", 0), "rows_sweep": meta.get("rows_sweep", 0)}


def read_arrays(path):
    buf = open(path, "rb").read()
    n, = struct.unpack_from("<I", buf, 0)
    o, out = 4, []
    for _ in range(n):
        ln, = struct.unpack_from("<Q", buf, o)
        o += 8
        out.append(np.frombuffer(buf, dtype="<f8", count=ln, offset=o).copy())
        o += ln * 8
    return out


class HostModel:
    """The generated amwg::UserModel compiled for the host."""

    def __init__(self, name):
        d = workdir()
        self.name = name
        self.source = open(os.path.join(d, name + ".hip")).read()
        self.meta = json.load(open(os.path.join(d, name + ".meta.json")))
        self.arrays = read_arrays(os.path.join(d, name + ".arrays.bin"))
        so = os.path.join(d, name + ".so")
        cmd = ["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared",
               "-I", os.path.join(ROOT, "bayes.js_amd", "csrc"), '-DAMWG_USER_SOURCE="%s"' % os.path.join(d, name + ".hip"),
               "-o", so, os.path.join(ROOT, "tests", "host", "user_eval_host.cpp")]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-4000:]
        self.lib = C.CDLL(so)
        self.lib.user_eval.restype = C.c_double
        self.lib.user_eval.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_void_p), C.c_int, C.c_int, C.POINTER(C.c_double)]
        # the storage types the translator chose for the device (f64 / u8 / i32): the host build reads the same types
        self.typed = [a.astype([np.float64, np.uint8, np.int32][t]) for a, t in zip(self.arrays, self.meta["array_types"])]
        self.ptrs = (C.c_void_p * max(1, len(self.typed)))(*[a.ctypes.data for a in self.typed])
        self.D = self.lib.user_num_derived()

    def eval(self, state, lanes=1, derived=False):
        st = np.ascontiguousarray(state, dtype=np.float64)
        dv = np.zeros(max(1, self.D))
        v = self.lib.user_eval(st.ctypes.data_as(C.POINTER(C.c_double)), self.ptrs, len(self.arrays), lanes,
                               dv.ctypes.data_as(C.POINTER(C.c_double)) if derived else None)
        return (v, dv[: self.D].tolist()) if derived else v


def host_model(name):
    if name not in _cache:
        _cache[name] = HostModel(name)
    return _cache[name]
