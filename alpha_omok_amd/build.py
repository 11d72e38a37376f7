"""Builds alpha_omok_amd/libomok_hip.so (gfx950) in-tree with hipcc.

    python -m alpha_omok_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting .so travels with the source tree.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INC = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libomok_hip.so")
SOURCES = ["tree_kernels.hip", "step_kernels.hip", "engine.hip", "net.hip", "net_w16.hip", "replay.hip", "rollout.hip"]
# every header under csrc/ (listed from the directory, so a new kernel header can never be missing from the staleness
# check or from source_hash() -- round 4 shipped net_layer_ksplit.hpp outside this list) + the C ABI
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".hpp")) + [os.path.join(INC, "omok_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result", "-I", INC]


def source_hash():
    """sha256 (16 hex digits) over the kernel sources and headers: ties a rocprofv3 counter summary under
    profiles/ to the code it was collected from (bench.py reports `traffic` only when it matches)."""
    import hashlib
    import re
    h = hashlib.sha256()
    for f in sorted(SOURCES + [x for x in HEADERS if not os.path.isabs(x)]):
        with open(os.path.join(CSRC, f), "r", errors="replace") as fh:
            text = fh.read()
        # comments and layout do not change the kernels: hash the code only
        text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
        text = re.sub(r"//[^\n]*", " ", text)
        text = " ".join(text.split())
        h.update(f.encode())
        h.update(text.encode())
    return h.hexdigest()[:16]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, tag=None, extra_flags=None):
    """tag / extra_flags: an experiment build (knock-outs, -DAO_PROF ...) into libomok_hip_<tag>.so, loaded with
    AO_LIB_TAG=<tag>; the product library has no tag."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    tag = tag or os.environ.get("AO_BUILD_TAG")
    lib = LIB if not tag else os.path.join(HERE, "libomok_hip_%s.so" % tag)
    objdir = os.path.join(HERE, "build" if not tag else "build_" + tag)
    extra = (extra_flags if extra_flags is not None else os.environ.get("AO_EXTRA_FLAGS", "")).split()
    os.makedirs(objdir, exist_ok=True)
    hdrs = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc] + FLAGS + extra + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode(errors="replace")))
        if verbose and out:
            print(out.decode(errors="replace"))
    if force or procs or _stale(lib, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-lpthread"]
        subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
