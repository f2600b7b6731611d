"""Build libasvgp_hip.so (gfx950) in-tree with hipcc.  `python -m asvgp_amd.build` or build.build().

Every source is compiled to an object in parallel; elbo.hip (the template-heavy band chains) is compiled once per
bandwidth and launcher (-DASVGP_ELBO_ONLY_K=k -DASVGP_ELBO_PART=p hold the instantiations) plus once for its C entry points, then everything is linked."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libasvgp_hip.so")
SOURCES = ["phi_pass.hip", "band_ops.hip", "elbo.hip", "kron.hip", "additive.hip", "handle.hip", "prior_dd.hip"]
EXTRA = {"prior_dd.hip": ["-ffp-contract=off"]}   # error-free transforms: nothing may be fused or re-associated
HOST_SOURCES = ["prior_plan.cpp"]   # plain C++ (host planner of the prior chain): no FMA contraction, see prior_plan.cpp
ELBO_KS = (1, 2, 3, 4, 5, 6)
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-munsafe-fp-atomics"]


def _deps():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "asvgp_hip.h")]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _units():
    """(object file, source, extra flags, dependency list)"""
    hdrs = [d for d in _deps() if d.endswith((".hpp", ".h"))]
    units = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        units.append((os.path.join(OBJ, s.replace(".hip", ".o")), src, EXTRA.get(s, []), hdrs + [src]))
    for s in HOST_SOURCES:
        src = os.path.join(CSRC, s)
        units.append((os.path.join(OBJ, s.replace(".cpp", ".o")), src, ["-ffp-contract=off", "-x", "c++"], hdrs + [src]))
    src = os.path.join(CSRC, "elbo.hip")
    for k in ELBO_KS:   # longest first; part 1 = ELBO + gradient (tangent chains), part 2 = posterior
        for part in (2, 1):
            units.insert(0, (os.path.join(OBJ, "elbo_k%d_p%d.o" % (k, part)), src,
                             ["-DASVGP_ELBO_ONLY_K=%d" % k, "-DASVGP_ELBO_PART=%d" % part], hdrs + [src]))
    return units


def build(force=False, verbose=True, jobs=None):
    if not force and not _stale(LIB, _deps()):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    units = _units()
    todo = [u for u in units if force or _stale(u[0], u[3])]

    def compile_one(u):
        obj, src, extra, _ = u
        flags = [f for f in FLAGS if not (src.endswith(".cpp") and f.startswith(("--offload-arch", "-munsafe")))]
        cmd = [hipcc] + flags + extra + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=CSRC)

    jobs = jobs or int(os.environ.get("ASVGP_BUILD_JOBS", str(min(8, os.cpu_count() or 1))))
    with ThreadPoolExecutor(max_workers=max(1, jobs)) as pool:
        list(pool.map(compile_one, todo))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + [u[0] for u in units]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
