"""Build recipe for libmhe_hip.so (gfx950 only, in-tree so that it travels with
the snapshot to the GPU box).  `python -m mhentropy_amd.build` or
__graft_entry__.build()."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libmhe_hip.so")
SOURCES = ["api.hip", "rng.hip", "mano.hip", "mano_skin.hip", "mano_bwd.hip", "flow.hip", "flow_bf16.hip", "flow_ns.hip", "conv.hip", "conv_p8.hip", "conv_stream.hip", "conv_tail.hip", "conv_fuse.hip", "conv_gram.hip", "conv_fold.hip", "conv_halo.hip", "stem_pool.hip", "conv_wide.hip", "wgrad.hip", "flow_bwd.hip", "flow_rev.hip", "flow_fwd.hip", "trunk_bwd.hip", "glow.hip", "glow_affine.hip", "glow_fwd.hip", "glow_rev.hip", "metrics.hip", "body.hip", "lbs_skin.hip", "ho3d.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"] + os.environ.get("MHE_EXTRA_FLAGS", "").split()


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=True):
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "mhe.h"))
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        if not os.path.exists(s):
            continue
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [HIPCC] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    if force or procs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv)
    print(LIB)
