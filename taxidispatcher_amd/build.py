"""Builds the gfx950 shared library in-tree: taxidispatcher_amd/libtaxidispatcher_amd.so.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the CPU-only container too.
The .so is git-ignored but travels with the repo snapshot to the GPU box.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = [os.path.join(HERE, "csrc", f) for f in ("td_core.hip", "td_assign.hip", "td_lcm.hip", "td_pool.hip", "td_line.hip", "td_tick.hip")]
HDR = [os.path.join(HERE, "csrc", "td_common.h"), os.path.join(HERE, "csrc", "td_forest.h"), os.path.join(ROOT, "include", "taxidispatcher_amd.h")]
LIB = os.environ.get("TD_LIB_OUT") or os.path.join(HERE, "libtaxidispatcher_amd.so")


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.exists(f) and os.path.getmtime(f) > t for f in SRC + HDR)


def build(force=False, verbose=False):
    srcs = [f for f in SRC if os.path.exists(f)]
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    flags = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-fvisibility=hidden",
             "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(HERE, "csrc"),
             "-DTD_BUILDING=1", "-DTD_NT=%s" % os.environ.get("TD_NT", "2"),
             "-DTD_SX_G=%s" % os.environ.get("TD_SX_G", "8"), "-DTD_SX_NG=%s" % os.environ.get("TD_SX_NG", "2"),
             "-DTD_SX_TX=%s" % os.environ.get("TD_SX_TX", "128")]
    # one object per source, compiled side by side (no device symbol crosses a file), then one link; an object is
    # rebuilt only when its source, a header or the flags changed
    objdir = os.path.join(HERE, "csrc", "_obj")
    os.makedirs(objdir, exist_ok=True)
    stamp = " ".join(flags)
    hdr_t = max(os.path.getmtime(f) for f in HDR if os.path.exists(f))
    jobs, objs = [], []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        fl = o + ".flags"
        fresh = (not force and os.path.exists(o) and os.path.exists(fl) and open(fl).read() == stamp
                 and os.path.getmtime(o) > max(os.path.getmtime(s), hdr_t))
        if fresh:
            continue
        cmd = [hipcc] + flags + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        jobs.append((s, o, fl, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    errs = []
    for s, o, fl, pr in jobs:
        out = pr.communicate()[0]
        if pr.returncode != 0:
            errs.append(out)
            if os.path.exists(o):
                os.remove(o)
        else:
            with open(fl, "w") as f:
                f.write(stamp)
    if errs:
        raise RuntimeError("hipcc failed:\n" + "\n".join(errs))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + r.stdout + r.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
