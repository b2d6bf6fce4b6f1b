"""Build libnerfmeshes_hip.so (gfx950) in-tree with hipcc.

    python -m nerfmeshes_amd.build            # incremental
    python -m nerfmeshes_amd.build --force

The shared object is git-ignored but travels to the GPU box with the gpurun snapshot.  hipcc
cross-compiles for gfx950 without a GPU present.
"""
import os
import shutil
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_NAME = "libnerfmeshes_hip.so"
LIB_PATH = os.path.join(CSRC, LIB_NAME)
SOURCES = ["nerf_mlp.hip", "nerf_mlp_generic_a.hip", "nerf_mlp_generic_b.hip", "nerf_mlp_generic_c.hip", "nerf_mlp_generic_d.hip", "nerf_mlp_generic_e.hip", "nerf_mlp_generic_s.hip", "nerf_mlp_generic_a_long.hip", "nerf_mlp_generic_b_long.hip", "nerf_mlp_generic_c_long.hip", "nerf_mlp_generic_d_long.hip", "nerf_mlp_generic_s_long.hip", "nerf_mlp_generic_s_long2.hip", "nerf_train.hip", "nerf_bwd_fused.hip", "nerf_dw.hip", "nerf_dw_g.hip", "nerf_layerwise.hip", "mlp_api.hip", "ray_ops.hip", "marching_cubes.hip", "buff_tree.hip", "np_reduce.hip", "obj_writer.cpp"]
# every header next to the sources (mlp_device*.h, nm_internal.h, mc_luts.h, ...) + the public C ABI
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join("..", "..", "include", "nerfmeshes_hip.h")]
# -ffp-contract=off: the reference computes a*b+c with two roundings (torch eager ops); every fused
# multiply-add in the kernels is an explicit fmaf()/MFMA.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall",
         "-Wno-unused-function"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; cannot build the gfx950 kernels")
    return exe


def _present(files):
    return [f for f in files if os.path.exists(os.path.join(CSRC, f))]


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = _present(SOURCES) + _present(HEADERS) + [os.path.abspath(__file__)]
    return any(os.path.getmtime(os.path.join(CSRC, d) if not os.path.isabs(d) else d) > t for d in deps)


ABLATION_LIB_PATH = os.path.join(CSRC, "libnerfmeshes_hip_ablations.so")


def _includes(path, seen):
    """Quoted includes of `path`, transitively (the headers an object depends on)."""
    try:
        text = open(path, errors="replace").read()
    except OSError:
        return
    for line in text.splitlines():
        line = line.strip()
        if line.startswith("#include \""):
            dep = os.path.normpath(os.path.join(os.path.dirname(path), line.split('"')[1]))
            if dep not in seen and os.path.exists(dep):
                seen.add(dep)
                _includes(dep, seen)


def _stale(src_path, obj):
    """An object is rebuilt when its source, a header it includes (transitively) or this script is newer than it."""
    if not os.path.exists(obj):
        return True
    deps = {src_path, os.path.abspath(__file__)}
    _includes(src_path, deps)
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, ablations=False, jobs=None):
    """Product library by default.  ablations=True builds a SEPARATE library with -DNM_ABLATIONS (the MLP kernel
    A/B variants, some of which compute wrong results on purpose, selectable there by NM_MLP_VARIANT); nothing in
    the package loads it -- scripts/bench_mlp.py points nerfmeshes_amd._lib at it explicitly.
    Per-object incremental: only the translation units whose source or included headers changed are recompiled
    (`jobs` of them at a time, default = the host's cores)."""
    lib_path = ABLATION_LIB_PATH if ablations else LIB_PATH
    bdir = os.path.join(CSRC, "build_ablations" if ablations else "build")
    os.makedirs(bdir, exist_ok=True)
    objs, todo = [], []
    for src in _present(SOURCES):
        obj = os.path.join(bdir, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if force or _stale(os.path.join(CSRC, src), obj):
            todo.append((src, obj))
    if not todo and os.path.exists(lib_path) and all(os.path.getmtime(o) <= os.path.getmtime(lib_path) for o in objs):
        return lib_path
    jobs = jobs or int(os.environ.get("NM_BUILD_JOBS", "0")) or os.cpu_count() or 4
    # the largest translation units first, so that they are not what the build waits for at the end
    todo.sort(key=lambda so: -_weight(so[0]))
    running, failed = [], None
    while (todo or running) and failed is None:
        while todo and len(running) < jobs:
            src, obj = todo.pop(0)
            cmd = [hipcc()] + FLAGS + (["-DNM_ABLATIONS"] if ablations else []) + ["-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            running.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        src, p = running.pop(0)
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = f"hipcc failed on {src}:\n{out.decode()}"
        elif verbose and out.strip():
            print(out.decode())
    if failed is not None:
        for _, p in running:
            p.kill()
        raise RuntimeError(failed)
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return lib_path


# compile cost of the translation units (CPU seconds on this image, measured in round 6): scheduling order only -- the split
# kernels' long-encoding instantiations were the critical path (172 s in one unit: two units now); 740 s of CPU in all
_COST = {"nerf_mlp_generic_s_long.hip": 90, "nerf_mlp_generic_s_long2.hip": 90, "nerf_mlp_generic_s.hip": 74, "nerf_mlp_generic_b_long.hip": 67, "nerf_mlp_generic_c_long.hip": 65,
         "nerf_mlp_generic_a_long.hip": 63, "nerf_mlp_generic_d_long.hip": 39, "nerf_mlp_generic_c.hip": 34, "nerf_mlp_generic_b.hip": 31,
         "nerf_dw_g.hip": 29, "nerf_mlp_generic_a.hip": 25, "nerf_mlp_generic_d.hip": 24, "nerf_mlp.hip": 13, "nerf_train.hip": 10}


def _weight(src):
    return _COST.get(src, 3)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, ablations="--ablations" in sys.argv))
