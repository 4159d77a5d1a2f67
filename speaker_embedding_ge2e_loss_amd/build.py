"""Build libge2e_hip.so in-tree with hipcc for gfx950 (no JIT cache, no pip install).

    python -m speaker_embedding_ge2e_loss_amd.build [--force]

The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libge2e_hip.so")
TORCH_EXT_SRC = os.path.join(PKG_DIR, "csrc_torch", "ge2e_autograd.cpp")
TORCH_EXT_PATH = os.path.join(PKG_DIR, "libge2e_torch.so")      # the autograd node in C++ (host code only, links libge2e_hip.so)
INCLUDE = os.path.join(os.path.dirname(PKG_DIR), "include")
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, /opt/rocm/bin/hipcc, PATH)")


def sources() -> list[str]:
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def source_hash() -> str:
    """sha256 over the kernel sources, headers and per-source flags: identifies the code a profile was taken on
    (profiles/traffic.json carries it; bench.py reports measured traffic only while it still matches)."""
    import hashlib
    h = hashlib.sha256()
    for f in sources() + sorted(glob.glob(os.path.join(CSRC, "*.hpp"))) + sorted(glob.glob(os.path.join(INCLUDE, "*.h"))):
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(repr(sorted(EXTRA_FLAGS.items())).encode())
    return h.hexdigest()[:16]


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(INCLUDE, "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


# Per-source compile flags.  ge2e_team.hip: no SLP vectorisation -- hipcc packs the fp32 epilogue arithmetic that
# follows its (inline-asm) MFMA chains into v_pk_mul_f32 / v_pk_fma_f32, and on gfx950 those lost the fused-in term in
# the low register of a pair, lanes 48..63, in up to 70 % of the launches (hazards).
EXTRA_FLAGS = {"ge2e_team.hip": ["-fno-slp-vectorize"], "ge2e_team_fwd.hip": ["-fno-slp-vectorize"]}
OBJ_DIR = os.path.join(PKG_DIR, "csrc", "_obj")


def _compile_one(args):
    src, obj, verbose = args
    cmd = [_hipcc(), "-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-c", "-Wall", "-Wno-unused-function",
           f"-I{INCLUDE}"] + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-o", obj, src]
    if verbose:
        print("[ge2e build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return obj


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile every csrc/*.hip to an object (in parallel, only the stale ones) and link one shared object."""
    if not force and not is_stale():
        _build_torch_ext_optional(False, verbose)
        return LIB_PATH
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJ_DIR, exist_ok=True)
    hdr_t = max(os.path.getmtime(d) for d in glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(INCLUDE, "*.h"))
                + [os.path.abspath(__file__)])
    jobs, objs = [], []
    for src in sources():
        obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            jobs.append((src, obj, verbose))
    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(_compile_one, jobs))
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH + ".tmp"] + objs
    if verbose:
        print("[ge2e build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    _build_torch_ext_optional(True, verbose)
    return LIB_PATH


TORCH_EXT_STAMP = TORCH_EXT_PATH + ".stamp"      # the torch version the extension was compiled against


def torch_ext_is_stale() -> bool:
    """Older than its source / the header / the core library it links, or built against another torch."""
    if not os.path.exists(TORCH_EXT_PATH):
        return True
    t = os.path.getmtime(TORCH_EXT_PATH)
    deps = [TORCH_EXT_SRC] + glob.glob(os.path.join(INCLUDE, "*.h")) + ([LIB_PATH] if os.path.exists(LIB_PATH) else [])
    if any(os.path.getmtime(d) > t for d in deps):
        return True
    try:
        import torch
        with open(TORCH_EXT_STAMP) as f:
            return f.read().strip() != torch.__version__
    except OSError:
        return True


def build_torch_ext(force: bool = False, verbose: bool = True) -> str:
    """g++ the C++ autograd node (csrc_torch/ge2e_autograd.cpp: torch.ops.ge2e_amd.loss) against this interpreter's torch
    and the in-tree libge2e_hip.so.  Host code only -- every kernel stays in libge2e_hip.so."""
    if not force and not torch_ext_is_stale():
        return TORCH_EXT_PATH
    import torch
    tdir = os.path.dirname(torch.__file__)
    cxx = shutil.which("g++") or shutil.which("c++")
    if cxx is None:
        raise RuntimeError("g++ not found (needed for libge2e_torch.so)")
    abi = int(getattr(torch._C, "_GLIBCXX_USE_CXX11_ABI", True))
    cmd = [cxx, "-O2", "-std=c++17", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={abi}", "-Wno-deprecated-declarations",
           f"-I{tdir}/include", f"-I{tdir}/include/torch/csrc/api/include", "-I/opt/rocm/include", f"-I{INCLUDE}",
           TORCH_EXT_SRC, "-o", TORCH_EXT_PATH + ".tmp", f"-L{tdir}/lib", "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip",
           "-ltorch_hip", f"-L{PKG_DIR}", "-lge2e_hip", "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{tdir}/lib"]
    if verbose:
        print("[ge2e build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(TORCH_EXT_PATH + ".tmp", TORCH_EXT_PATH)
    with open(TORCH_EXT_STAMP, "w") as f:
        f.write(torch.__version__)
    return TORCH_EXT_PATH


def _build_torch_ext_optional(force: bool, verbose: bool) -> None:
    """The C++ autograd node is an accelerator of the host path, not a requirement: without g++ (or with a torch whose
    headers do not compile it) the Python node makes the same launches.  Say so and go on."""
    try:
        build_torch_ext(force=force, verbose=verbose)
    except (RuntimeError, subprocess.CalledProcessError, OSError) as ex:
        print(f"[ge2e build] libge2e_torch.so not built ({str(ex)[:160]}): the Python autograd node will be used", flush=True)
        for stale in (TORCH_EXT_PATH, TORCH_EXT_STAMP):
            if os.path.exists(stale):
                os.remove(stale)


def build_variant(out_path: str, defs: list[str], verbose: bool = False) -> str:
    """Diagnostic / experiment builds (tools/): the same per-source flags plus extra -D definitions, objects in a
    directory of their own, nothing shared with the product library."""
    from concurrent.futures import ThreadPoolExecutor
    odir = out_path + ".obj"
    os.makedirs(odir, exist_ok=True)

    def one(src):
        obj = os.path.join(odir, os.path.basename(src)[:-4] + ".o")
        cmd = [_hipcc(), "-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-c", f"-I{INCLUDE}"] + list(defs) + \
            EXTRA_FLAGS.get(os.path.basename(src), []) + ["-o", obj, src]
        if verbose:
            print("[ge2e build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, stderr=None if verbose else subprocess.DEVNULL)
        return obj
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(one, sources()))
    subprocess.run([_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", out_path] + objs, check=True)
    shutil.rmtree(odir, ignore_errors=True)
    return out_path


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
