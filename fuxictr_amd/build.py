"""Build recipe for libfxctr.so (gfx950 / MI355X only): hipcc, in-tree output.

The .so lands next to this file so that it travels to the GPU box with the repo snapshot.
hipcc cross-compiles without a GPU, so `build()` is also the CPU-side "does it build" check.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ_DIR = os.path.join(CSRC, "build")
LIB_PATH = os.path.join(HERE, "libfxctr.so")
SOURCES = ["fx_api.cpp", "fx_embed.hip", "fx_sparse.hip", "fx_gemm.hip", "fx_gemm_x6.hip", "fx_din.hip", "fx_din_attn.hip", "fx_cin.hip", "fx_cin_mfma.hip", "fx_metrics.hip", "fx_fused.hip", "fx_sort.hip", "fx_series.hip", "fx_dedup_lds.hip"]
HEADERS = [os.path.join(CSRC, "fx_common.h"), os.path.join(CSRC, "fx_cin.h"), os.path.join(CSRC, "fx_gemm_int.h"),
           os.path.join(HERE, "..", "include", "fxctr.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
            "--offload-arch=" + ARCH]
# per-file additions.  fx_cin_mfma.hip: the arithmetic on MFMA result tiles must stay scalar v_fma_f32 (a
# packed v_pk_fma_f32 holds the matrix pipe ~22 cycles per issue)
EXTRA_FLAGS = {"fx_cin_mfma.hip": ["-fno-slp-vectorize"]}


def _digest(paths):
    h = hashlib.sha256()
    h.update(" ".join(CXXFLAGS).encode())
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build(force=False, verbose=True):
    """Compile every HIP translation unit for gfx950 and link libfxctr.so. Returns the path."""
    os.makedirs(OBJ_DIR, exist_ok=True)
    objs, todo = [], []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(OBJ_DIR, src.rsplit(".", 1)[0] + ".o")
        stamp = obj + ".sha"
        dig = _digest([sp] + HEADERS) + "".join(EXTRA_FLAGS.get(src, []))
        fresh = (not force and os.path.exists(obj) and os.path.exists(stamp)
                 and open(stamp).read() == dig)
        if not fresh:
            todo.append((sp, obj, stamp, dig))
        objs.append(obj)

    def compile_one(job):
        sp, obj, stamp, dig = job
        cmd = [HIPCC] + CXXFLAGS + EXTRA_FLAGS.get(os.path.basename(sp), []) + ["-x", "hip", "-c", sp, "-o", obj]
        if verbose:
            print("[fuxictr_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(stamp, "w") as f:
            f.write(dig)

    if todo:
        # the translation units are independent: compile them side by side
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 1, 8)) as pool:
            list(pool.map(compile_one, todo))
    rebuilt = bool(todo)
    if rebuilt or force or not os.path.exists(LIB_PATH):
        cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB_PATH] + objs
        if verbose:
            print("[fuxictr_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB_PATH)
