"""Build the C-ABI shared library of hand-written HIP kernels for gfx950 (MI355X).

    python -m virtex_amd.build            # hipcc --offload-arch=gfx950  -> virtex_amd/lib/libvirtex_amd.so
    python -m virtex_amd.build --emu      # CPU fiber-emulator build of the SAME sources (tests only)

hipcc cross-compiles without a GPU.  Objects are rebuilt only when a source/header changed.
"""
import argparse
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libvirtex_amd.so")
EMU_DIR = os.path.join(ROOT, "tests", "hipemu")
EMU_LIB_PATH = os.path.join(EMU_DIR, "libvirtex_amd_emu.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HOST_CLANG = os.environ.get("VTX_HOST_CLANG", "/opt/rocm/lib/llvm/bin/clang++")

HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
             "-ffp-contract=fast", "-Wno-unused-result"]
EMU_FLAGS = ["-x", "c++", "-O2", "-g", "-std=c++17", "-fPIC", "-pthread", "-DHIPEMU=1",
             "-I", os.path.join(EMU_DIR, "include"), "-Wno-unused-value", "-Wno-unknown-pragmas",
             "-Wno-pass-failed", "-ffp-contract=off"]


def csrc_hash() -> str:
    """sha256 over every kernel source and header (name + content, sorted): what a measured PMC traffic table is valid for
    (profiles/traffic_table.json, bench.py's roofline.traffic)."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    files.append(os.path.join(ROOT, "include", "virtex_amd.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime(extra=()):
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(ROOT, "include", "virtex_amd.h"))
    hdrs.extend(extra)
    return max(os.path.getmtime(h) for h in hdrs)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


# per-file flags of the HIP build.  conv3_bwd.hip: hipcc's SLP vectoriser turns the BatchNorm arithmetic of the transform and
# the epilogue into v_pk_*_f32; the kernel is issue- / latency-bound with two waves per SIMD and runs 4 % faster without
# (225.5 vs 234.1 us, profiles/r05_conv3_bwd_variants.txt)
FILE_FLAGS = {"conv3_bwd.hip": ["-fno-slp-vectorize"]}


def _compile_all(compiler, flags, objdir, hdr_mtime, verbose):
    os.makedirs(objdir, exist_ok=True)
    jobs, objs = [], []
    for src in _sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), hdr_mtime):
            continue
        extra = FILE_FLAGS.get(os.path.basename(src), []) if compiler == HIPCC else []
        jobs.append([compiler] + flags + extra + ["-c", src, "-o", obj])
    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for out in ex.map(_run, jobs):
                if verbose and out.strip():
                    print(out)
    return objs, bool(jobs)


def build_hip(verbose=False, variant=None, extra_flags=()) -> str:
    """variant / extra_flags: an A/B build next to the product library (libvirtex_amd_<variant>.so, own object
    directory; selected at run time with VIRTEX_AMD_LIB) -- measurement sessions only."""
    lib = LIB_PATH if not variant else os.path.join(LIB_DIR, f"libvirtex_amd_{variant}.so")
    objdir = os.path.join(LIB_DIR, "obj" if not variant else f"obj_{variant}")
    objs, changed = _compile_all(HIPCC, HIP_FLAGS + list(extra_flags), objdir, _deps_mtime(), verbose)
    if changed or not os.path.exists(lib):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return lib


def build_emu(verbose=False) -> str:
    emu_hdr = os.path.join(EMU_DIR, "include", "hip", "hip_runtime.h")
    rt_src = os.path.join(EMU_DIR, "hipemu.cpp")
    objdir = os.path.join(EMU_DIR, "obj")
    objs, changed = _compile_all(HOST_CLANG, EMU_FLAGS, objdir, _deps_mtime([emu_hdr]), verbose)
    rt_obj = os.path.join(objdir, "hipemu_rt.o")
    if (not os.path.exists(rt_obj)
            or os.path.getmtime(rt_obj) < max(os.path.getmtime(rt_src), os.path.getmtime(emu_hdr))):
        _run([HOST_CLANG] + EMU_FLAGS + ["-c", rt_src, "-o", rt_obj])
        changed = True
    if changed or not os.path.exists(EMU_LIB_PATH):
        _run([HOST_CLANG, "-shared", "-fPIC", "-pthread", "-o", EMU_LIB_PATH, rt_obj] + objs)
    return EMU_LIB_PATH


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--emu", action="store_true")
    ap.add_argument("-v", "--verbose", action="store_true")
    ap.add_argument("--variant", default=None, help="A/B build: libvirtex_amd_<variant>.so")
    ap.add_argument("--define", action="append", default=[], help="extra -D for an A/B build")
    ap.add_argument("--flag", action="append", default=[], help="extra compiler flag for an A/B build (e.g. --flag=-fno-slp-vectorize)")
    a = ap.parse_args()
    path = build_emu(a.verbose) if a.emu else build_hip(a.verbose, a.variant, ["-D" + d for d in a.define] + list(a.flag))
    print(path)


if __name__ == "__main__":
    sys.exit(main())
