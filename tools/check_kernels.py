#!/usr/bin/env python3
"""Build-time proof that libhiprz.so cannot name a kernel its code objects do not hold (no GPU needed; part of __graft_entry__.build()).

Why: round 3's GPU log gpurun_out/r03/pytest7.log ends in
    hip_global.cpp:109 ... Cannot find Symbol with name: _ZN5hiprz15rz_shade_kernelILb1ELb0ELb0ELi8EEE...   Fatal Python error: Aborted
— the host half of a translation unit had registered the stub of a kernel instantiation whose device half was not in the code object that
travelled to the GPU box, and the HIP runtime turns that into a process abort at the first launch, which no return code can catch.

What is checked, per device translation unit (*.o built from *.hip) and for the linked library:
  1. every host stub (`__device_stub__<kernel>`, what hipLaunchKernelGGL calls) has its kernel descriptor (`<kernel>.kd`) in the gfx950 code
     object embedded in the SAME object file, and the other way round;
  2. no kernel is instantiated in two translation units (the stubs are weak symbols: the linker would keep one and both units would register
     the same host address with their own code object);
  3. no object is older than a source or header it was compiled from (the objects ship prebuilt to the GPU box; `make` guarantees this only
     for the dependencies its rules list — this check reads the compiler's own -MD style list from `hipcc -M` when asked with --deps);
  4. libhiprz.so holds the union of the units' kernels.
Exit code 0 and one summary line, or 1 and the offending names."""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rayzath_amd", "csrc")
LLVM = os.environ.get("ROCM_LLVM", "/opt/rocm/lib/llvm/bin")
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def run(*cmd):
    return subprocess.run(cmd, check=True, capture_output=True, text=True).stdout


def host_stubs(path):
    """mangled kernel names the host half of an object / library can launch"""
    out = run("nm", "--defined-only", path)
    names = set()
    for line in out.splitlines():
        sym = line.split()[-1]
        # _ZN5hiprz30__device_stub__rz_shade_kernelI...  ->  _ZN5hiprz15rz_shade_kernelI...   (the length prefix of the name loses the 15 characters)
        if "__device_stub__" in sym:
            names.add(re.sub(r"(\d+)__device_stub__", lambda m: str(int(m.group(1)) - len("__device_stub__")), sym, count=1))
    return names


def quoted_includes(path, seen=None):
    """the repo's own files a source includes, transitively (#include "..." resolved against csrc/ and include/)"""
    seen = set() if seen is None else seen
    for name in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(path).read(), flags=re.M):
        for base in (os.path.dirname(path), CSRC, os.path.join(ROOT, "include")):
            cand = os.path.normpath(os.path.join(base, name))
            if os.path.exists(cand):
                if cand not in seen:
                    seen.add(cand)
                    quoted_includes(cand, seen)
                break
    return seen


def device_kernels(path):
    """mangled names of the kernels in the gfx950 code object(s) embedded in an object / library"""
    names = set()
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fatbin")
        # (an explicit output file: without one objcopy rewrites its INPUT in place — a new time stamp on the object, and for the library
        # exactly the in-place overwrite of a possibly mapped file that the Makefile's link-and-rename avoids)
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", path, os.path.join(tmp, "discard")], check=True, capture_output=True)
        data = open(fat, "rb").read()
        # a linked library concatenates the units' bundles: split at the bundler's magic
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(re.escape(magic), data)]
        for k, s in enumerate(starts):
            part = os.path.join(tmp, f"bundle{k}")
            open(part, "wb").write(data[s:starts[k + 1] if k + 1 < len(starts) else len(data)])
            co = os.path.join(tmp, f"co{k}")
            subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={part}", f"--targets={TARGET}", f"--output={co}"],
                           check=True, capture_output=True)
            for line in run(os.path.join(LLVM, "llvm-readelf"), "--dyn-syms", "-W", co).splitlines():
                f = line.split()
                if f and f[-1].endswith(".kd"):
                    names.add(f[-1][:-3])
    return names


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--deps", action="store_true", help="also ask hipcc -M for every unit's real include list and compare time stamps (seconds per unit)")
    args = ap.parse_args()
    units = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    problems, owner, total = [], {}, set()
    for unit in units:
        obj = os.path.join(CSRC, unit[:-4] + ".o")
        if not os.path.exists(obj):
            problems.append(f"{unit}: no object file (run make -C rayzath_amd/csrc)")
            continue
        stubs, kernels = host_stubs(obj), device_kernels(obj)
        for name in sorted(stubs - kernels):
            problems.append(f"{unit}: the host half launches {name} but the unit's gfx950 code object does not hold it")
        for name in sorted(kernels - stubs):
            problems.append(f"{unit}: the gfx950 code object holds {name} but the host half has no stub for it")
        for name in stubs:
            if name in owner:
                problems.append(f"{name} is instantiated in {owner[name]} and in {unit}")
            owner[name] = unit
        total |= stubs
        # staleness: the object against everything it was compiled from
        deps = [os.path.join(CSRC, unit)]
        if args.deps:
            text = run("/opt/rocm/bin/hipcc", "-M", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"), "-I", CSRC, os.path.join(CSRC, unit))
            deps += [d for d in text.replace("\\\n", " ").split()[1:] if d.startswith(ROOT)]
        else:
            deps += sorted(quoted_includes(os.path.join(CSRC, unit)))
        for d in sorted(set(deps)):
            if os.path.getmtime(d) > os.path.getmtime(obj):
                problems.append(f"{unit}: {os.path.basename(obj)} is older than {os.path.relpath(d, ROOT)}")
    lib = os.path.join(CSRC, "libhiprz.so")
    if not os.path.exists(lib):
        problems.append("libhiprz.so is missing")
    else:
        lib_kernels = device_kernels(lib)
        for name in sorted(total - lib_kernels):
            problems.append(f"libhiprz.so: {name} is launched by a unit but not in the library's code objects")
        for unit in units:
            obj = os.path.join(CSRC, unit[:-4] + ".o")
            if os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(lib):
                problems.append(f"libhiprz.so is older than {unit[:-4]}.o")
    if problems:
        print("\n".join(problems))
        return 1
    print(f"check_kernels: {len(total)} kernel instantiations in {len(units)} translation units, every host stub has its gfx950 kernel, no duplicates, no stale object")
    return 0


if __name__ == "__main__":
    sys.exit(main())
