"""The C-ABI library loads on a machine without a GPU, exports every symbol include/hiprz.h
declares, and its POD layouts agree with the Python mirror.  No compute call is made."""
import ctypes as C
import os
import re

from rayzath_amd import _abi, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "hiprz.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hiprz_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(built):
    lib = _lib.load()
    names = _declared_symbols()
    assert len(names) >= 35
    for name in names:
        assert hasattr(lib, name), f"{name} is declared in include/hiprz.h but not exported"
    assert set(names) == set(_abi.ENTRY_POINTS), set(names) ^ set(_abi.ENTRY_POINTS)


def test_record_layouts_match_the_header(built):
    lib = _lib.load()
    sizes = (C.c_uint32 * 13)()
    lib.hiprz_abi_sizes(sizes)
    order = ["node", "tri", "tri_attr", "instance", "material", "texture", "spot_light", "direct_light"]
    for i, name in enumerate(order):
        assert sizes[i] == _abi.RECORD_SIZES[name] == _abi.RECORD_DTYPES[name].itemsize, name
    for i, struct in enumerate((_abi.Scene, _abi.Camera, _abi.Config, _abi.Counters, _abi.MeshDesc)):
        assert sizes[8 + i] == C.sizeof(struct), struct.__name__
    # device code fetches records as float4: sizes must be multiples of 16 bytes
    for name in order:
        assert _abi.RECORD_SIZES[name] % 16 == 0


def test_version_and_seed_table_are_host_side(built):
    lib = _lib.load()
    assert b"gfx950" in lib.hiprz_version()
    v = [lib.hiprz_seed_value(20240501, 0, i) for i in range(256)]
    assert all(-10.0 <= x < 10.0 for x in v) and len(set(v)) > 250


def test_create_fails_loudly_without_a_gpu(built):
    """On a CPU-only machine the backend must refuse to come up (there is no CPU path)."""
    import torch
    if torch.cuda.is_available():
        return
    lib = _lib.load()
    ctx = C.c_void_p()
    rc = lib.hiprz_create(C.byref(ctx), 0)
    assert rc == _abi.ERR_DEVICE and not ctx
    assert b"HIP device" in lib.hiprz_last_error(None)


def test_every_launch_site_registered_its_kernel_and_the_built_files_hold_them(built):
    """hiprz_kernel_count(): the kernel instantiations the launchers can select, registered when the library was loaded (RZ_LAUNCH); the first
    hiprz_create on a device resolves each of them in the loaded code objects.  tools/check_kernels.py proves on the built files — no GPU
    needed — that every host stub has its gfx950 kernel, that no kernel is instantiated in two units and that no object is older than a
    file it includes; its count is the library's."""
    import subprocess
    import sys
    n = _lib.load().hiprz_kernel_count()
    assert n >= 200
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_kernels.py")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert int(re.search(r"check_kernels: (\d+) kernel instantiations", out.stdout).group(1)) == n
