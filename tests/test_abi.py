"""CPU: libgsrast.so loads, exports every function include/gs_rasterizer.h declares, its structs
have the layout the ctypes binding assumes, and it refuses to work without a GPU (no fallback)."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "gs_rasterizer.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gs_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from taichi_3d_gaussian_splatting_amd import _native
    L = _native.lib()
    names = _declared_functions()
    assert len(names) >= 12
    for n in names:
        assert hasattr(L, n), f"libgsrast.so does not export {n}"
    assert sorted(_native.SYMBOLS) == names
    assert L.gs_abi_version() == _native.ABI_VERSION
    # every name gs_kernel_names() lists is a kernel that exists in the sources (its position is its timing id)
    kernels = L.gs_kernel_names().decode().split(",")
    csrc = os.path.join(ROOT, "taichi_3d_gaussian_splatting_amd", "csrc")
    text = "".join(open(os.path.join(csrc, f)).read() for f in os.listdir(csrc) if f.endswith(".hip"))
    assert len(kernels) == len(set(kernels)) == 13
    for k in kernels:
        assert re.search(r"__global__[^;{]*\b" + k + r"\s*\(", text), f"{k} is listed by gs_kernel_names() but is not a kernel"


def test_struct_layouts_match_the_header(tmp_path):
    """Compile a probe against the real header with gcc and compare sizeof/offsetof with ctypes."""
    from taichi_3d_gaussian_splatting_amd import _native
    structs = {"gs_config": _native.GsConfig, "gs_scene": _native.GsScene, "gs_camera": _native.GsCamera,
               "gs_forward_out": _native.GsForwardOut, "gs_frame_info": _native.GsFrameInfo,
               "gs_backward_out": _native.GsBackwardOut}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', 'int main(void){']
    for cname, cls in structs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines.append('printf("GS_X_COUNT_ %d\\n", (int)GS_X_COUNT_); return 0; }')
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", str(src), "-o", str(exe)])   # header is plain C
    got = dict(l.split() for l in subprocess.check_output([str(exe)]).decode().splitlines())
    for cname, cls in structs.items():
        assert int(got[cname]) == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, f"{cname}.{fname}"
    assert int(got["GS_X_COUNT_"]) == len(_native.EXPORTS)


def test_no_fallback_without_gpu():
    """On a box without a GPU the library reports an error; nothing silently runs on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from taichi_3d_gaussian_splatting_amd import _native
    L = _native.lib()
    h = C.c_void_p()
    rc = L.gs_create(0, C.byref(h))
    assert rc < 0 and b"hip" in L.gs_last_error().lower()


def test_product_never_touches_the_oracle():
    """The shipped package must not import, link or mention oracle/ (it is test infrastructure)."""
    pkg = os.path.join(ROOT, "taichi_3d_gaussian_splatting_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "gs_oracle" not in text and "from oracle" not in text and "import oracle" not in text, fn
    out = subprocess.check_output(["ldd", os.path.join(pkg, "lib", "libgsrast.so")]).decode()
    assert "gsoracle" not in out and "torch" not in out


def test_missing_library_fails_loudly(monkeypatch):
    from taichi_3d_gaussian_splatting_amd import _native
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", "/nonexistent/libgsrast.so")
    with pytest.raises(_native.NativeLibraryError):
        _native.lib()
