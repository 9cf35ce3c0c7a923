"""Build recipe for ``libtiatoolbox_amd.so`` (hipcc, gfx950 only, in-tree)."""

from __future__ import annotations

import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
LIB_DIR = PKG / "lib"
LIB_PATH = LIB_DIR / "libtiatoolbox_amd.so"

HIPCC_FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-fno-gpu-rdc",
    "-Wno-unused-result",
]


def sources() -> list[Path]:
    return sorted(CSRC.glob("*.hip"))


def _headers() -> list[Path]:
    return [*CSRC.glob("*.hpp"), *CSRC.glob("*.h"), *CSRC.glob("*.inc"), *(ROOT / "include").glob("*.h")]


def needs_build() -> bool:
    if not LIB_PATH.exists():
        return True
    t = LIB_PATH.stat().st_mtime
    return any(p.stat().st_mtime > t for p in [*sources(), *_headers()])


def build(*, force: bool = False, verbose: bool = False, defines: tuple[str, ...] = (), out: Path | None = None) -> Path:
    """Compile every HIP source for gfx950 (cross-compiles without a GPU) and link them into one shared library.

    One ``hipcc -c`` per source, in parallel, into an object directory next to the target (``<target>.obj/``; objects are
    re-used while they are newer than their source and every header), then one link.  ``defines`` / ``out`` build an
    experimental variant next to the product library (e.g. ``build(defines=("TIA_F32_BINS=1",), out=LIB_DIR /
    "libtiatoolbox_amd_f32bins.so")``); select it at run time with the environment variable ``TIA_LIB_PATH`` (see
    ``_lib.lib_path``).
    """
    target = Path(out) if out is not None else LIB_PATH
    if out is None and not defines and not force and not needs_build():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    target.parent.mkdir(parents=True, exist_ok=True)
    obj_dir = target.parent / (target.name + ".obj")
    obj_dir.mkdir(parents=True, exist_ok=True)
    common = [hipcc, *HIPCC_FLAGS, *[f"-D{d}" for d in defines], f"-I{ROOT / 'include'}", f"-I{CSRC}"]
    stamp = obj_dir / "flags.txt"
    flags_text = " ".join(common)
    if force or not stamp.exists() or stamp.read_text() != flags_text:
        for o in obj_dir.glob("*.o"):
            o.unlink()
        stamp.write_text(flags_text)
    newest_header = max((h.stat().st_mtime for h in _headers()), default=0.0)

    def compile_one(src: Path) -> Path:
        obj = obj_dir / (src.stem + ".o")
        if obj.exists() and obj.stat().st_mtime > max(src.stat().st_mtime, newest_header):
            return obj
        cmd = [*common, "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(sources()), os.cpu_count() or 4)) as pool:
        objects = list(pool.map(compile_one, sources()))
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", *[str(o) for o in objects], "-o", str(target)]
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.run(link, check=True)
    return target


if __name__ == "__main__":
    print(build(force=True, verbose=True))
