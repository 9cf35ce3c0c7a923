"""Build recipe for ``libtiatoolbox_amd.so`` (hipcc, gfx950 only, in-tree)."""

from __future__ import annotations

import shutil
import subprocess
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
    "-shared",
    "-fno-gpu-rdc",
    "-Wno-unused-result",
]


def sources() -> list[Path]:
    return sorted(CSRC.glob("*.hip"))


def needs_build() -> bool:
    if not LIB_PATH.exists():
        return True
    t = LIB_PATH.stat().st_mtime
    deps = [*sources(), *CSRC.glob("*.hpp"), *(ROOT / "include").glob("*.h")]
    return any(p.stat().st_mtime > t for p in deps)


def build(*, force: bool = False, verbose: bool = False, defines: tuple[str, ...] = (), out: Path | None = None) -> Path:
    """Compile every HIP source into one shared library (cross-compiles without a GPU).

    ``defines`` / ``out`` build an experimental variant next to the product library (e.g.
    ``build(defines=("TIA_F32_BINS=1",), out=LIB_DIR / "libtiatoolbox_amd_f32bins.so")``); select it at run time with
    the environment variable ``TIA_LIB_PATH`` (see ``_lib.lib_path``).
    """
    target = Path(out) if out is not None else LIB_PATH
    if out is None and not defines and not force and not needs_build():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    LIB_DIR.mkdir(parents=True, exist_ok=True)
    cmd = [hipcc, *HIPCC_FLAGS, *[f"-D{d}" for d in defines], f"-I{ROOT / 'include'}", f"-I{CSRC}",
           *[str(s) for s in sources()], "-o", str(target)]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return target


if __name__ == "__main__":
    print(build(force=True, verbose=True))
