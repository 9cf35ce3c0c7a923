"""Records which third-party libraries of the reference's primitive layer a machine has (run on the GPU box once per round;
the answer decides whether ``tests/test_real_libs.py`` pins the oracle's cv2 / skimage restatements there or skips)."""
import importlib
import platform
import sys

print("python", sys.version.split()[0], platform.platform())
for name in ("cv2", "skimage", "torchvision", "scipy", "sklearn", "numpy", "torch", "pandas", "PIL", "shapely", "zarr", "dask",
             "numcodecs", "openslide", "tifffile", "imagecodecs", "albumentations"):
    try:
        mod = importlib.import_module(name)
        print(f"{name:16s} {getattr(mod, '__version__', '?')}")
    except Exception as exc:  # noqa: BLE001
        print(f"{name:16s} MISSING ({type(exc).__name__})")
