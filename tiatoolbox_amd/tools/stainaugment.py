"""Stain augmentation (API of reference ``tiatoolbox/tools/stainaugment.py:15-249``).

``fit`` estimates (or takes) the stain matrix and the luminosity tissue mask parameters;
``augment`` perturbs the concentrations ``C[mask,i] = C[mask,i]*alpha_i + beta_i`` and
recomposes the image in one streaming HIP kernel (``tia_stain_augment_u8``).  albumentations is
an optional dependency of the reference (``ImageOnlyTransform`` base class): when it is
importable the class derives from it, otherwise a minimal stand-in keeps the same call surface
(``apply``, ``get_params``, ``__call__(image=...)``).
"""

from __future__ import annotations

import numpy as np
import torch

from tiatoolbox_amd import _lib
from tiatoolbox_amd.tools import _stain_device as dev
from tiatoolbox_amd.tools.stainnorm import get_normalizer
from tiatoolbox_amd.utils import _tensors

try:  # pragma: no cover - albumentations is not installed in the build container
    from albumentations.core.transforms_interface import ImageOnlyTransform
except ImportError:

    class ImageOnlyTransform:  # type: ignore[no-redef]
        """Stand-in with albumentations' calling convention."""

        def __init__(self, always_apply: bool = False, p: float = 0.5) -> None:  # noqa: FBT001, FBT002
            self.always_apply = always_apply
            self.p = p

        def __call__(self, *, image: np.ndarray, force_apply: bool = False, **kwargs) -> dict:
            rng = np.random.default_rng()
            if force_apply or self.always_apply or rng.random() < self.p:
                image = self.apply(image, **self.get_params())
            return {"image": image, **kwargs}


class StainAugmentor(ImageOnlyTransform):
    """Stain augmentation in concentration space (ref. :15-249)."""

    def __init__(self, method: str = "vahadane", stain_matrix: np.ndarray | None = None, sigma1: float = 0.4,
                 sigma2: float = 0.2, p: float = 0.5, *, augment_background: bool = False,
                 always_apply: bool = False, precision: str = "f64") -> None:
        super().__init__(always_apply=always_apply, p=p)
        if precision not in {"f64", "f32"}:
            msg = "precision must be 'f64' (the reference's arithmetic) or 'f32' (fast path)."
            raise ValueError(msg)
        self.precision = precision
        self.augment_background = augment_background
        self.sigma1 = sigma1
        self.sigma2 = sigma2
        self.method = method
        self.stain_matrix = stain_matrix
        if self.method.lower() not in {"macenko", "vahadane"}:
            msg = (f"Unsupported stain extractor method {self.method!r} "
                   f"for StainAugmentor. Choose either 'vahadane' or 'macenko'.")
            raise ValueError(msg)
        self.stain_normalizer = get_normalizer(self.method.lower())
        self.alpha: float
        self.beta: float
        self.img_shape: tuple[int, ...]
        self.n_stains: int = 2
        self._batch = None
        self._own_matrix = False
        self._stats = None
        self._y_thr = 0
        self._kind = "np3"

    def fit(self, img, threshold: float = 0.85) -> None:
        """Stain matrix + concentrations + tissue-mask parameters of ``img`` (ref. :141-175).

        The reference computes the mask *after* ``rgb2od`` replaced zeros by ones in ``img``
        in place (:163-175); ``zero_to_one`` reproduces that.
        """
        batch, kind = _tensors.to_device_batch(img)
        n = batch.shape[0]
        if self.stain_matrix is None or self._own_matrix:
            # reference behaviour for one image (:153-161); for an NHWC batch every patch gets its own matrix, which is
            # then not kept as `stain_matrix` (a later fit() on other images estimates afresh)
            sm = self.stain_normalizer.extractor.get_stain_matrix(batch)
            sm = sm if isinstance(sm, torch.Tensor) else torch.from_numpy(np.asarray(sm))
            sm = sm.reshape(-1, 2, 3)
            self.stain_matrix = sm[0].cpu().numpy() if n == 1 else sm.cpu().numpy()
            self._own_matrix = n != 1
        else:
            sm = torch.from_numpy(np.asarray(self.stain_matrix, dtype=np.float64).reshape(1, 2, 3)).expand(n, 2, 3)
        params = dev.make_params(mode=_lib.MODE_GIVEN, luminosity_threshold=threshold, zero_to_one=True)
        self._stats = dev.stain_stats(batch, params, stain_given=sm.contiguous())
        self._y_thr = params.y_thr
        self._batch, self._kind = batch, kind
        self.n_stains = 2
        self.img_shape = tuple(batch.shape[1:]) if kind.endswith("3") else tuple(batch.shape)

    @property
    def source_concentrations(self) -> np.ndarray:
        conc = dev.concentrations(self._batch, self._stats)
        return _tensors.from_device(conc, self._kind)

    @property
    def tissue_mask(self) -> np.ndarray:
        mask = dev.luminosity_mask(self._batch, self._stats, self._y_thr, zero_to_one=True)
        return _tensors.from_device(mask.flatten(1), self._kind)

    def augment(self, alpha_beta: np.ndarray | None = None):
        """Augmented image(s) from the fitted source (ref. :177-206).

        ``alpha_beta`` (``[N,4] = a0,a1,b0,b1``) injects the random draw; by default one
        ``get_params()`` draw per stain channel, as the reference does.
        """
        n = self._batch.shape[0]
        if alpha_beta is None:
            ab = np.empty((n, 4))
            for k in range(n):
                for i in range(self.n_stains):
                    self.get_params()
                    ab[k, i], ab[k, 2 + i] = self.alpha, self.beta
        else:
            ab = np.asarray(alpha_beta, dtype=np.float64).reshape(n, 4)
        if not self.augment_background:
            empty = ~dev.luminosity_mask(self._batch, self._stats, self._y_thr, zero_to_one=True).flatten(1).any(1)
            if bool(empty.any()):
                msg = "Empty tissue mask computed."
                raise ValueError(msg)
        out = dev.augment(self._batch, self._stats, torch.from_numpy(ab).to(self._batch.device), self._y_thr,
                          augment_background=self.augment_background, zero_to_one=True,
                          math=_lib.MATH_F32 if self._fast_path_ok() else _lib.MATH_F64)
        return _tensors.from_device(out, self._kind)

    def _fast_path_ok(self) -> bool:
        """f32 / 16-byte-access kernel: whole 3072-byte chunks per image (any H*W multiple of 1024)."""
        return self.precision == "f32" and (self._batch.shape[1] * self._batch.shape[2] * 3) % 3072 == 0

    def apply(self, img, **params):  # noqa: ARG002
        """``fit`` + ``augment`` (ref. :208-228)."""
        self.fit(img, threshold=0.85)
        return self.augment()

    def get_params(self) -> dict:
        """Draw ``alpha ~ U(1-s1, 1+s1)``, ``beta ~ U(-s2, s2)`` from a fresh generator (ref. :230-235)."""
        rng = np.random.default_rng()
        self.alpha = rng.uniform(1 - self.sigma1, 1 + self.sigma1)
        self.beta = rng.uniform(-self.sigma2, self.sigma2)
        return {}

    def get_params_dependent_on_targets(self, params: dict) -> dict:  # noqa: ARG002
        return {}

    @staticmethod
    def get_transform_init_args_names(**kwargs) -> tuple[str, ...]:  # noqa: ARG004
        return "method", "stain_matrix", "sigma1", "sigma2", "augment_background"
