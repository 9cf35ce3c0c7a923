"""Engine base class (API of reference ``tiatoolbox/models/engine/engine_abc.py``).

Keeps the reference's constructor / ``run()`` signature, kwargs-as-attributes behaviour,
validation errors and ``dict`` output, but the inner loop is device-first: patches are
uploaded in large pinned chunks, pre-processing (stain normalisation, ``ToTensor``) runs
batched on the GPU, outputs stay resident in HBM and are copied back once at the end.
With ``torch.distributed`` initialised (one process per GPU) the patch list is sharded
across ranks and the per-patch outputs are all-gathered (RCCL).
"""

from __future__ import annotations

import contextlib
import logging
import os
from pathlib import Path

import numpy as np
import torch

from tiatoolbox_amd import distributed as tdist
from tiatoolbox_amd.utils import tracing
from tiatoolbox_amd.models.architecture import get_pretrained_model
from tiatoolbox_amd.models.dataset.dataset_abc import PatchDataset
from tiatoolbox_amd.models.engine.io_config import ModelIOConfigABC
from tiatoolbox_amd.models.models_abc import load_torch_model

logger = logging.getLogger("tiatoolbox_amd")

_DTYPES = {"float32": torch.float32, "float16": torch.float16, "bfloat16": torch.bfloat16,
           "fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}


def _clear_last_hip_error() -> int:
    """Read-and-clear the HIP runtime's sticky last-error through the product library itself (``tia_clear_last_error``):
    it links the runtime instance whose ``hipGetLastError()`` every ``tia_*`` entry point checks after its launches
    (``torch.cuda.cudart()`` does not expose ``cudaGetLastError``).  Returns the value that was pending."""
    from tiatoolbox_amd import _lib

    return int(_lib.load().tia_clear_last_error())


_WARNED_NO_PIN = False


class _HostFeed:
    """Asynchronous host -> device feed of uint8 patch batches.

    The caller's NumPy array is page-locked *in place* (``hipHostRegister``: no staging copy on the host) and
    every batch is copied on a dedicated stream one batch ahead of the compute stream, so the PCIe transfer of
    batch k+1 overlaps the kernels of batch k.  Falls back to a plain synchronous copy when registration fails.
    """

    def __init__(self, array: np.ndarray, device: torch.device) -> None:
        self.array, self.device = array, device
        self.stream = torch.cuda.Stream(device)
        self.registered = False
        self._pending: dict[tuple[int, int], tuple[torch.Tensor, torch.cuda.Event]] = {}
        if (os.environ.get("TIA_HOST_REGISTER", "1") == "1" and array.flags.c_contiguous and array.flags.writeable
                and array.nbytes >= (1 << 22)):
            try:
                rc = torch.cuda.cudart().cudaHostRegister(array.ctypes.data, array.nbytes, 0)
                self.registered = int(getattr(rc, "value", rc)) == 0
            except Exception:  # noqa: BLE001  (any runtime refusal: keep the synchronous path)
                self.registered = False
            if not self.registered:
                # a refused registration (mmap'd / already registered memory) leaves the runtime's sticky last-error
                # set; clear it so the next launch check (tia_* entry points return hipGetLastError()) is not blamed
                _clear_last_hip_error()
                global _WARNED_NO_PIN  # noqa: PLW0603
                if not _WARNED_NO_PIN:
                    _WARNED_NO_PIN = True
                    logger.warning("hipHostRegister refused the input array (%d bytes): host patches are copied synchronously, "
                                   "batch by batch, instead of one batch ahead on the copy stream.", array.nbytes)

    def prefetch(self, lo: int, hi: int) -> None:
        if not self.registered or (lo, hi) in self._pending or lo >= hi:
            return
        host = torch.from_numpy(self.array[lo:hi])
        with torch.cuda.stream(self.stream):
            dev = host.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self._pending[(lo, hi)] = (dev, ev)

    def get(self, lo: int, hi: int) -> torch.Tensor:
        if not self.registered:
            return torch.from_numpy(np.ascontiguousarray(self.array[lo:hi])).to(self.device)
        self.prefetch(lo, hi)
        dev, ev = self._pending.pop((lo, hi))
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)
        dev.record_stream(cur)
        return dev

    def close(self) -> None:
        if self.registered:
            torch.cuda.synchronize(self.device)
            torch.cuda.cudart().cudaHostUnregister(self.array.ctypes.data)
            self.registered = False


def _weights_version(model: torch.nn.Module) -> tuple:
    """Changes whenever a parameter/buffer is rewritten in place (``load_state_dict`` bumps ``Tensor._version``) or
    replaced (``data_ptr``): the derived inference copy (BN folded, cast) must then be rebuilt."""
    return tuple((t.data_ptr(), t._version) for t in list(model.parameters()) + list(model.buffers()))  # noqa: SLF001


def _make_save_dir(save_dir: Path, overwrite: bool) -> None:
    import shutil

    if save_dir.exists() and overwrite:
        shutil.rmtree(save_dir)
    save_dir.mkdir(parents=True)


def prepare_engines_save_dir(save_dir, *, patch_mode: bool, overwrite: bool = False, distributed: bool = False) -> Path | None:
    """Create or validate the output directory exactly like the reference (``engine_abc.py:1832-1885``): WSI mode without
    ``save_dir`` is an ``OSError``; an existing directory is removed first when ``overwrite`` and is a ``FileExistsError``
    otherwise (``mkdir(parents=True)``).

    ``distributed`` (one process per GPU, every rank calls ``run()``): rank 0 alone removes / creates the directory and
    broadcasts the outcome, so that the other ranks neither trip over the directory rank 0 has just made nor remove it,
    and a refusal (``FileExistsError``, permissions) is raised by EVERY rank instead of leaving the others waiting in the
    run's first collective."""
    if patch_mode and save_dir is None:
        return None
    if save_dir is None:
        msg = "Input WSIs detected but no save directory provided. Please provide a 'save_dir'."
        raise OSError(msg)
    save_dir = Path(save_dir)
    rank, world_size = tdist.world() if distributed else (0, 1)
    if world_size == 1:
        _make_save_dir(save_dir, overwrite)
        return save_dir
    _rank0_does(lambda: _make_save_dir(save_dir, overwrite))
    return save_dir


_OS_ERRORS = {"FileExistsError": FileExistsError, "PermissionError": PermissionError, "NotADirectoryError": NotADirectoryError,
              "FileNotFoundError": FileNotFoundError, "IsADirectoryError": IsADirectoryError}


def _rank0_does(action) -> None:
    """Run a file-system ``action`` on rank 0 alone and make its outcome every rank's outcome: an ``OSError`` is broadcast as
    ``(class name, errno, strerror, filename)`` and re-raised everywhere as the same class with the same attributes (so
    ``exc.filename`` / ``exc.errno`` survive and the message is not doubled), instead of leaving the other ranks waiting in
    the next collective.  Any OTHER exception of the action (a ``ValueError`` / ``TypeError`` from the conversion work of a write
    closure, ``MemoryError`` ...) is broadcast too and raised on every rank as ``RuntimeError("<class>: <message>")`` -- rank 0
    re-raises the original -- so that no rank is ever left behind in ``broadcast_object_list``.  The broadcast also orders the
    action before every rank's return.  COLLECTIVE."""
    rank, _ = tdist.world()
    outcome: list = [None]
    original: BaseException | None = None
    if rank == 0:
        try:
            action()
        except OSError as exc:  # FileExistsError, PermissionError, disk full, ...
            outcome[0] = (type(exc).__name__, exc.errno, exc.strerror, exc.filename, str(exc))
        except Exception as exc:  # noqa: BLE001  (anything else must reach the other ranks as well)
            original = exc
            outcome[0] = ("!" + type(exc).__name__, None, None, None, f"{type(exc).__name__}: {exc}")
    torch.distributed.broadcast_object_list(outcome, src=0)
    if outcome[0] is not None:
        name, errno_, strerror, filename, text = outcome[0]
        if name.startswith("!"):
            if original is not None:
                raise original
            raise RuntimeError(f"rank 0 failed while writing outputs -- {text}")
        cls = _OS_ERRORS.get(name, OSError)
        if errno_ is None:
            raise cls(text)
        raise cls(errno_, strerror, filename) if filename is not None else cls(errno_, strerror)


def write_outputs(distributed: bool, action) -> None:
    """Write one slide's result files: every process when not distributed; otherwise rank 0 alone, with the outcome (including
    an ``OSError`` such as a full disk) shared by all ranks -- see :func:`_rank0_does`.  The returned paths exist on every rank
    when this returns."""
    if distributed and tdist.is_distributed():
        _rank0_does(action)
    else:
        action()


def outputs_written(distributed: bool) -> None:
    """Rank 0 writes a WSI run's files; the paths ``run()`` returns must exist on every rank when it returns.  (The engines now
    write through :func:`write_outputs`, whose broadcast already gives that guarantee; kept for callers that write themselves.)"""
    if distributed and tdist.is_distributed():
        torch.distributed.barrier()


def iter_row_outputs(infer, row_sels: list[np.ndarray], batch_size: int):
    """Inference over the patch rows of a slide in batches of ONE size: batches run across row boundaries and only the
    very last one is padded (its last patch repeated), so the convolution library sees a single shape per run whatever the
    tissue mask leaves of each row (a per-row tail batch would trigger a solver search per new size).

    ``infer(idx)`` maps an index array of ``batch_size`` patches to a tensor ``[batch_size, ...]`` or a tuple of such
    tensors (one per head).  Yields ``(k, outputs)`` per row ``k`` in order -- ``outputs`` = the row's tensors (tuple if
    ``infer`` returns tuples), ``None`` for a row without patches."""
    flat = [np.asarray(s, dtype=np.int64) for s in row_sels if len(s)]
    flat = np.concatenate(flat) if flat else np.zeros(0, np.int64)
    pos, have, buf, is_tuple = 0, 0, None, False
    for k, sel in enumerate(row_sels):
        need = len(sel)
        if need == 0:
            yield k, None
            continue
        while have < need:
            idx = flat[pos:pos + batch_size]
            real = len(idx)
            if real < batch_size:
                idx = np.concatenate([idx, np.repeat(idx[-1:], batch_size - real)])
            out = infer(idx)
            is_tuple = isinstance(out, tuple)
            out = tuple(o[:real] for o in (out if is_tuple else (out,)))
            buf = out if buf is None else tuple(torch.cat([b, o]) for b, o in zip(buf, out))
            have += real
            pos += real
        row = tuple(b[:need] for b in buf)
        buf = tuple(b[need:] for b in buf) if have > need else None
        have -= need
        yield k, (row if is_tuple else row[0])


class EngineABC:
    """Abstract engine: model + ioconfig + ``run()`` (ref. :136-1885)."""

    def __init__(self, model, batch_size: int = 8, num_workers: int = 0, weights=None, *,
                 device: str = "cpu", verbose: bool = False) -> None:
        self.images = None
        self.masks = None
        self.patch_mode = None
        self.device = device
        self.model, self.ioconfig = self._initialize_model_ioconfig(model=model, weights=weights)
        self.model.to(device=self.device)
        self._ioconfig = self.ioconfig
        self.batch_size = batch_size
        self.labels = None
        self.num_workers = num_workers
        self.patch_input_shape = None
        self.input_resolutions = None
        self.return_labels = False
        self.stride_shape = None
        self.verbose = verbose
        self.dataloader = None
        self.drop_keys: list = []
        self.output_type = None
        # MI355X extensions (plain attributes, settable through run(**kwargs) like every other)
        self.compute_dtype = "float32"    # arithmetic type of the CNN forward
        # float32 3x3 / stride-1 block convolutions: "auto" (default: Winograd F(2x2, 3x3) on every layer whose shape the committed
        # per-layer error-bound test covers -- tests/test_engine.py::test_winograd_conv_matches_torch_cpu_fp32, <= 1e-5 of the
        # output scale against torch-CPU float32 -- the direct implicit GEMM elsewhere), "direct" (audit mode: the reference's
        # order of accumulation over taps everywhere), "winograd" (same layers as "auto"; kept for explicitness)
        self.conv_algo = "auto"
        self.distributed = True           # shard over ranks when torch.distributed is initialised
        self.fold_batchnorm = True        # inference copy with BN folded into the convolutions
        self.miopen_find = None           # True: MIOpen solver search for graphs that still run as plain torch modules
        self._fast_model = None
        self._fast_key = None

    # ------------------------------------------------------------------ model / ioconfig
    @staticmethod
    def _initialize_model_ioconfig(model, weights):
        """Resolve a registry name or take an ``nn.Module`` as is (ref. :338-387)."""
        if not isinstance(model, (str, torch.nn.Module)):
            msg = "Input model must be a string or 'torch.nn.Module'."
            raise TypeError(msg)
        if isinstance(model, str):
            return get_pretrained_model(model, weights)
        if weights is not None:
            model = load_torch_model(model=model, weights=weights)
        return model, None

    def _get_model_attr(self, name: str):
        model = self.model.module if hasattr(self.model, "module") else self.model
        return getattr(model, name)

    def _load_ioconfig(self, ioconfig):
        if self.ioconfig is None and ioconfig is None:
            msg = "Please provide a valid ModelIOConfigABC. No default ModelIOConfigABC found."
            raise ValueError(msg)
        if ioconfig and isinstance(ioconfig, ModelIOConfigABC):
            self.ioconfig = ioconfig
        return self.ioconfig

    def _update_ioconfig(self, ioconfig, patch_input_shape, stride_shape, input_resolutions):
        config_flag = (patch_input_shape is None, input_resolutions is None)
        if isinstance(ioconfig, ModelIOConfigABC):
            return ioconfig
        if self.ioconfig is None and any(config_flag):
            msg = ("Must provide either `ioconfig` or `patch_input_shape` and `input_resolutions`.")
            raise ValueError(msg)
        if stride_shape is None:
            stride_shape = patch_input_shape
        if self.ioconfig:
            cfg = self.ioconfig
            if patch_input_shape is not None:
                cfg.patch_input_shape = patch_input_shape
            if input_resolutions is not None:
                cfg.input_resolutions = input_resolutions
            if stride_shape is not None:
                cfg.stride_shape = stride_shape
            return cfg
        return ModelIOConfigABC(input_resolutions=input_resolutions, patch_input_shape=patch_input_shape,
                                stride_shape=stride_shape, output_resolutions=[])

    # ----------------------------------------------------------------------- validation
    @staticmethod
    def _validate_images_masks(images):
        """ref. :1121-1159"""
        if isinstance(images, torch.Tensor):  # MI355X overload: an NHWC batch already resident in HBM
            if images.dim() != 4:  # noqa: PLR2004
                msg = ("The input numpy array should be four dimensional."
                       "The shape of the numpy array should be NHWC.")
                raise ValueError(msg)
            return images
        if not isinstance(images, (list, np.ndarray)):
            msg = "Input must be a list of file paths or a numpy array."
            raise TypeError(msg)
        if isinstance(images, np.ndarray) and images.ndim != 4:  # noqa: PLR2004
            msg = ("The input numpy array should be four dimensional."
                   "The shape of the numpy array should be NHWC.")
            raise ValueError(msg)
        if isinstance(images, np.ndarray):
            return images
        return [Path(image) if isinstance(image, str) else image for image in images]

    @staticmethod
    def _validate_input_numbers(images, masks=None, labels=None) -> None:
        """ref. :1161-1209"""
        if masks is None and labels is None:
            return
        len_images = len(images)
        if masks is not None and len_images != len(masks):
            msg = f"len(masks) is not equal to len(images) : {len(masks)} != {len(images)}"
            raise ValueError(msg)
        if labels is not None and len_images != len(labels):
            msg = f"len(labels) is not equal to len(images) : {len(labels)} != {len(images)}"
            raise ValueError(msg)

    def _update_run_params(self, images, masks=None, input_resolutions=None, patch_input_shape=None,
                           save_dir=None, ioconfig=None, output_type: str = "dict", *, overwrite: bool = False,
                           patch_mode: bool, **kwargs):
        """ref. :1211-1372: every kwarg becomes an attribute on the engine (and persists)."""
        for key in kwargs:
            setattr(self, key, kwargs.get(key))
        if hasattr(self, "return_probabilities"):
            # decided per call, like the reference (`kwargs.get("return_probabilities")`, patch_predictor.py:535-537,
            # semantic_segmentor.py:795, multi_task_segmentor.py:982): a run without the kwarg drops the probabilities even if an
            # earlier run asked for them
            self.return_probabilities = bool(kwargs.get("return_probabilities", False))
        if "miopen_find" not in kwargs:
            self.miopen_find = None  # the process-global solver-search switch is chosen per run, never inherited
        if input_resolutions:
            self.input_resolutions = input_resolutions
        if patch_input_shape is not None:
            self.patch_input_shape = patch_input_shape
        if not self.return_labels:
            self.drop_keys.append("label")
        self.patch_mode = patch_mode
        self._validate_input_numbers(images=images, masks=masks, labels=self.labels)
        if output_type.lower() not in ["dict", "zarr", "qupath", "annotationstore"]:
            msg = "output_type must be 'dict' or 'zarr', 'qupath' or 'annotationstore'."
            raise TypeError(msg)
        self.output_type = output_type
        if save_dir is not None and output_type.lower() == "dict":
            self.output_type = "zarr"
        if save_dir is None and output_type.lower() in ["zarr", "qupath", "annotationstore"]:
            msg = f"Please provide save_dir for output_type={output_type}"
            raise ValueError(msg)
        if not patch_mode and save_dir is not None and output_type.lower() == "dict":
            self.output_type = "npz"  # WSI mode: one <name>.npz per slide under save_dir (zarr itself is out of scope)
        if self.output_type.lower() not in ("dict", "npz"):
            if save_dir is not None and output_type.lower() == "dict":
                msg = ("`save_dir` turns dict output into a zarr store in the reference (engine_abc.py:1330-1332); "
                       "zarr / annotation-store writers are outside the accelerated hot path: call run() without "
                       "`save_dir` and save the returned dict yourself.")
            else:
                msg = (f"output_type={self.output_type!r} needs zarr / the annotation store, which are outside "
                       "the accelerated hot path; use output_type='dict' without `save_dir`.")
            raise NotImplementedError(msg)
        if not patch_mode and save_dir is None:
            msg = "Input WSIs detected but no save directory provided. Please provide a 'save_dir'."
            raise OSError(msg)
        if not patch_mode and not isinstance(images, (list, tuple)):
            msg = "Input must be a list of file paths or a numpy array."
            raise TypeError(msg)
        self.images = self._validate_images_masks(images=images) if patch_mode else list(images)
        if masks is not None and not patch_mode:
            self.masks = list(masks)
        elif masks is not None:
            self.masks = self._validate_images_masks(images=masks)
        self._ioconfig = self._load_ioconfig(ioconfig=ioconfig)
        self.model = self.model.to(device=self.device)
        self._ioconfig = self._update_ioconfig(ioconfig, self.patch_input_shape, self.stride_shape,
                                               self.input_resolutions)
        return save_dir

    # ------------------------------------------------------------------------ inference
    def get_dataloader(self, images, labels=None, ioconfig=None, **_):
        """Patch mode: a :class:`PatchDataset` carrying the model's ``preproc_func`` (ref. :397-480)."""
        shape = ioconfig.patch_input_shape if ioconfig is not None else None
        ds = PatchDataset(inputs=images, labels=labels, patch_input_shape=shape)
        ds.preproc_func = self._get_model_attr("preproc_func")
        return ds

    def _inference_model(self, dtype: torch.dtype):
        """The module used for the forward pass: parameters in ``dtype``, channels-last (MIOpen NHWC)."""
        algo = str(getattr(self, "conv_algo", None) or "auto")
        if algo not in ("auto", "direct", "winograd"):  # (the same message on every device; on the CPU the option has no effect)
            self.conv_algo = "auto"  # (run kwargs persist as attributes: do not leave the rejected value behind)
            msg = f"conv_algo must be 'auto', 'direct' or 'winograd', got {algo!r}."
            raise ValueError(msg)
        if algo == "auto":  # the modules know two forms; "auto" = Winograd wherever a layer qualifies (their own shape checks)
            algo = "winograd"
        if dtype == torch.float32 and torch.device(self.device).type != "cuda":
            return self.model
        key = (dtype, str(self.device), id(self.model), self.fold_batchnorm, _weights_version(self.model), algo)
        if self._fast_key != key:
            import copy

            on_gpu = torch.device(self.device).type == "cuda"
            m = copy.deepcopy(self.model)
            if self.fold_batchnorm and hasattr(m, "feat_extract"):
                from tiatoolbox_amd.models.architecture.fused import fuse_cnn_model
                from tiatoolbox_amd.models.architecture.resnet import BasicBlock, Bottleneck

                # on the GPU ResNet trunks (BasicBlock and Bottleneck) run on the hand-written kernels only -- stem and block
                # convolutions (architecture/fused.py: MfmaResNet; float32: tia_conv2d_nhwc_f32, fp16 / bf16:
                # tia_conv2d_nhwc_h); every other trunk, and the CPU: BatchNorm folding only
                resnet = any(isinstance(mod, (BasicBlock, Bottleneck)) for mod in m.modules())
                use_mfma = on_gpu and resnet
                m = fuse_cnn_model(m, epilogue_fusion="mfma" if use_mfma else False)
            elif on_gpu and dtype == torch.float32:
                from tiatoolbox_amd.models.architecture.hovernet import HoVerNet
                from tiatoolbox_amd.models.architecture.unet import UNetModel

                if isinstance(m, HoVerNet):
                    # HoVer-Net / HoVerNet+ in float32: 104 of its 144 convolutions on the hand-written MFMA kernel,
                    # BN folded or fused with the ReLU, residual adds in the epilogues (architecture/hovernet_fused.py)
                    from tiatoolbox_amd.models.architecture.hovernet_fused import FusedHoVerNet

                    m = FusedHoVerNet(m.to(device=self.device))
                    from tiatoolbox_amd.models.architecture.hovernet_fused import set_conv_algo

                    set_conv_algo(m, algo)
                elif isinstance(m, UNetModel) and hasattr(m.backbone, "layer1") and m.skip_type == "add":
                    # UNet with the ResNet-50 encoder in float32: 61 of its 63 convolutions on the MFMA kernel
                    from tiatoolbox_amd.models.architecture.unet_fused import FusedUNet

                    m = FusedUNet(m.to(device=self.device))
                    from tiatoolbox_amd.models.architecture.hovernet_fused import set_conv_algo

                    set_conv_algo(m, algo)
            m = m.to(device=self.device)
            if on_gpu:  # hand-written trunks pack their weights (and keep float32 biases) from the float32 parameters
                for mod in m.modules():
                    if type(mod).__name__ == "MfmaResNet":
                        mod.set_conv_algo(algo)  # run kwarg `conv_algo="winograd"`: opt-in float32 Winograd for the 3x3 / stride-1 layers
                        mod.prepare(dtype)
            m = m.to(dtype=dtype) if dtype != torch.float32 else m
            if on_gpu:
                m = m.to(memory_format=torch.channels_last)
            m.eval()
            self._fast_model, self._fast_key = m, key
        return self._fast_model

    @contextlib.contextmanager
    def _miopen_scope(self):
        """Solver search of library convolutions (plain torch modules only: see ``_use_miopen_find``).  ``miopen_find=True`` (run kwarg) lets MIOpen search once per
        convolution shape -- worth it for long runs at one batch shape, costly when batch sizes vary.  The switch is
        PROCESS-GLOBAL in torch (``torch.backends.cudnn.benchmark``): it is set for the run and restored after, so engines
        running concurrently in threads of one process share it.  A ``miopen_find`` passed to one ``run()`` does not persist
        (``_update_run_params`` resets it)."""
        prev = torch.backends.cudnn.benchmark
        torch.backends.cudnn.benchmark = self._use_miopen_find()
        try:
            yield
        finally:
            torch.backends.cudnn.benchmark = prev

    def _use_miopen_find(self) -> bool:
        """Only an explicit ``miopen_find=True`` switches the library's solver search on.  The float32 inference copies (ResNet
        classifiers, ``FusedHoVerNet``, ``FusedUNet``) launch no library convolution at all since round 3; the switch matters
        for what still runs as a plain torch module (half-precision segmentation networks, user-supplied architectures),
        where MIOpen's immediate mode can pick a naive NHWC kernel."""
        return bool(getattr(self, "miopen_find", None))

    def invalidate_inference_cache(self) -> None:
        """Drop the derived (BN-folded / cast) inference copy; it is rebuilt on the next run."""
        self._fast_model, self._fast_key = None, None

    def _device_preproc(self, hook, batch: torch.Tensor, dtype: torch.dtype):
        """The hook's batched device form; when the inference copy's stem reads uint8 (``accepts_uint8``) and the hook ends in
        ``ToTensor``, that step is deferred into the stem kernel (the batch stays uint8, wrapped in ``UnitUInt8``)."""
        if getattr(self, "_defer_unit", False) and getattr(hook, "supports_deferred_unit_scale", False):
            return hook.device_batch(batch, dtype, defer_unit_scale=True)
        return hook.device_batch(batch, dtype)

    def _prenormalized(self, hook, inputs: torch.Tensor, lo: int, hi: int):
        """Stain normalisation of a device-resident patch list in CHUNKS of ``stain_chunk`` patches (default 4096, at most 2 GiB
        of pixels), not per CNN micro-batch: the statistics kernel runs one workgroup per patch, two resident per CU, so a
        1024-patch launch is two rounds with a ragged tail (1.27 ms per 1024 against 2.45 ms per 4096 in one launch,
        ``profiles/r03p_bench_rocprofv3_summary.txt``).  Returns the micro-batch ``[lo, hi)`` of the cached uint8 result
        wrapped for the stem kernel, or ``None`` when the fast path does not apply (then the hook runs per micro-batch)."""
        from tiatoolbox_amd.models.dataset.classification import StainNormPreproc, UnitUInt8

        span = getattr(self, "_shard_span", None)
        if (span is None or not isinstance(hook, StainNormPreproc) or not getattr(self, "_defer_unit", False)
                or not hook.supports_deferred_unit_scale or inputs.dtype != torch.uint8 or not inputs.is_cuda):
            return None
        per_patch = int(inputs[0].numel())
        limit = max(1, min(int(getattr(self, "stain_chunk", 4096)), (2 << 30) // max(per_patch, 1)))
        chunk = max(self.batch_size, limit // self.batch_size * self.batch_size)
        c0 = span[0] + (lo - span[0]) // chunk * chunk
        c1 = min(c0 + chunk, span[1])
        cache = getattr(self, "_norm_cache", None)
        if cache is None or cache[0] != c0 or cache[1] != c1:
            self._norm_cache = cache = (c0, c1, hook.normalizer.transform(inputs[c0:c1], out="uint8"))
        return UnitUInt8(cache[2][lo - c0:hi - c0])

    def _set_defer_unit(self, model, dtype: torch.dtype) -> None:  # noqa: ARG002
        """``ToTensor`` may be deferred into the stem kernel only for the stock classifiers: the inference copy is EXACTLY a
        ``CNNModel`` / ``CNNBackbone`` (not a subclass with its own ``forward`` / ``infer_batch``, whose pre-processing inside
        the model would then see raw bytes) whose trunk is the uint8-reading :class:`MfmaResNet`.  Segmentation models (``FusedUNet``
        reads uint8 through its own ``infer_batch``) never take this path: their ``infer_batch`` also crops and soft-maxes."""
        from tiatoolbox_amd.models.architecture.vanilla import CNNBackbone, CNNModel

        stock = type(model) in (CNNModel, CNNBackbone) and type(self.model) in (CNNModel, CNNBackbone)
        if stock:
            own = type(self.model).__dict__.get("infer_batch")
            bound = getattr(self.model, "infer_batch", None)
            stock = own is not None and getattr(own, "__func__", own) is getattr(bound, "__func__", bound)
        self._defer_unit = bool(torch.device(self.device).type == "cuda" and stock
                                and getattr(getattr(model, "feat_extract", None), "accepts_uint8", False))

    def _forward_batch(self, model, infer_batch, batch):
        from tiatoolbox_amd.models.dataset.classification import UnitUInt8

        if isinstance(batch, UnitUInt8):  # ToTensor deferred: the stem kernel divides by 255 while it loads the bytes
            with torch.inference_mode():
                return model(batch.data.permute(0, 3, 1, 2)).float()
        return infer_batch(model, batch, device=self.device)

    @contextlib.contextmanager
    def _deferred_norm_checks(self, hook):
        """Data-dependent stain-normaliser errors (empty tissue mask, degenerate statistics) are flags in the per-patch
        statistics; inside this scope they are collected on the device and raised ONCE when the run's loop ends, instead of
        one device -> host synchronisation per batch."""
        norm = getattr(hook, "normalizer", None) or getattr(hook, "__self__", None)
        scope = getattr(norm, "deferred_checks", None)
        if scope is None:
            yield
            return
        with scope():
            yield

    def _preprocess_batch(self, dataset: PatchDataset, lo: int, hi: int, dtype: torch.dtype) -> torch.Tensor:
        """Raw patches [lo,hi) -> model-ready NHWC tensor on ``self.device``."""
        dev = torch.device(self.device)
        hook = dataset.preproc_func
        if isinstance(dataset.inputs, torch.Tensor):  # device-resident batch: no host round trip at all
            dataset.check_shape(dataset.inputs.shape[1:])
            t = dataset.inputs[lo:hi]
            if t.device != dev:
                t = t.to(dev)
            device_batch = getattr(hook, "device_batch", None)
            if device_batch is not None:
                if t.device == dataset.inputs.device:
                    pre = self._prenormalized(hook, dataset.inputs, lo, hi)
                    if pre is not None:
                        return pre
                return self._device_preproc(hook, t, dtype)
            from tiatoolbox_amd.models.models_abc import ModelABC as _MA

            if hook is _MA.preproc or hook is PatchDataset.preproc:
                return t
            bound_norm = getattr(hook, "__self__", None)
            from tiatoolbox_amd.tools.stainnorm import StainNormalizer as _SN

            if isinstance(bound_norm, _SN):
                return bound_norm.transform(t).to(dtype)
            items = [torch.as_tensor(np.asarray(hook(p))) for p in t.cpu().numpy()]
            return torch.stack(items).to(dev)
        if isinstance(dataset.inputs, np.ndarray):
            dataset.check_shape(dataset.inputs.shape[1:])
            raw = np.ascontiguousarray(dataset.inputs[lo:hi])
        else:
            raw = np.stack([dataset.raw(i) for i in range(lo, hi)])
        from tiatoolbox_amd.tools.stainnorm import StainNormalizer

        device_batch = getattr(hook, "device_batch", None)
        bound_norm = getattr(hook, "__self__", None)
        from tiatoolbox_amd.models.models_abc import ModelABC

        identity = hook is ModelABC.preproc or hook is PatchDataset.preproc  # the models' default hook: image as is
        if dev.type == "cuda" and raw.dtype == np.uint8 and (
                device_batch is not None or identity or isinstance(bound_norm, StainNormalizer)):
            feed = getattr(self, "_feed", None)
            if feed is not None and feed.array is dataset.inputs:
                t = feed.get(lo, hi)
            else:
                t = torch.from_numpy(raw)
                t = t.pin_memory().to(dev, non_blocking=True) if raw.nbytes > (1 << 20) else t.to(dev)
            if device_batch is not None:
                return self._device_preproc(hook, t, dtype)
            if identity:  # HoVer-Net / UNet scale inside forward(): the uint8 batch goes to infer_batch untouched
                return t
            # bare `model.preproc_func = normalizer.transform`: the reference then feeds 0..255 floats
            return bound_norm.transform(t).to(dtype)
        if device_batch is not None and dev.type != "cuda" and not hasattr(hook, "normalizer"):
            return device_batch(torch.from_numpy(raw), dtype)
        # arbitrary user hook: per patch on the host, exactly like Dataset.__getitem__ in the reference
        items = [torch.as_tensor(np.asarray(hook(p))) for p in raw]
        return torch.stack(items).to(dev)

    def _batch_cuts(self, lo: int, hi: int) -> list[int]:
        """Batch boundaries of this rank's shard ``[lo, hi)``: ``batch_size`` patches each.  With host input on the asynchronous
        feed the FIRST batch's copy is the one transfer nothing can hide (1024 patches of 256 x 256 x 3 are 201 MB: ~8 ms of a 150 ms
        run), so the first ``batch_size`` patches go as batch_size / 8, / 4 and the remaining 5 / 8: 1 ms exposed, and every later copy
        is shorter than the compute of the batch before it.  Per-patch results depend on the batching only within float32 rounding: the
        convolution route (tile shapes, round-fill rule, ring vs slice kernel) is chosen per launch from the number of images, and the
        routes accumulate in different orders -- a ramped host-fed run and a device-resident run of the same data may differ in the
        last ulp (the tests compare them with a tolerance, not bit for bit)."""
        bs = max(int(self.batch_size), 1)
        cuts = [lo]
        feed = getattr(self, "_feed", None)
        if feed is not None and feed.registered and bs >= 64 and hi - lo >= 2 * bs:  # noqa: PLR2004
            cuts += [lo + bs // 8, lo + bs // 8 + bs // 4]
        while cuts[-1] < hi:
            nxt = lo + ((cuts[-1] - lo) // bs + 1) * bs  # the next multiple of the batch size (the ramp ends on the first one)
            cuts.append(min(nxt, hi))
        return cuts

    def infer_patches(self, dataloader: PatchDataset, *, return_coordinates: bool = False) -> dict:
        """Forward every patch; results stay on the device until the end (ref. :505-588)."""
        n = len(dataloader)
        dtype = _DTYPES[str(self.compute_dtype).replace("torch.", "")]
        model = self._inference_model(dtype)
        infer_batch = self._get_model_attr("infer_batch")
        rank, world_size = tdist.world() if self.distributed else (0, 1)
        lo, hi = tdist.shard_bounds(n, rank, world_size)
        outs = []
        self._feed = None
        dev = torch.device(self.device)
        if dev.type == "cuda" and isinstance(dataloader.inputs, np.ndarray) and dataloader.inputs.dtype == np.uint8:
            self._feed = _HostFeed(dataloader.inputs, dev)
        self._set_defer_unit(model, dtype)
        self._shard_span, self._norm_cache = (lo, hi), None
        try:
            with self._miopen_scope(), self._deferred_norm_checks(dataloader.preproc_func):
                cuts = self._batch_cuts(lo, hi)
                for k, (s, e) in enumerate(zip(cuts[:-1], cuts[1:])):
                    if self._feed is not None:  # batch k+1 crosses PCIe while batch k computes
                        self._feed.prefetch(s, e)
                        if k + 2 < len(cuts):
                            self._feed.prefetch(e, cuts[k + 2])
                    with tracing.range("preprocess_batch"):  # stain pre-normalisation kernels (statistics, apply) live here
                        batch = self._preprocess_batch(dataloader, s, e, dtype)
                    with tracing.range("cnn_forward"):
                        outs.append(self._forward_batch(model, infer_batch, batch))
        finally:
            if self._feed is not None:
                self._feed.close()
                self._feed = None
            self._shard_span, self._norm_cache = None, None
        if not outs:  # empty shard: a one-patch probe supplies the row shapes
            probe = self._forward_batch(model, infer_batch, self._preprocess_batch(dataloader, 0, 1, dtype))
            outs = [tuple(p[:0] for p in probe) if isinstance(probe, tuple) else probe[:0]]
        multi_head = isinstance(outs[0], tuple)
        heads = list(zip(*outs)) if multi_head else [outs]
        gathered = []
        # engines that post-process their own shard (instance segmentation) gather results, not raw head maps
        gather_now = world_size > 1 and getattr(self, "gather_raw_predictions", True)
        for chunks in heads:
            chunks = [c if isinstance(c, torch.Tensor) else torch.from_numpy(np.asarray(c)) for c in chunks]
            local = torch.cat(chunks)
            if gather_now:
                if torch.device(self.device).type == "cuda":
                    local = local.to(self.device)
                with tracing.range("all_gather_rows"):
                    local = tdist.all_gather_rows(local, n)
            gathered.append(local)
        local = tuple(gathered) if multi_head else gathered[0]
        raw_predictions = {"probabilities": local}
        if world_size > 1 and not gather_now:
            raw_predictions["shard"] = (lo, hi, n)
        if self.return_labels and dataloader.labels is not None:
            raw_predictions["labels"] = np.asarray(dataloader.labels).reshape(-1)
        if return_coordinates:
            raw_predictions["coordinates"] = np.zeros((n, 4), dtype=np.int64)
        return raw_predictions

    def post_process_patches(self, raw_predictions: dict, **_) -> dict:
        return raw_predictions

    def save_predictions(self, processed_predictions: dict, output_type: str, **_):
        """``dict`` output: computed NumPy arrays, dropped keys removed (ref. :650-767)."""
        out = {}
        for key, val in processed_predictions.items():
            if key in self.drop_keys:
                continue
            out[key] = val.cpu().numpy() if isinstance(val, torch.Tensor) else np.asarray(val)
        return out

    def _run_patch_mode(self, output_type: str, save_dir, **kwargs):
        self.dataloader = self.get_dataloader(images=self.images, labels=self.labels, ioconfig=self._ioconfig)
        raw = self.infer_patches(dataloader=self.dataloader,
                                 return_coordinates=output_type.lower() in ["annotationstore", "qupath"])
        with tracing.range("post_process_patches"):
            processed = self.post_process_patches(raw_predictions=raw, **kwargs)
        return self.save_predictions(processed_predictions=processed, output_type=output_type, **kwargs)

    # ------------------------------------------------------------------------------ WSI mode
    def _open_slide(self, image, *, as_mask: bool = False):
        """In-memory slide: an ``ArrayWSIReader``, an ``H x W (x 3)`` array / tensor, or a ``.npy`` path.  File-format
        readers (OpenSlide, TIFF, ...) are out of scope."""
        from tiatoolbox_amd.wsicore import ArrayWSIReader

        if isinstance(image, ArrayWSIReader):
            return image
        if isinstance(image, (str, Path)):
            path = Path(image)
            if path.suffix != ".npy":
                msg = (f"cannot open `{path}`: whole-slide file formats are outside the accelerated hot path; pass an "
                       "ArrayWSIReader, an array, or a .npy file.")
                raise NotImplementedError(msg)
            image = np.load(path)
        kw = {"mpp": None, "power": None, "mode": "bool"} if as_mask else {}
        if as_mask and not isinstance(image, torch.Tensor):
            image = (np.asarray(image) > 0).astype(np.uint8)
        return ArrayWSIReader(image, **kw)

    @staticmethod
    def _check_read_resolution(reader, ioconfig) -> None:
        """The in-memory reader holds ONE level: the model's input resolution must be that level."""
        res = ioconfig.input_resolutions[0]
        units, value = res["units"], float(res["resolution"])
        native = {"mpp": reader.mpp, "power": reader.power, "baseline": 1.0}.get(units)
        if native is not None and np.ndim(native) > 0:  # (mpp_x, mpp_y): both components must be the requested value
            comps = [float(v) for v in np.asarray(native, dtype=np.float64).ravel()]
            native = comps[0] if all(abs(c - comps[0]) <= 1e-6 * max(1.0, abs(comps[0])) for c in comps) else tuple(comps)
        if native is None or isinstance(native, tuple) or abs(float(native) - value) > 1e-6 * max(1.0, abs(value)):  # noqa: PLR2004
            msg = (f"the model reads at {value} {units} but the in-memory slide is at {native} {units}: ArrayWSIReader "
                   "has no resolution pyramid; resample the slide first, wrap the array as "
                   f"`ArrayWSIReader(array, {units}={value})` if that is its true resolution, or pass a matching "
                   "`input_resolutions`.")
            raise ValueError(msg)

    def get_wsi_coordinates(self, reader, mask_reader, *, min_mask_ratio: float = 0.0) -> np.ndarray:
        """Patch grid of one slide, filtered by the tissue mask (``WSIPatchDataset.__init__``, dataset_abc.py:309-345)."""
        from tiatoolbox_amd.tools.patchextraction import PatchExtractor

        cfg = self._ioconfig
        wsi_shape = reader.slide_dimensions
        coords = PatchExtractor.get_coordinates(image_shape=wsi_shape, patch_input_shape=tuple(cfg.patch_input_shape)[::-1],
                                                stride_shape=tuple(cfg.stride_shape)[::-1])
        if mask_reader is not None:
            keep = PatchExtractor.filter_coordinates(mask_reader, coords, wsi_shape=wsi_shape, min_mask_ratio=min_mask_ratio)
            coords = coords[keep]
        if len(coords) == 0:
            msg = "No patch coordinates remain after filtering."
            raise ValueError(msg)
        return coords

    def infer_wsi(self, reader, coords: np.ndarray) -> dict:
        """Forward every patch of one slide: device gather of the patch batch (``tia_gather_patches_u8``) -> the model's
        ``preproc_func`` in its batched device form -> ``infer_batch``; patches sharded over ranks (ref. :590-648)."""
        dev = torch.device(self.device)
        if dev.type != "cuda":
            msg = "WSI mode reads patches with a HIP gather kernel (device='cuda'); there is no CPU fallback."
            from tiatoolbox_amd import _lib

            raise _lib.HipLibraryError(msg)
        dtype = _DTYPES[str(self.compute_dtype).replace("torch.", "")]
        model = self._inference_model(dtype)
        infer_batch = self._get_model_attr("infer_batch")
        hook = self._get_model_attr("preproc_func")
        device_batch = getattr(hook, "device_batch", None)
        n = len(coords)
        rank, world_size = tdist.world() if self.distributed else (0, 1)
        lo, hi = tdist.shard_bounds(n, rank, world_size)
        outs = []
        self._set_defer_unit(model, dtype)
        # the shard's patch bounds go to the device once (a per-batch upload is a synchronous copy: the host would wait for the
        # previous batch's kernels before it can launch the next forward)
        c_np = np.ascontiguousarray(np.asarray(coords[lo:hi]).reshape(-1, 4), dtype=np.int32)
        size = (int(c_np[0, 2] - c_np[0, 0]), int(c_np[0, 3] - c_np[0, 1])) if len(c_np) else (0, 0)
        uniform = len(c_np) > 0 and bool(np.all(c_np[:, 2] - c_np[:, 0] == size[0]) and np.all(c_np[:, 3] - c_np[:, 1] == size[1]))
        coords_dev = torch.from_numpy(c_np).to(dev) if uniform else None
        with self._miopen_scope(), self._deferred_norm_checks(hook):
            for s in range(lo, hi, self.batch_size):
                e = min(s + self.batch_size, hi)
                raw = (reader.read_bounds_batch(coords_dev[s - lo:e - lo], size=size) if coords_dev is not None
                       else reader.read_bounds_batch(coords[s:e]))
                if device_batch is not None:
                    batch = self._device_preproc(hook, raw, dtype)
                else:  # arbitrary user hook: per patch on the host, as Dataset.__getitem__ does in the reference
                    batch = torch.stack([torch.as_tensor(np.asarray(hook(p))) for p in raw.cpu().numpy()]).to(dev)
                outs.append(self._forward_batch(model, infer_batch, batch))
        if not outs:
            probe = reader.read_bounds_batch(coords[:1])
            probe = self._device_preproc(hook, probe, dtype) if device_batch is not None else probe
            outs = [self._forward_batch(model, infer_batch, probe)[:0]]
        local = torch.cat([o if isinstance(o, torch.Tensor) else torch.from_numpy(np.asarray(o)) for o in outs])
        if world_size > 1:
            local = tdist.all_gather_rows(local.to(dev), n)
        return {"probabilities": local, "coordinates": coords}

    def _run_wsi_mode(self, save_dir, **kwargs) -> dict:
        """Per slide: tissue mask -> patch grid -> ``infer_wsi`` -> ``post_process_patches`` -> ``<stem>.npz`` under
        ``save_dir`` (ref. ``_run_wsi_mode`` :1540-1682; arrays ``predictions``, ``coordinates`` and, on request,
        ``probabilities`` -- the members of the reference's zarr store).  Returns ``{image key: Path}``."""
        save_dir = prepare_engines_save_dir(save_dir, patch_mode=False, overwrite=bool(kwargs.pop("overwrite", False)),
                                            distributed=self.distributed)
        out: dict = {}
        images = self.images if isinstance(self.images, (list, tuple)) else [self.images]
        for num, image in enumerate(images):
            reader = self._open_slide(image)
            self._check_read_resolution(reader, self._ioconfig)
            mask_reader = None
            if self.masks is not None:
                mask_reader = self._open_slide(self.masks[num], as_mask=True)
            elif kwargs.get("auto_get_mask", getattr(self, "auto_get_mask", True)):
                mask_reader = reader.tissue_mask(resolution=1.25, units="power")
            coords = self.get_wsi_coordinates(reader, mask_reader, min_mask_ratio=float(kwargs.get("min_mask_ratio", 0.0)))
            raw = self.infer_wsi(reader, coords)
            processed = self.post_process_patches(raw_predictions=raw, **kwargs)
            arrays = self.save_predictions(processed_predictions=processed, output_type="dict", **kwargs)
            key = image if isinstance(image, (str, Path)) else num
            stem = Path(image).stem if isinstance(image, (str, Path)) else str(num)
            path = save_dir / f"{stem}.npz"
            write_outputs(self.distributed, lambda path=path, arrays=arrays: np.savez(path, **arrays))
            out[key] = path
        return out

    def run(self, images, *, masks=None, input_resolutions=None, patch_input_shape=None, ioconfig=None,
            patch_mode: bool = True, save_dir=None, overwrite: bool = False, output_type: str = "dict", **kwargs):
        """Run the engine on patches (ref. :1684-1829)."""
        save_dir = self._update_run_params(
            images=images, masks=masks, input_resolutions=input_resolutions, patch_input_shape=patch_input_shape,
            save_dir=save_dir, ioconfig=ioconfig, overwrite=overwrite, patch_mode=patch_mode,
            output_type=output_type, **kwargs)
        if patch_mode:
            return self._run_patch_mode(output_type=self.output_type, save_dir=save_dir, **kwargs)
        return self._run_wsi_mode(save_dir=save_dir, overwrite=overwrite, **kwargs)

    predict = run  # tiatoolbox 1.x name, still used by the reference's example notebooks
