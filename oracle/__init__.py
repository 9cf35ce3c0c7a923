"""CPU oracle for the tiatoolbox per-patch hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it, and only as the *checker* (or as the reported CPU baseline), never as the
thing measured or shipped.  The product path (``tiatoolbox_amd``) must never import
this package and fails loudly when its HIP library is missing.

What it is: a NumPy/SciPy restatement of the reference's (tiatoolbox v2.0.1,
``/root/reference``) algorithm for the hot path, each function citing the reference
``file:line`` it follows.  The reference itself is pure Python but cannot be imported
in this environment (``cv2``, ``skimage``, ``torchvision``, ``dask``, ... are absent),
so third-party primitives (OpenCV 8-bit colour conversion, scikit-image
``rescale_intensity``/``threshold_otsu``/``watershed`` ...) are restated from their
published algorithms in ``cvref.py`` / ``skref.py``.

Pinning status
--------------
* pinned to the reference's own offline golden vectors / known answers
  (``tests/test_oracle_golden.py``): ``contrast_enhancer`` 27-value golden
  (reference ``tests/test_utils.py:882-911``), extractor helper truth tables
  (``tests/test_stainnorm.py:16-68``), morphological masker 10x10 known answer
  (``tests/test_tissuemask.py:186-212``), canvas-merge known answers
  (``tests/engines/test_semantic_segmentor.py:283-362``).
* pinned to the reference's *Python-level* logic by executing the real reference
  modules from ``/root/reference`` with only the absent third-party primitives
  shimmed (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``).
  Fixtures: stain (Macenko / Vahadane / Ruifrok / Custom / augment), maskers, HoVer-Net
  ``_proc_np_hv`` + ``get_instance_info`` (incl. contour polygons), patch grids + canvas merges,
  Reinhard, and the WSI tile-mode merge of instance predictions (tile sets, margin rules, id
  stitching: the reference's own functions driven like ``_process_tile_mode`` drives them).
* the OpenCV / scikit-image / shapely primitives themselves (``cv2.cvtColor`` RGB<->LAB 8-bit,
  ``cv2.Sobel``, ``cv2.findContours`` [``cvref.first_contour``: Suzuki-Abe border following with
  structural known answers], ``skimage.segmentation.watershed``, shapely ``box`` / ``STRtree``
  [``geomref.py``] ...) are **parity unpinned** against the real libraries: they are not
  installable here.  Every constant in ``cvref.py`` says where it comes from.
"""
