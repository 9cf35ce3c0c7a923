"""Deprecated import location kept by the reference (``nucleus_instance_segmentor.py:18-174``)."""

from tiatoolbox_amd.models.engine.multi_task_segmentor import NucleusInstanceSegmentor

__all__ = ["NucleusInstanceSegmentor"]
