"""Exception types with the reference's names (``tiatoolbox/utils/exceptions.py``)."""

from __future__ import annotations


class FileNotSupportedError(Exception):
    """Unsupported file format (reference ``utils/exceptions.py:6-19``)."""

    def __init__(self, message: str = "File format is not supported") -> None:
        super().__init__(message)


class MethodNotSupportedError(Exception):
    """Unsupported method (reference ``utils/exceptions.py:22-35``)."""

    def __init__(self, message: str = "Method is not supported") -> None:
        super().__init__(message)


class DimensionMismatchError(Exception):
    """Shape mismatch (reference ``utils/exceptions.py:38-55``)."""

    def __init__(self, expected_dims, actual_dims) -> None:
        self.expected_dims = expected_dims
        self.actual_dims = actual_dims
        super().__init__(f"Expected dimensions {expected_dims}, but got {actual_dims}.")
