"""bionumpy/encodings/exceptions.py:1-4"""
from ..exceptions import EncodingError  # noqa: F401
