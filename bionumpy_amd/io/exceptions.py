"""bionumpy/io/exceptions.py:1-9"""
from ..exceptions import ParsingException, FormatException, IncompleteEntryException  # noqa: F401
