from .buffers import FastQBuffer, TwoLineFastaBuffer, MultiLineFastaBuffer, OneLineBuffer, FileBuffer
from .exceptions import FormatException, ParsingException, IncompleteEntryException
from .files import bnp_open, count_entries
from .npdataclassreader import NpDataclassReader
from .parser import NumpyFileReader

__all__ = ["FastQBuffer", "TwoLineFastaBuffer", "MultiLineFastaBuffer", "OneLineBuffer", "FileBuffer",
           "FormatException", "ParsingException", "IncompleteEntryException", "bnp_open", "count_entries",
           "NpDataclassReader", "NumpyFileReader"]
