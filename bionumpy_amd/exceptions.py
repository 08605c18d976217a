"""Exceptions of the sequence path, same names and fields as the reference:
bionumpy/io/exceptions.py:1-9 (ParsingException, FormatException),
bionumpy/io/file_buffers.py:272-273 (IncompleteEntryException),
bionumpy/encodings/exceptions.py:1-4 (EncodingError)."""


class ParsingException(Exception):
    pass


class FormatException(ParsingException):
    def __init__(self, message, byte_position=None, line_number=None, offending_text=None):
        super().__init__(message)
        self.byte_position = byte_position
        self.line_number = line_number
        self.offending_text = offending_text


class IncompleteEntryException(Exception):
    pass


class NoCompleteEntry(RuntimeError, IncompleteEntryException):
    """MultiLineFastaBuffer.from_raw_buffer found no complete entry (a RuntimeError in the reference,
    multiline_buffer.py:95-96); the reader reads on when it sees it"""


class EncodingError(Exception):
    def __init__(self, message, offset=0):
        super().__init__(message)
        self.message = message
        self.offset = offset
