"""NumpyFileReader: the chunk loop between a file object and the device buffers
(bionumpy/io/parser.py:36-206), statement by statement.

Host side: ``file.read(min_chunk_size)`` into a uint8 array, append '\\n' at EOF, carry the
unconsumed tail (``seek`` back on plain files, ``_prepend`` on gzip streams).  Device side: the
buffer class's ``from_raw_buffer`` uploads the chunk and runs the newline scan / validation kernels,
and reports how many bytes the complete entries cover (``buff.size``).
"""
import numpy as np

from ..exceptions import FormatException
from ..ops import get_ops


class NumpyFileReader:
    def __init__(self, file_obj, buffer_type, has_header=False):
        self._file_obj = file_obj
        self._is_finished = False
        self._buffer_type = buffer_type
        self._has_header = has_header
        self._f_name = self._file_obj.name if hasattr(self._file_obj, "name") else str(self._file_obj)
        self._header_data = self._buffer_type.read_header(self._file_obj)
        self._buffer_type = self._buffer_type.modify_class_with_header_data(self._header_data)
        self._do_prepend = False
        self._prepend = []
        self.n_bytes_read = 0
        self.n_lines_read = 0

    def __enter__(self):
        return self

    def __exit__(self, *args):
        self._file_obj.close()

    def __iter__(self):
        return self.read_chunks()

    def set_prepend_mode(self):
        self._do_prepend = True

    def close(self):
        self._file_obj.close()

    def read(self):
        # parser.py:89-94
        chunk = np.frombuffer(self._file_obj.read(), dtype=np.uint8)
        if chunk.size == 0:
            return None
        chunk, _ = self.__add_newline_to_end(chunk, chunk.size)
        return self._buffer_type.from_raw_buffer(chunk, header_data=self._header_data)

    def read_chunk(self, min_chunk_size=5000000, max_chunk_size=None):
        # parser.py:96-171
        complete_entry_found = False
        temp_chunks = []
        if len(self._prepend):
            temp_chunks.append(self._prepend)
        made_buffer = None
        chunk = None
        while not complete_entry_found:
            # chunks that must outlive the next read cannot stay views of a (recycled) pinned staging buffer
            temp_chunks = [self._detach(c) for c in temp_chunks]
            chunk = self._get_buffer(min_chunk_size, max_chunk_size)
            if chunk is None:
                return None
            temp_chunks.append(chunk)
            if max_chunk_size is not None and sum(c.size for c in temp_chunks) > max_chunk_size:
                raise Exception("No complete entry found")
            try:
                complete_entry_found = self._buffer_type.contains_complete_entry(temp_chunks)
            except FormatException as e:
                e.line_number += self.n_lines_read
                raise e
            if isinstance(complete_entry_found, tuple):
                complete_entry_found, made_buffer = complete_entry_found
        if made_buffer is not None:
            buff = made_buffer
        else:
            chunk = temp_chunks[0] if len(temp_chunks) == 1 else np.concatenate(temp_chunks)
            try:
                buff = self._buffer_type.from_raw_buffer(chunk, header_data=self._header_data)
            except FormatException as e:
                e.line_number += self.n_lines_read
                raise e
        self._prepend = []
        if not self._is_finished:
            if not self._do_prepend:
                self._file_obj.seek(buff.size - chunk.size, 1)
            else:
                self._prepend = self._detach(chunk[buff.size:])
        if chunk is not None and chunk.size:
            self.n_bytes_read += buff.size
            self.n_lines_read += buff.n_lines
            return buff

    def read_chunks(self, min_chunk_size=5000000, max_chunk_size=None):
        while not self._is_finished:
            chunk = self.read_chunk(min_chunk_size, max_chunk_size)
            if chunk is None:
                break
            yield chunk

    def __add_newline_to_end(self, chunk, bytes_read):
        # parser.py:183-190
        if chunk[bytes_read - 1] != ord("\n"):
            chunk = np.append(chunk, np.uint8(ord("\n")))
            bytes_read += 1
        if hasattr(self._buffer_type, "_new_entry_marker"):
            chunk = np.append(chunk, np.uint8(ord(self._buffer_type._new_entry_marker)))
            bytes_read += 1
        return chunk, bytes_read

    def _get_buffer(self, min_chunk_size=5000000, max_chunk_size=None):
        # parser.py:192-206; on the GPU the bytes land in a pinned staging buffer (io/pinned.py) so that the
        # upload in from_raw_buffer is one hipMemcpyAsync out of page-locked memory
        if hasattr(self._file_obj, "readinto") and not getattr(get_ops(), "host_only", False):
            return self._get_pinned_buffer(min_chunk_size)
        a = np.frombuffer(self._file_obj.read(min_chunk_size), dtype="uint8")
        bytes_read = a.size
        self._is_finished = bytes_read < min_chunk_size
        if bytes_read == 0:
            return None
        if self._is_finished:
            a, bytes_read = self.__add_newline_to_end(a, bytes_read)
        return a[:bytes_read]

    @staticmethod
    def _detach(chunk):
        from .pinned import pool
        if isinstance(chunk, np.ndarray) and pool().owner_of(chunk) is not None:
            return chunk.copy()
        return chunk

    def _get_pinned_buffer(self, min_chunk_size):
        from .pinned import read_into_pinned
        a, buf = read_into_pinned(self._file_obj, min_chunk_size, headroom=2)
        bytes_read = a.size
        self._is_finished = bytes_read < min_chunk_size
        if bytes_read == 0:
            return None
        if self._is_finished:                       # same rule as __add_newline_to_end, written in place
            if buf.array[bytes_read - 1] != ord("\n"):
                buf.array[bytes_read] = ord("\n")
                bytes_read += 1
            if hasattr(self._buffer_type, "_new_entry_marker"):
                buf.array[bytes_read] = ord(self._buffer_type._new_entry_marker)
                bytes_read += 1
        return buf.array[:bytes_read]
