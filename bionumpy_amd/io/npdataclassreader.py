"""NpDataclassReader: what ``bnp.open`` returns (bionumpy/io/npdataclassreader.py:14-142)."""
from ..exceptions import FormatException
from ..streams import NpDataclassStream


class NpDataclassReader:
    def __init__(self, numpyfilereader, lazy=None, shard=None):
        self._reader = numpyfilereader
        self._lazy = lazy
        self._shard = shard                # io.sharding.Shard: this reader reads one rank's part of the file

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_value, traceback):
        self._reader.close()

    def close(self):
        self._reader.close()

    def _wrap(self, chunk, n_lines_read=0):
        if self._should_be_lazy(chunk):
            return chunk.dataclass._lazy(chunk, n_lines_read)
        return chunk.get_data()

    def read(self):
        """the whole file as one chunk object (npdataclassreader.py:36-58); of a sharded reader: the rank's part"""
        if getattr(self._reader, "_chunk_modulo", None) is not None:
            # the shard of a gzip stream is every n-th chunk (io/sharding.py): its entries are those chunks, joined
            import numpy as np
            chunks = list(self.read_chunks())
            if not chunks:
                return self._reader._buffer_type.dataclass.empty()
            return chunks[0] if len(chunks) == 1 else np.concatenate(chunks)
        chunk = self._reader.read()
        if chunk is None:
            return self._reader._buffer_type.dataclass.empty()
        return self._wrap(chunk)

    def read_chunk(self, min_chunk_size=5000000, max_chunk_size=None):
        """all complete entries of the next >= min_chunk_size bytes (npdataclassreader.py:60-92)"""
        chunk = self._reader.read_chunk(min_chunk_size, max_chunk_size)
        if chunk is None:
            return self._reader._buffer_type.dataclass.empty()
        n_lines_read = self._reader.lines_before_chunk(chunk.n_lines)     # (behind the read: a shard may have skipped chunks)
        try:
            return self._wrap(chunk, n_lines_read)
        except FormatException as e:
            e.line_number += n_lines_read
            raise e

    def read_chunks(self, min_chunk_size=5000000, max_chunk_size=None):
        def chunks():                                  # the file reader's own generator: it reads ahead for big batches
            for chunk in self._reader.read_chunks(min_chunk_size, max_chunk_size):
                n_lines_read = self._reader.lines_before_chunk(chunk.n_lines)
                try:
                    wrapped = self._wrap(chunk, n_lines_read)
                except FormatException as e:
                    e.line_number += n_lines_read
                    raise e
                if len(wrapped) == 0:
                    return
                yield wrapped
        def again(bigger):                                # (streams.NpDataclassStream._coalesced: before anything was read)
            return self.read_chunks(max(bigger, min_chunk_size), max_chunk_size)._stream
        return NpDataclassStream(chunks(), dataclass=self._reader._buffer_type.dataclass,
                                 rebatch=again if max_chunk_size is None else None, shard=self._shard)

    def __iter__(self):
        return self.read_chunks()

    def _should_be_lazy(self, chunk):
        # npdataclassreader.py:135-142 with config.LAZY = True; multi-line FASTA has SKIP_LAZY
        if self._lazy is False:
            return False
        return hasattr(chunk, "get_field_by_number") and not getattr(chunk, "SKIP_LAZY", False)
