"""Position weight matrix scores on the MI355X path (bionumpy/sequence/position_weight_matrix.py).

``PWM`` keeps the reference's constructor and factories (log-likelihood-ratio matrix, ``matrix[letter][position]``);
``get_motif_scores(sequence, pwm)`` returns a ragged float64 array with one score per window of ``pwm.window_size``
bases of every row.  The scores are accumulated in double precision in the reference's order (offset 0, 1, ...), on
the packed 2-bit form (``bnpk_pwm_scores``).
"""
import numpy as np

from ..encoded_array import (EncodedArray, EncodedRaggedArray, AlphabetEncoding, as_encoded_array, packed_words)
from ..ops import get_ops
from ..ragged import RaggedArray


def _pwm_from_counts(count_matrix):
    with_pseudo = count_matrix + 1                                            # position_weight_matrix.py:29-31
    return np.log(with_pseudo / with_pseudo.sum(axis=0, keepdims=True))


class PWM:
    """position_weight_matrix.py:34-148"""

    def __init__(self, matrix, alphabet):
        self._matrix = np.asarray(matrix, dtype=float)
        self._alphabet = alphabet
        self._encoding = AlphabetEncoding(alphabet)

    @property
    def alphabet(self):
        return self._alphabet

    @property
    def window_size(self):
        return self._matrix.shape[-1]

    def as_valid_encoded_array(self, sequence):
        if isinstance(sequence, (EncodedArray, EncodedRaggedArray)) and isinstance(sequence.encoding, AlphabetEncoding):
            alphabet = list(sequence.encoding.get_alphabet())
            if alphabet[:len(self._alphabet)] != list(self._alphabet):
                raise Exception("Could not calculate pwm for alphabet %s on %s encoded array"
                                % (list(self._alphabet), alphabet))
            return sequence
        return as_encoded_array(sequence, self._encoding)

    def _scores(self, sequence):
        sequence = self.as_valid_encoded_array(sequence)
        if "".join(self._alphabet).upper() != "ACGT":
            raise NotImplementedError("motif scores on the MI355X path: the ACGT alphabet (2-bit codes)")
        if self.window_size > 64:
            raise NotImplementedError("motifs longer than 64 positions are not on the MI355X path")
        single = isinstance(sequence, EncodedArray)
        ragged = EncodedRaggedArray(sequence.ravel(), [sequence.size]) if single else sequence
        ragged._compact()
        ops = get_ops()
        n_rows, total = len(ragged), ragged.total()
        out_off, n_out = ops.row_offsets(ragged._lens, self.window_size)
        scores = ops.pwm_scores(packed_words(ragged._data), ragged.offsets(), n_rows, total, n_out, self._matrix)
        self._last_offsets = out_off
        return scores, ragged, single

    def calculate_score(self, sequence):
        """score of one window of exactly window_size symbols (position_weight_matrix.py:67-80)"""
        sequence = self.as_valid_encoded_array(sequence)
        assert sequence.shape[-1] == self.window_size
        scores, _, _ = self._scores(sequence)
        return float(scores.host()[0])

    @classmethod
    def from_dict(cls, dictionary, background=None):
        """position probabilities -> log-likelihood ratios (position_weight_matrix.py:106-134)"""
        if background is None:
            background = {key: 1 / len(dictionary) for key in dictionary}
        alphabet = "".join(dictionary.keys())
        with np.errstate(divide="ignore"):
            matrix = np.log(np.array(list(dictionary.values()), dtype=float)) - \
                np.log([background[key] for key in dictionary])[:, np.newaxis]
        return cls(matrix, alphabet)

    @classmethod
    def from_counts(cls, counts):
        return cls(_pwm_from_counts(np.array(list(counts.values()))), "".join(counts.keys()))

    def __str__(self):
        matrix = self._matrix.transpose()
        return "PWM with alphabet " + self._alphabet + "\n" + \
            "\n".join(" ".join(str(round(c, 2)) for c in row) for row in matrix)


def get_motif_scores(sequence, pwm):
    """motif score at every position of every read (position_weight_matrix.py:177-196): a ragged float64 array,
    row r holds max(0, L_r - window_size + 1) scores"""
    scores, ragged, single = pwm._scores(as_encoded_array(sequence) if not isinstance(
        sequence, (EncodedArray, EncodedRaggedArray)) else sequence)
    if single:
        return scores.host()
    # the scores stay in HBM (a float64 per window: 56 GB for 50 M reads would not cross PCIe in the time of a thousand such
    # kernels); ``.max(axis=-1)`` / ``.sum(axis=-1)`` reduce them there (position_weight_matrix.py:177-196 returns a ragged array)
    from .kmers import _LazyLens
    out_off = pwm._last_offsets
    lens = _LazyLens(out_off) if pwm.window_size > 1 else ragged._lens
    return RaggedArray._from_parts(scores, None, lens, out_off, len(ragged), scores.size)


class PositionWeightMatrix:
    """PositionWeightMatrix (sequence/position_weight_matrix.py:13-25): the rollable form of a PWM — a call scores one window
    of window_size symbols, ``rolling_window`` every window of every row (== get_motif_scores)."""

    def __init__(self, pwm):
        self._pwm = pwm
        self._encoding = pwm._encoding
        self.window_size = pwm.window_size

    def __call__(self, sequence):
        return self._pwm.calculate_score(sequence)

    def rolling_window(self, sequence):
        return get_motif_scores(sequence, self._pwm)

