"""The reference's own documented examples, run against this package under its name.

``tests/golden/reference_doctests.json`` holds the ``>>>`` statements and printed answers of the reference's
documentation for the sequence path (made by ``tests/golden/make_reference_doctests.py`` from /root/reference:
docs_source/topics/kmers.rst, source/encoding.rst, source/reading_files.rst, source/sequences.rst, README.rst and the
docstrings of io/files.py, sequence/kmers.py, minimizers.py, string_matcher.py, position_weight_matrix.py,
streams/decorators.py, encoded_array.py, bnpdataclassfunction.py).  Here ``import bionumpy`` resolves to ``bionumpy_amd`` and
every group runs through ``doctest`` on both backends: the host logic over the oracle on CPU, the HIP kernels under
``-m gpu``.  A user script written against the reference must print what the reference's documentation prints.

What is not run is listed: SKIP names every example outside SURVEY §8 (other formats, genomic_data, simulate,
open_indexed …) with its reason; ``out_of_scope`` in the JSON names the files that were not extracted at all.
The two example scripts the reference benchmarks with (scripts/kmer_counting_example.py,
scripts/fastq_filtering_example.py) are run from /root/reference where that exists (this container) — their text is
not a vector and is not kept here; on the GPU box their equivalents are tests/test_api.py's stream tests.
"""
import doctest
import importlib
import json
import os
import sys

import pytest

from backends import bnp  # noqa: F401  (fixture: hostlogic on CPU, hip under -m gpu)

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "reference_doctests.json")) as _f:
    VECTORS = json.load(_f)
GROUPS = VECTORS["groups"]

# (group name, line of the example in the reference) -> why it is not run; a whole group: line None
_RST = "docs_source/source/"
SKIP = {
    (_RST + "reading_files.rst", 13): "reads a .bed file (intervals: SURVEY §2 rows 20-28)",
    (_RST + "reading_files.rst", 14): "prints the .bed table",
    (_RST + "sequences.rst", 52): "bionumpy.simulate is outside §8; REPLACED below by the sequences the document prints",
    (_RST + "sequences.rst", 53): "rng of the simulation",
    (_RST + "sequences.rst", 54): "rng of the simulation",
    (_RST + "sequences.rst", 55): "alphabet of the simulation (bnp.encodings.alphabet_encoding.AminoAcidEncoding.get_labels)",
    (_RST + "sequences.rst", 56): "simulate_sequences(...)",
    (_RST + "sequences.rst", 181): "bnp.open_indexed (.fai random access) is outside §8",
    (_RST + "sequences.rst", 182): "bnp.open_indexed",
}

# statements run in place of a skipped example: what the skipped lines would have produced, taken from what the document
# itself prints a few lines further down (sequences.rst:58-71 lists the ten simulated sequences) — so that the twelve
# examples behind them, which exercise EncodedRaggedArray (§8 T2), do run
REPLACE = {
    (_RST + "sequences.rst", 56): (
        "named_seqs = bnp.SequenceEntry(['s%d' % i for i in range(10)], bnp.as_encoded_array("
        "['LMSYAEVYGH', 'WKGVGKQNCAWSVNVH', 'LTDHDL*DKKWFMGASC', 'GMMD*S*CSHNYG', 'SEH*KMHDKQLTIP', 'TYKASNWLICLQTVFP', "
        "'TGIVPMRM*S', 'CENVC', 'RSTWF', 'NTIFMC'], bnp.AminoAcidEncoding))\n"),
}

_SUBMODULES = ["io", "io.files", "io.parser", "io.buffers", "io.exceptions", "io.gzip_reading", "io.npdataclassreader",
               "sequence", "sequence.kmers", "sequence.minimizers", "sequence.count_encoded", "sequence.string_matcher",
               "sequence.position_weight_matrix", "sequence.dna", "sequence.debruin", "sequence.indexing",
               "sequence.indexing.kmer_indexing", "encodings", "encodings.kmer_encodings", "encodings.exceptions",
               "streams", "encoded_array", "datatypes", "exceptions", "memory_mapping"]


# modules of the reference whose contents live under another name here
_MODULE_ALIASES = {"bnpdataclass.bnpdataclassfunction": "datatypes", "streams.decorators": "streams"}


@pytest.fixture
def as_bionumpy(bnp, tmp_path, monkeypatch):  # noqa: F811
    """``import bionumpy`` == this package; the examples' relative ``example_data/…`` paths resolve to the fixtures"""
    saved = {k: v for k, v in sys.modules.items() if k == "bionumpy" or k.startswith("bionumpy.")}
    sys.modules["bionumpy"] = bnp
    for sub in _SUBMODULES:
        sys.modules["bionumpy." + sub] = importlib.import_module("bionumpy_amd." + sub)
    os.symlink(os.path.join(HERE, "golden"), tmp_path / "example_data")
    monkeypatch.chdir(tmp_path)
    yield bnp
    for k in [k for k in sys.modules if k == "bionumpy" or k.startswith("bionumpy.")]:
        del sys.modules[k]
    sys.modules.update(saved)


def _run_group(group):
    examples, skipped = [], []
    for e in group["examples"]:
        why = SKIP.get((group["name"], e["lineno"]), SKIP.get((group["name"], None)))
        if why is not None:
            skipped.append((e["lineno"], why))
            if (group["name"], e["lineno"]) in REPLACE:
                examples.append(doctest.Example(REPLACE[(group["name"], e["lineno"])], "", lineno=e["lineno"] - 1))
            continue
        options = {doctest.OPTIONFLAGS_BY_NAME[k]: v for k, v in e.get("options", {}).items()}
        examples.append(doctest.Example(e["source"], e["want"], lineno=e["lineno"] - 1, options=options))
    globs = {}
    if group.get("module"):                 # the examples of a docstring run in their module's namespace, as doctest runs them
        globs.update(vars(importlib.import_module("bionumpy_amd." + _MODULE_ALIASES.get(group["module"], group["module"]))))
    test = doctest.DocTest(examples, globs, group["name"], group["file"], 0, None)
    failures = []

    class Runner(doctest.DocTestRunner):
        def report_failure(self, out, test, example, got):
            failures.append("%s:%d\n>>> %swant:\n%sgot:\n%s" % (group["file"], example.lineno + 1, example.source, example.want, got))

        def report_unexpected_exception(self, out, test, example, exc_info):
            import traceback
            failures.append("%s:%d\n>>> %sraised:\n%s" % (group["file"], example.lineno + 1, example.source,
                                                           "".join(traceback.format_exception(*exc_info)[-6:])))

    Runner(verbose=False, optionflags=doctest.NORMALIZE_WHITESPACE).run(test, out=lambda s: None, clear_globs=True)
    return failures, skipped, len(examples)


@pytest.mark.parametrize("group", GROUPS, ids=[g["name"].replace("bionumpy/", "").replace("docs_source/", "") for g in GROUPS])
def test_reference_examples(as_bionumpy, group):
    failures, skipped, n_run = _run_group(group)
    assert not failures, "\n\n".join(failures)
    assert n_run or skipped


def test_every_skip_names_an_example():
    """a SKIP entry that matches nothing is a stale excuse"""
    known = {(g["name"], e["lineno"]) for g in GROUPS for e in g["examples"]} | {(g["name"], None) for g in GROUPS}
    assert set(SKIP) <= known, set(SKIP) - known


REF_SCRIPTS = "/root/reference/scripts"


@pytest.mark.skipif(not os.path.isdir(REF_SCRIPTS), reason="the reference's scripts are only in the build container")
@pytest.mark.parametrize("script", ["kmer_counting_example.py", "fastq_filtering_example.py"])
def test_reference_example_scripts(as_bionumpy, script, tmp_path):
    """the reference's benchmark scripts, unmodified, against this package (their ``test()`` is what its CI runs)"""
    import runpy
    mod = runpy.run_path(os.path.join(REF_SCRIPTS, script), run_name="reference_script")
    assert "test" in mod, script
    mod["test"]()
